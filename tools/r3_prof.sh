# per-kernel times of dish (configs[1]) and cavity (configs[4], one rank's share) with and without the class split,
# and the bench with the library built without machine LICM throughout
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $O
for sp in 1 0; do
  (cd /tmp && TRC_STREAM_SHADE_SPLIT=$sp timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/dish_split$sp --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_dish.py > $O/dish_split$sp.log 2>&1)
  echo "== dish split=$sp"; python3 tools/kstats.py $O/dish_split$sp | sort -k6 -n -r | head -12
  (cd /tmp && TRC_STREAM_SHADE_SPLIT=$sp timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/cav_split$sp --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_cavity.py 5e7 > $O/cav_split$sp.log 2>&1)
  echo "== cavity split=$sp"; python3 tools/kstats.py $O/cav_split$sp | sort -k6 -n -r | head -14
done
for rep in 1 2; do for lib in libtracer_amd.so var_nolicm.so; do
  TRACER_AMD_LIB=$GRAFT_REPO_ROOT/tracer_amd/lib/$lib timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-rays 0 > $O/bench_$lib.$rep.json 2> /dev/null
  python -c "import json; d=json.load(open('$O/bench_$lib.$rep.json')); print('$lib', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms_per_launch'],3))"
done; done
for lib in libtracer_amd.so var_nolicm.so; do
  echo "== dish $lib"; TRACER_AMD_LIB=$GRAFT_REPO_ROOT/tracer_amd/lib/$lib timeout -k 10 200 python tools/gpu_dish.py 2>&1 | tail -2
  echo "== cavity $lib"; TRACER_AMD_LIB=$GRAFT_REPO_ROOT/tracer_amd/lib/$lib timeout -k 10 300 python tools/gpu_cavity.py 5e7 2>&1 | tail -1
done
