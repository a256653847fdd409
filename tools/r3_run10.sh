set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3l
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3l
L=$GRAFT_REPO_ROOT/tracer_amd/lib
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -80 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for rep in 1 2; do for cfg in "libtracer_amd.so TRC_X=0" "libtracer_amd.so TRC_STREAM_ABSORB=1" "var_pref2.so TRC_X=0"; do
  set -- $cfg
  env $2 TRACER_AMD_LIB=$L/$1 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-rays 0 --api-steps 0 --no-extras > $O/bench.json 2> /dev/null
  python -c "import json; d=json.load(open('$O/bench.json')); print('$1 $2', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms_per_launch'],3))"
done; done
for cfg in "libtracer_amd.so TRC_X=0" "libtracer_amd.so TRC_STREAM_ABSORB=1" "var_pref2.so TRC_X=0"; do
  set -- $cfg
  echo "== dish $1 $2"; env $2 TRACER_AMD_LIB=$L/$1 timeout -k 10 200 python tools/gpu_dish.py 2>&1 | tail -1
  echo "== cavity $1 $2"; env $2 TRACER_AMD_LIB=$L/$1 timeout -k 10 300 python tools/gpu_cavity.py 5e7 2>&1 | tail -1
done
