# diagnostic variants of k_s_bounce<2> (-DSB_DIAG=1..4): time of the launch for the fresh rays with parts of the search left out
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/tracer_amd/lib
O=$GRAFT_REPO_ROOT/gpurun_out/r3mesh10
mkdir -p $O
export TRC_STREAM_REFILL=0
for lib in libtracer_amd.so var_d1.so var_d2.so var_d3.so var_d4.so; do
  (cd /tmp && TRACER_AMD_LIB=$L/$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$lib -- python3 $GRAFT_REPO_ROOT/tools/gpu_mesh.py 1e7 > $O/$lib.log 2>&1)
  echo "== $lib"; grep -a "^run 2" $O/$lib.log | cut -c1-120; python3 tools/kstats.py $O/$lib | grep "k_s_bounce\|k_s_shade"
done
