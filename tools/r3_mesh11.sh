set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/tracer_amd/lib
export TRC_STREAM_REFILL=0
for d in 2 6 16 40; do echo "== density $d"; TRC_GRID32_DENSITY=$d timeout -k 10 120 python tools/gpu_mesh.py 1e7 2>&1 | tail -1 | cut -c1-150; done
echo "== refill, density 16";  TRC_STREAM_REFILL=1 TRC_GRID32_DENSITY=16 timeout -k 10 120 python tools/gpu_mesh.py 1e7 2>&1 | tail -1 | cut -c1-150
echo "== stats density 2"; TRC_GRID32_DENSITY=2 TRACER_AMD_LIB=$L/var_stats.so timeout -k 10 300 python tools/gpu_mesh.py 1e7 2>&1 | tail -2 | head -1
echo "== stats density 16"; TRC_GRID32_DENSITY=16 TRACER_AMD_LIB=$L/var_stats.so timeout -k 10 300 python tools/gpu_mesh.py 1e7 2>&1 | tail -2 | head -1
timeout -k 10 900 python -m pytest tests/test_gpu_stream.py -m gpu -x -q -k "mesh" 2>&1 | tail -3
