set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/tracer_amd/lib
echo "== product"; timeout -k 10 120 python tools/gpu_mesh.py 1e7 2>&1 | tail -2
echo "== stats"; TRACER_AMD_LIB=$L/var_stats.so timeout -k 10 300 python tools/gpu_mesh.py 1e7 2>&1 | tail -3 | head -2
timeout -k 10 900 python -m pytest tests/test_gpu_stream.py -m gpu -x -q -k "mesh" 2>&1 | tail -3
