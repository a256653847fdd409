set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/tracer_amd/lib
echo "== coop, 2e5 rays first"; timeout -k 10 60 python tools/gpu_mesh.py 2e5 2>&1 | tail -1 | cut -c1-200
echo "== coop"; timeout -k 10 60 python tools/gpu_mesh.py 1e7 2>&1 | tail -2 | cut -c1-200
echo "== lane per ray"; TRC_STREAM_COOP=0 timeout -k 10 60 python tools/gpu_mesh.py 1e7 2>&1 | tail -1 | cut -c1-200
echo "== stats"; TRACER_AMD_LIB=$L/var_stats.so timeout -k 10 120 python tools/gpu_mesh.py 1e7 2>&1 | tail -3 | head -2
timeout -k 10 600 python -m pytest tests/test_gpu_stream.py -m gpu -x -q -k "mesh" 2>&1 | tail -3
