set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export TRC_STREAM_REFILL=0
for d in 2 6 16; do echo "== density $d"; TRC_GRID32_DENSITY=$d timeout -k 10 120 python tools/gpu_mesh.py 1e7 2>&1 | tail -1 | cut -c1-150; done
echo "== refill, density 6";  TRC_STREAM_REFILL=1 TRC_GRID32_DENSITY=6 timeout -k 10 120 python tools/gpu_mesh.py 1e7 2>&1 | tail -1 | cut -c1-150
timeout -k 10 900 python -m pytest tests/test_gpu_stream.py -m gpu -x -q -k "mesh" 2>&1 | tail -3
