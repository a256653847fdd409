set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python tools/gpu_mesh.py 1e7 2>&1 | tail -3
timeout -k 10 900 python -m pytest tests/test_gpu_stream.py -m gpu -x -q -k "mesh" 2>&1 | tail -3
