set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== mesh"; timeout -k 10 300 python tools/gpu_mesh.py 1e7 2>&1 | tail -4
echo "== mesh 2e5"; timeout -k 10 300 python tools/gpu_mesh.py 2e5 2>&1 | tail -1
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
