set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3n
mkdir -p $O
echo "== cavity"; timeout -k 10 300 python tools/gpu_cavity.py 5e7 2>&1 | tail -2
echo "== mesh"; timeout -k 10 300 python tools/gpu_mesh.py 1e7 2>&1 | tail -1
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -80 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
