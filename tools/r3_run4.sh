set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3f
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 300 python tools/api_profile.py 1e8 read > $O/api_profile.log 2>&1; head -45 $O/api_profile.log
timeout -k 10 300 python tools/api_tree.py 1e7 > $O/api_tree.log 2>&1; cat $O/api_tree.log
echo "== dish"; timeout -k 10 200 python tools/gpu_dish.py 2>&1 | tail -1
echo "== cavity"; timeout -k 10 300 python tools/gpu_cavity.py 5e7 2>&1 | tail -1
