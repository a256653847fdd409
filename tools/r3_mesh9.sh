set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rf in 0 1; do echo "== refill $rf"; TRC_STREAM_REFILL=$rf timeout -k 10 300 python tools/gpu_mesh_sorted.py 1e7 2>&1 | tail -4; done
