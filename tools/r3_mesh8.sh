set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rad in 12 3 0.5; do for rf in 1 0; do echo "== source radius $rad refill $rf"; MESH_SRC_RADIUS=$rad TRC_STREAM_REFILL=$rf timeout -k 10 120 python tools/gpu_mesh.py 1e7 2>&1 | tail -1; done; done
