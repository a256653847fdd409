set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3i
timeout -k 10 300 python tools/api_tree.py 1e7 d > $O/api_tree.log 2>&1; tail -32 $O/api_tree.log
timeout -k 10 900 python -m pytest tests/test_gpu_stream.py -m gpu -x -q -k "modest_hit_buffer or scattering_slab" 2>&1 | tail -5
