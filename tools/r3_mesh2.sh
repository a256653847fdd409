set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/tracer_amd/lib
for lib in libtracer_amd.so var_prev.so; do echo "== mesh $lib"; TRACER_AMD_LIB=$L/$lib timeout -k 10 300 python tools/gpu_mesh.py 1e7 2>&1 | tail -2; done
timeout -k 10 900 python -m pytest tests/test_gpu_stream.py -m gpu -x -q -k "mesh or routes or overflow" 2>&1 | tail -3
