set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/tracer_amd/lib
for lib in libtracer_amd.so var_t768.so var_t512.so; do echo "== $lib"; TRACER_AMD_LIB=$L/$lib timeout -k 10 120 python tools/gpu_mesh.py 1e7 2>&1 | tail -1 | cut -c1-150; done
