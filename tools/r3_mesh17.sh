set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/tracer_amd/lib
for lib in libtracer_amd.so var_c2.so var_c3.so var_c6.so; do echo "== $lib"; TRACER_AMD_LIB=$L/$lib timeout -k 10 60 python tools/gpu_mesh.py 1e7 2>&1 | tail -1 | cut -c1-150; done
for d in 3 12; do echo "== density $d"; TRC_GRID32_DENSITY=$d timeout -k 10 60 python tools/gpu_mesh.py 1e7 2>&1 | tail -1 | cut -c1-150; done
