# mesh diagnostics: what the scattered tallies cost (a library built with -DSHC_DIAG_NO_TALLY), and the cell density of the large grid
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/tracer_amd/lib
O=$GRAFT_REPO_ROOT/gpurun_out/r3mesh6
mkdir -p $O
echo "== no tallies"; (cd /tmp && TRACER_AMD_LIB=$L/var_notally.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/nt -- python3 $GRAFT_REPO_ROOT/tools/gpu_mesh.py 1e7 > $O/nt.log 2>&1); grep -a "^run" $O/nt.log | tail -1; python3 tools/kstats.py $O/nt | sort -k6 -n -r | head -4
for d in 1 4 8 16; do echo "== density $d"; TRC_GRID32_DENSITY=$d timeout -k 10 300 python tools/gpu_mesh.py 1e7 2>&1 | tail -1; done
