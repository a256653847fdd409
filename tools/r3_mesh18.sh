set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3mesh18
mkdir -p $O
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $GRAFT_REPO_ROOT/tools/gpu_mesh.py 1e7 > $O/stats.log 2>&1); grep -a "^run\|faces:" $O/stats.log
python3 tools/kstats.py $O/stats | sort -k6 -n -r | head -8
python3 tools/ktrace_tail.py $O/stats 24 2>&1 | tail -26
