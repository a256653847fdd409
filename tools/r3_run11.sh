set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3m
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3m
for rep in 1 2; do for f2 in 1 0; do
  TRC_STREAM_FRESH2=$f2 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-rays 0 --api-steps 0 --no-extras > $O/bench_$f2.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
  python -c "import json; d=json.load(open('$O/bench_$f2.json')); print('fresh2=$f2', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms_per_launch'],3), d['check']['receiver_hits'], d['check']['heliostat_hits'], d['check']['ok'])"
done; done
for f2 in 1 0; do echo "== dish fresh2=$f2"; TRC_STREAM_FRESH2=$f2 timeout -k 10 200 python tools/gpu_dish.py 2>&1 | tail -1; done
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -80 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
(cd /tmp && TRC_STREAM_SLOTS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/bench1slot --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --cpu-rays 0 --api-steps 0 --no-extras > $O/bench1slot.log 2>&1)
echo "== NSTTF one slot"; python3 tools/kstats.py $O/bench1slot | sort -k6 -n -r | head -12
