set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3h
( time timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3h/bench.json'))
print('bench', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms_per_launch'],3), 'frac', round(d['roofline']['frac'],3))
print({k:v for k,v in d['api_level'].items() if k!='includes'})
for k,v in (d.get('other_configs') or {}).items(): print(k, {a:b for a,b in v.items() if a!='workload'})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
timeout -k 10 300 python tools/api_tree.py 1e7 > $O/api_tree.log 2>&1; cat $O/api_tree.log
(cd /tmp && TRC_STREAM_SLOTS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/bench1slot --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --cpu-rays 0 --api-steps 0 --no-extras > $O/bench1slot.log 2>&1)
echo "== NSTTF one slot"; python3 tools/kstats.py $O/bench1slot | sort -k6 -n -r | head -14
timeout -k 10 600 python -m pytest tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -3
