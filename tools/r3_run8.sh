set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j
timeout -k 10 300 python tools/api_tree.py 1e7 > $O/api_tree.log 2>&1; tail -14 $O/api_tree.log
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -80 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for rep in 1 2; do timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-rays 0 --api-steps 0 > $O/bench$rep.json 2> $O/bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/r3j/bench$rep.json'))
print('bench', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms_per_launch'],3))
for k,v in (d.get('other_configs') or {}).items(): print(k, round(v['kernel_ms'],2), round(v['Gsegments_per_s_kernels'],2))
PY
done
