set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3e
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for five in 1 0; do
TRC_SHADE_FIVE=$five timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-rays 0 > $O/bench$five.json 2> $O/bench$five.err
python -c "import json; d=json.load(open('$O/bench$five.json')); print('bench five=$five', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms_per_launch'],3)); a=d.get('api_level') or {}; print({k:v for k,v in a.items() if k!='includes'})"
echo "== dish five=$five"; TRC_SHADE_FIVE=$five timeout -k 10 200 python tools/gpu_dish.py 2>&1 | tail -1
echo "== cavity five=$five"; TRC_SHADE_FIVE=$five timeout -k 10 300 python tools/gpu_cavity.py 5e7 2>&1 | tail -1
done
