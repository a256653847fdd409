set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3c
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-rays 0 > $O/bench.json 2> $O/bench.err
python -c "import json; d=json.load(open('$O/bench.json')); print('bench', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms_per_launch'],3), d.get('api_level'))"
echo "== dish"; timeout -k 10 200 python tools/gpu_dish.py 2>&1 | tail -2
echo "== cavity"; timeout -k 10 300 python tools/gpu_cavity.py 5e7 2>&1 | tail -2
echo "== dish small=0"; TRC_STREAM_SMALL=0 timeout -k 10 200 python tools/gpu_dish.py 2>&1 | tail -1
echo "== cavity small=0"; TRC_STREAM_SMALL=0 timeout -k 10 300 python tools/gpu_cavity.py 5e7 2>&1 | tail -1
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/dish --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_dish.py > $O/dish.log 2>&1)
echo "== dish kernels"; python3 tools/kstats.py $O/dish | sort -k6 -n -r | head -12
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/cav --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_cavity.py 5e7 > $O/cav.log 2>&1)
echo "== cavity kernels"; python3 tools/kstats.py $O/cav | sort -k6 -n -r | head -14
