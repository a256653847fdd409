set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1 || { tail -40 gpurun_out/r3a/pytest.log; exit 1; }
tail -3 gpurun_out/r3a/pytest.log
for sp in 1 0; do
  TRC_STREAM_SHADE_SPLIT=$sp timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-rays 0 > gpurun_out/r3a/bench_split$sp.json 2> gpurun_out/r3a/bench_split$sp.err
  python -c "import json; d=json.load(open('gpurun_out/r3a/bench_split$sp.json')); print('split$sp', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms_per_launch'],3))"
  echo "== dish split=$sp"; TRC_STREAM_SHADE_SPLIT=$sp timeout -k 10 200 python tools/gpu_dish.py 2>&1 | tail -2
  echo "== cavity split=$sp"; TRC_STREAM_SHADE_SPLIT=$sp timeout -k 10 300 python tools/gpu_cavity.py 2>&1 | tail -4
done
