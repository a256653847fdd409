# mesh timing, the whole GPU test suite, a short bench run: what is run after every change of the kernels
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== mesh"; timeout -k 10 300 python tools/gpu_mesh.py 1e7 2>&1 | tail -1
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== bench"; timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-rays 0 --api-steps 0 --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['roofline']['kernels'] is not None)"
