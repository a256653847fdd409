# usage: r3_ab.sh <libA.so> <libB.so> : bench (x3 interleaved), dish, cavity 5e7
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/tracer_amd/lib
O=$GRAFT_REPO_ROOT/gpurun_out/r3ab
mkdir -p $O
for rep in 1 2 3; do for lib in "$@"; do
  TRACER_AMD_LIB=$L/$lib timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-rays 0 --api-steps 0 --no-extras > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python -c "import json; d=json.load(open('$O/b.json')); print('$lib', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms_per_launch'],3), d['check']['receiver_hits'], d['check']['ok'])"
done; done
for lib in "$@"; do
  echo "== dish $lib"; TRACER_AMD_LIB=$L/$lib timeout -k 10 200 python tools/gpu_dish.py 2>&1 | tail -1
  echo "== cavity $lib"; TRACER_AMD_LIB=$L/$lib timeout -k 10 300 python tools/gpu_cavity.py 5e7 2>&1 | tail -1
done
