set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3ab
mkdir -p $O
for rep in 1 2 3; do for f in 0 1; do
  TRC_STREAM_FIRST=$f timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-rays 0 --api-steps 0 --no-extras > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python -c "import json; d=json.load(open('$O/b.json')); print('first=$f', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms_per_launch'],3), d['check']['receiver_hits'], d['check']['ok'])"
done; done
