set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3k
L=$GRAFT_REPO_ROOT/tracer_amd/lib
for rep in 1 2; do for lib in libtracer_amd.so var_pref.so var_512.so; do
  TRACER_AMD_LIB=$L/$lib timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-rays 0 --api-steps 0 --no-extras > $O/bench_$lib.$rep.json 2> /dev/null
  python -c "import json; d=json.load(open('$O/bench_$lib.$rep.json')); print('$lib', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms_per_launch'],3))"
done; done
for lib in libtracer_amd.so var_pref.so var_512.so; do
  echo "== dish $lib"; TRACER_AMD_LIB=$L/$lib timeout -k 10 200 python tools/gpu_dish.py 2>&1 | tail -1
  echo "== cavity $lib"; TRACER_AMD_LIB=$L/$lib timeout -k 10 300 python tools/gpu_cavity.py 5e7 2>&1 | tail -1
done
export TRC_STREAM_SLOTS=1
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-rays 0 --api-steps 0 --no-extras > $O/pmc1.log 2>&1)
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-rays 0 --api-steps 0 --no-extras > $O/pmc2.log 2>&1)
python3 tools/pmc_kernels.py 3 1e8 $O/pmc1 $O/pmc2 > $O/sq_counters.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r3k/sq_counters.json'))
for k,v in d['per_kernel'].items():
    c=v['counters']
    if c.get('SQ_WAVES',0)<64: continue
    print('%-44s waves %7d valu/wc %.3f wait %.2f wait_inst %.2f  valu %.3g salu %.3g lds %.3g vmem %.3g bankconf %.3g' % (k[:44], c['SQ_WAVES'], v.get('valu_per_wave_cycle',0), v.get('wait_frac',0), v.get('wait_inst_frac',0), c.get('SQ_INSTS_VALU',0), c.get('SQ_INSTS_SALU',0), c.get('SQ_INSTS_LDS',0), c.get('SQ_INSTS_VMEM',0), c.get('SQ_LDS_BANK_CONFLICT',0)))
PY
unset TRC_STREAM_SLOTS
timeout -k 10 300 python tools/api_tree.py 1e7 > $O/api_tree.log 2>&1; grep "run" $O/api_tree.log
