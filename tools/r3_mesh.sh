set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MESH_OUT=${MESH_OUT:-r3mesh}
O=$GRAFT_REPO_ROOT/gpurun_out/$MESH_OUT
mkdir -p $O
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/gpu_mesh.py 1e7 > $O/stats.log 2>&1); grep -a "^run\|faces:" $O/stats.log
python3 tools/kstats.py $O/stats | sort -k6 -n -r | head -10
(cd /tmp && TRC_STREAM_SLOTS=1 timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc1 -- python3 $R/tools/gpu_mesh.py 1e7 > $O/pmc1.log 2>&1)
(cd /tmp && TRC_STREAM_SLOTS=1 timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc2 -- python3 $R/tools/gpu_mesh.py 1e7 > $O/pmc2.log 2>&1) || true
python3 tools/pmc_kernels.py 3 1e7 $O/pmc1 $O/pmc2 > $O/sq.json
python3 - <<'PY'
import json
import os
d=json.load(open('gpurun_out/%s/sq.json' % os.environ.get('MESH_OUT','r3mesh')))
for k,v in d['per_kernel'].items():
    c=v['counters']
    if c.get('SQ_WAVES',0)<64: continue
    print('%-44s waves %7d valu/wc %.3f wait %.2f wait_inst %.2f' % (k[:44], c['SQ_WAVES'], v.get('valu_per_wave_cycle',0), v.get('wait_frac',0), v.get('wait_inst_frac',0)), {a:'%.3g'%b for a,b in c.items() if a not in ('SQ_WAVES',)})
PY
