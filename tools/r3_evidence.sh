# the evidence set of round 3 (profiles/r03_*): bench line as the driver runs it, kernel stats and span, HBM traffic, SQ counters
# of one batch alone, the other configurations, the API timings
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
V=${1:-v1}
O=$GRAFT_REPO_ROOT/gpurun_out/r3ev_$V
mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python -c "import json; d=json.load(open('$O/bench.json')); print('bench', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms_per_launch'],3), round(d['roofline']['frac'],3))"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 4 --warmup 2 --cpu-rays 0 --api-steps 0 --no-extras > $O/stats.log 2>&1)
python3 tools/ktrace_span.py $O/stats > $O/span.txt 2>&1 || true
cat $O/span.txt | tail -5
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-rays 0 --api-steps 0 --no-extras > $O/fetch.log 2>&1)
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-rays 0 --api-steps 0 --no-extras > $O/write.log 2>&1)
python3 tools/pmc_traffic.py $O/fetch $O/write 3 1e8 1 > $O/traffic.json
python -c "import json; d=json.load(open('$O/traffic.json')); print('traffic', d['hbm_bytes_per_launch']/1e9, d['hbm_bytes_per_launch_fetch_as_counted']/1e9)"
export TRC_STREAM_SLOTS=1
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/one -- python3 $R/bench.py --steps 4 --warmup 2 --cpu-rays 0 --api-steps 0 --no-extras > $O/one.log 2>&1)
python3 tools/ktrace_tail.py $O/one 16 > $O/one_timeline.txt 2>&1 || true
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc1 -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-rays 0 --api-steps 0 --no-extras > $O/pmc1.log 2>&1)
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc2 -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-rays 0 --api-steps 0 --no-extras > $O/pmc2.log 2>&1)
python3 tools/pmc_kernels.py 3 1e8 $O/pmc1 $O/pmc2 > $O/sq_counters.json
unset TRC_STREAM_SLOTS
timeout -k 10 300 python tools/api_tree.py 1e7 > $O/api_tree.txt 2>&1; grep run $O/api_tree.txt
timeout -k 10 300 python tools/api_profile.py 1e8 read 2>&1 | head -4 > $O/api_fast.txt; cat $O/api_fast.txt
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dish -- python3 $R/tools/gpu_dish.py > $O/dish.log 2>&1); tail -1 $O/dish.log
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cav -- python3 $R/tools/gpu_cavity.py > $O/cav.log 2>&1); tail -1 $O/cav.log
timeout -k 10 300 python tools/gpu_mesh.py > $O/mesh.txt 2>&1 || true; tail -3 $O/mesh.txt
timeout -k 10 300 python tools/gpu_poly.py 2e6 16 2>&1 | grep -v WARNING > $O/poly.txt || true; tail -4 $O/poly.txt
