# one batch alone (TRC_STREAM_SLOTS=1): kernel timeline and SQ counters of the cavity and the dish
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3d
mkdir -p $O
export TRC_STREAM_SLOTS=1
for sm in 1 0; do
  (cd /tmp && TRC_STREAM_SMALL=$sm timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/cav_small$sm --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_cavity.py 2e7 > $O/cav_small$sm.log 2>&1)
  echo "== cavity 2e7 one slot small=$sm"; tail -1 $O/cav_small$sm.log; python3 tools/kstats.py $O/cav_small$sm | sort -k6 -n -r | head -10
done
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/dish --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_dish.py 8e6 > $O/dish.log 2>&1)
echo "== dish 8e6 one slot"; tail -1 $O/dish.log; python3 tools/kstats.py $O/dish | sort -k6 -n -r | head -10
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc1 -- python3 $GRAFT_REPO_ROOT/tools/gpu_cavity.py 2e7 > $O/pmc1.log 2>&1)
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc2 -- python3 $GRAFT_REPO_ROOT/tools/gpu_cavity.py 2e7 > $O/pmc2.log 2>&1)
python3 tools/pmc_summary.py $O/pmc1 > $O/pmc1_summary.txt; python3 tools/pmc_summary.py $O/pmc2 > $O/pmc2_summary.txt
cat $O/pmc1_summary.txt $O/pmc2_summary.txt
