#!/bin/bash
# (an entry may carry one environment setting: lib.so@NAME=VALUE)
# A/B of several builds of the library on the bench workload: bench line (two runs each, interleaved) and, with PMC=1, the
# VALU instruction count and busy cycles per dispatch.  usage: tools/ab_libs.sh <tag> lib1.so lib2.so ...
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$tag
for rep in 1 2; do
  for lib in "$@"; do
    envset=""; case "$lib" in *@*) envset="${lib#*@}"; lib="${lib%%@*}";; esac
    name=$(basename $lib .so)${envset:+_$envset}
    [ -n "$envset" ] && export "$envset"
    TRACER_AMD_LIB=$R/$lib timeout -k 10 120 python3 $R/bench.py --steps 6 --warmup 2 --cpu-rays 0 --no-extras 2>/dev/null > $R/gpurun_out/$tag/bench_${name}_$rep.json
    python3 -c "import json,sys; d=json.load(open('$R/gpurun_out/$tag/bench_${name}_$rep.json')); print('$name', $rep, round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms_per_launch'],3))"
    [ -n "$envset" ] && unset "${envset%%=*}"
  done
done
if [ -n "$PMC" ]; then
  for lib in "$@"; do
    envset=""; case "$lib" in *@*) envset="${lib#*@}"; lib="${lib%%@*}";; esac
    name=$(basename $lib .so)${envset:+_$envset}
    [ -n "$envset" ] && export "$envset"
    (cd /tmp && TRACER_AMD_LIB=$R/$lib TRC_STREAM_SLOTS=1 timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/$tag/pmc_$name -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-extras > /dev/null 2>&1)
    [ -n "$envset" ] && unset "${envset%%=*}"
  done
fi
