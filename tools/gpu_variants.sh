#!/bin/bash
# usage: gpu_variants.sh <tag> <lib.so> ...  -- per-kernel rocprofv3 stats of the streaming engine for each library build
cd /tmp && export TMPDIR=/tmp
export TRC_FAST_STREAM=1
while [ $# -gt 1 ]; do
  tag=$1; lib=$2; shift 2
  export TRACER_AMD_LIB=$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/var_$tag -- python3 /root/repo/tools/gpu_one.py 2e7 kd > /root/repo/gpurun_out/var_$tag.log 2>&1 || exit 1
  echo "== $tag: $(grep accel /root/repo/gpurun_out/var_$tag.log)"
  python3 /root/repo/tools/kstats.py /root/repo/gpurun_out/var_$tag
done
