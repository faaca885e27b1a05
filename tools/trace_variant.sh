#!/bin/bash
# Kernel trace of one bench command with a library variant (variants/libmcull_<name>.so, or "base"), on the GPU box:
#   tools/trace_variant.sh <name> <bench.py arguments...>      -> gpurun_out/trace_<name>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$1; shift
cp $R/vulkan-renderer_amd/libmcull.so /tmp/libmcull_keep.so
[ "$V" = base ] || cp $R/variants/libmcull_$V.so $R/vulkan-renderer_amd/libmcull.so
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$V
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$V -o p -- python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 10 "$@" > /tmp/kt_$V.log 2>&1
rc=$?
cp /tmp/libmcull_keep.so $R/vulkan-renderer_amd/libmcull.so
[ $rc -eq 0 ] || { tail -5 /tmp/kt_$V.log; exit 1; }
python3 $R/tools/summarize_profile.py --tag trace_$V --stats /tmp/kt_$V --out-dir $R/gpurun_out > /dev/null || exit 1
head -8 $R/gpurun_out/trace_${V}_kernel_stats.csv
