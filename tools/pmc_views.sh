#!/bin/bash
# instruction counters of the multi-view mask pass of config 5 for one library variant and test subset:
#   tools/pmc_views.sh <variant|base> <tests>
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$1; T=$2
cp $R/vulkan-renderer_amd/libmcull.so /tmp/libmcull_keep.so
[ "$V" = base ] || cp $R/variants/libmcull_$V.so $R/vulkan-renderer_amd/libmcull.so
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_run
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_LDS --kernel-include-regex "cull_mask_views" --output-format csv -d /tmp/pmc_run -o p -- python3 $R/bench.py --workload config5 --tests $T --steps 2 --warmup 1 --ramp-ms 0 --no-cpu-baseline > /tmp/pmc_run.log 2>&1
rc=$?
cp /tmp/libmcull_keep.so $R/vulkan-renderer_amd/libmcull.so
[ $rc -eq 0 ] || { tail -5 /tmp/pmc_run.log; exit 1; }
echo "== $V tests $T"; python3 $R/tools/pmc_summary.py /tmp/pmc_run
