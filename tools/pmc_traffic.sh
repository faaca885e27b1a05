#!/bin/bash
# Memory-side request counters of the config-4 step, one rocprofv3 run per counter set (a run holds at most 4 TCC
# counters; FETCH_SIZE takes 3 of them, WRITE_SIZE 2).  Usage: tools/pmc_traffic.sh <tag> [bench flags...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_traffic_$TAG.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  rm -rf /tmp/pmc_run
  echo "== $TAG: $1" | tee -a $OUT
  timeout -s KILL 120 rocprofv3 --pmc $1 --kernel-include-regex "mcull" --output-format csv -d /tmp/pmc_run -o p -- python3 $R/bench.py --steps 3 --warmup 1 --ramp-ms 0 --no-cpu-baseline "${@:2}" > /tmp/pmc_run.log 2>&1 || { tail -5 /tmp/pmc_run.log; exit 1; }
  python3 $R/tools/pmc_summary.py /tmp/pmc_run --per-launch | tee -a $OUT
}
run "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "$@" || exit 1
run "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_HIT_sum TCC_MISS_sum" "$@" || exit 1
run "FETCH_SIZE" "$@" || exit 1
run "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "$@" || exit 1
