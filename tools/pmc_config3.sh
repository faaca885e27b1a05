#!/bin/bash
# instruction counters of a cull workload's mask pass (reduced instance count), fast and all-IEEE predicate forms,
# and the kernel trace of the full-size workload in both forms (separate rocprofv3 runs); summaries only are kept
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-config3}
SMALL=${2:---instances 100000}
OUT=$R/gpurun_out/pmc_${W}.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
for mode in fast exact; do
  extra=""; [ $mode = exact ] && extra="--exact"
  rm -rf /tmp/pmc_run /tmp/kt_run
  echo "== $W $mode: counters ($SMALL)" | tee -a $OUT
  timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-include-regex "mcull" --output-format csv -d /tmp/pmc_run -o p -- python3 $R/bench.py --workload $W $SMALL --steps 2 --warmup 1 --ramp-ms 0 --no-cpu-baseline $extra > /tmp/pmc_run.log 2>&1 || { tail -5 /tmp/pmc_run.log; exit 1; }
  python3 $R/tools/pmc_summary.py /tmp/pmc_run | tee -a $OUT
  echo "== $W $mode: kernel trace (full size)" | tee -a $OUT
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_run -o p -- python3 $R/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline $extra > /tmp/kt_run.log 2>&1 || { tail -5 /tmp/kt_run.log; exit 1; }
  python3 $R/tools/pmc_summary.py /tmp/kt_run | tee -a $OUT
  grep '"metric"' /tmp/kt_run.log | python3 -c "import sys,json; j=json.loads(sys.stdin.readline()); print('ms_per_step', j['ms_per_step'])" | tee -a $OUT
done
