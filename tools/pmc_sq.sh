#!/bin/bash
# instruction and wait counters of one bench workload's kernels: tools/pmc_sq.sh <workload> [bench flags...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
W=$1; shift
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  rm -rf /tmp/pmc_run
  echo "== $W $*: $set"
  timeout -s KILL 150 rocprofv3 --pmc $set --kernel-include-regex "mcull" --output-format csv -d /tmp/pmc_run -o p -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --ramp-ms 0 --no-cpu-baseline "$@" > /tmp/pmc_run.log 2>&1 || { tail -5 /tmp/pmc_run.log; exit 1; }
  python3 $R/tools/pmc_summary.py /tmp/pmc_run
done
