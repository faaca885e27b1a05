#!/bin/bash
# kernel-trace summary of one bench workload with extra flags: tools/trace_workload.sh <workload> [flags...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
W=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_run
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_run -o p -- python3 $R/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline "$@" > /tmp/kt_run.log 2>&1 || { tail -5 /tmp/kt_run.log; exit 1; }
echo "== $W $*"
python3 $R/tools/pmc_summary.py /tmp/kt_run
grep '"metric"' /tmp/kt_run.log | python3 -c "import sys,json; j=json.loads(sys.stdin.readline()); print('ms_per_step', j['ms_per_step'], j['config'].get('instance_precull'))"
