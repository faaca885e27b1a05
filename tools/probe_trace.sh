#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/probe_run
timeout -s KILL 250 rocprofv3 --kernel-trace --stats -d /tmp/probe_run -o probe -- python3 $R/tools/precull_probe.py "$@" 2>/dev/null | grep precull
python3 $R/tools/phase_stats.py /tmp/probe_run
