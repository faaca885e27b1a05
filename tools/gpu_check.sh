#!/bin/bash
# GPU box: the gpu test suite, then the default bench line, then a rocprofv3 kernel trace of the same command.
# Each GPU step has its own time limit; a failing step ends the script.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 700 python -u -m pytest tests -x -q --timeout 120 --timeout-method thread -m gpu > gpurun_out/gpu_suite.log 2>&1
rc=$?; tail -4 gpurun_out/gpu_suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/b1.json 2> gpurun_out/b1.err || { tail -5 gpurun_out/b1.err; exit 1; }
cat gpurun_out/b1.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c4 -o c4 -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof_c4.log 2>&1
echo "rocprof rc=$?"
find $R/gpurun_out/prof_c4 -name "*kernel_stats*" | head
