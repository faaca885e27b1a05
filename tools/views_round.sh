#!/bin/bash
# GPU box: the tests that go through the multi-view pass, then config 5 with the tree's library and the named variants.
#   tools/views_round.sh [variant ...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_streaming_kernels.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py tests/test_gpu_precull_hiz.py tests/test_gpu_precull.py tests/test_gpu_guarded.py tests/test_gpu_depth.py tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -m gpu -k "view or config5 or fuzz" > gpurun_out/views_tests.log 2>&1
rc=$?; tail -5 gpurun_out/views_tests.log
[ $rc -eq 0 ] || exit $rc
tools/try_variants.sh "--workload config5" base "$@" || exit 1
tools/trace_variant.sh base --workload config5 || exit 1
