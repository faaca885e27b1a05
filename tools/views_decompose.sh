#!/bin/bash
# GPU box: the multi-view mask pass of config 5 with subsets of its tests (kernel trace each): tools/views_decompose.sh <variant> "<tests> ..."
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$1; shift
for t in $1; do
  echo "== variant $V tests $t"
  $R/tools/trace_variant.sh $V --workload config5 --tests $t | grep -E "views_kernel|classify" || exit 1
done
