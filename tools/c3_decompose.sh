#!/bin/bash
# GPU box: config 3 with subsets of its tests (kernel trace each): tools/c3_decompose.sh <variant> "<tests> ..." [bench flags]
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$1; T=$2; shift; shift
for t in $T; do
  echo "== variant $V tests $t $*"
  $R/tools/trace_variant.sh $V --workload config3 --tests $t "$@" | grep -E "cull_mask|precull|compact" || exit 1
done
