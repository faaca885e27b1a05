#!/bin/bash
# The GPU suite once under each evaluation switch of the library (one process per combination, each with its own time
# limit; a run that ends in a signal or a time limit ends the script, a run with failing tests does not):
#   tools/forced_suite.sh [name ...]      logs: gpurun_out/forced_<name>.log
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
declare -A COMBO=(
  [streaming]="MCULL_FORCE_STREAMING=1"
  [no_flags]="MCULL_PRECULL_FLAGS=0"
  [no_groups]="MCULL_PRECULL_GROUPS=0"
  [no_adaptive]="MCULL_PRECULL_ADAPTIVE=0"
  [no_strips]="MCULL_HIZ_STRIPS=0"
  [no_windows]="MCULL_HIZ_WINDOWS=0"
  [hiz_stream]="MCULL_HIZ_STREAM=1"
  [hiz_tiles]="MCULL_HIZ_TILES=1"
  [hiz_single]="MCULL_HIZ_SINGLE_LAUNCH=1"
  [short_rounds]="MCULL_COMPACT_SHORT_ROUNDS=1"
  [all_off]="MCULL_FORCE_STREAMING=1 MCULL_PRECULL_FLAGS=0 MCULL_PRECULL_GROUPS=0 MCULL_PRECULL_ADAPTIVE=0 MCULL_HIZ_STRIPS=0 MCULL_HIZ_WINDOWS=0"
)
NAMES=${*:-streaming no_flags no_groups no_adaptive no_strips no_windows hiz_stream hiz_tiles hiz_single short_rounds all_off}
for n in $NAMES; do
  env ${COMBO[$n]} timeout -k 10 400 python -u -m pytest tests -q -m gpu --timeout 200 --timeout-method thread -p no:cacheprovider > gpurun_out/forced_$n.log 2>&1
  rc=$?
  echo "$n (${COMBO[$n]}): exit $rc: $(tail -1 gpurun_out/forced_$n.log)"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
done
