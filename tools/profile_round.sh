#!/bin/bash
# The measured evidence of one round, produced on the GPU box: tools/profile_round.sh <tag> [lines|traces|pmc|insts ...]
# Writes <tag>_bench_<workload>.json (bench lines), <tag>_bench_<workload>_kernel_stats.csv and _summary.json
# (rocprofv3 kernel trace of the same command, and for the headline the FETCH_SIZE / WRITE_SIZE passes) under
# gpurun_out/<tag>/; what is to be kept is then copied to profiles/.  Every GPU step has its own time limit and
# a failing step ends the script.  ONLY=<regex> restricts every stage to the workloads whose name matches.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; shift
WHAT=${*:-lines traces pmc insts}
O=$R/gpurun_out/$T
mkdir -p $O
line() {
  local name=$1; shift
  [[ $name =~ ^(${ONLY:-.*})$ ]] || return 0
  cd $R
  timeout -k 10 280 python bench.py "$@" > $O/${T}_bench_$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }
  python3 -c "import json,sys; j=json.load(open(sys.argv[1])); print(sys.argv[2], j['ms_per_step'], 'ms', j['value'], j['unit'], 'frac', j['roofline']['frac'])" $O/${T}_bench_$name.json $name
}
trace() {
  local name=$1; shift
  [[ $name =~ ^(${ONLY:-.*})$ ]] || return 0
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/kt_$name
  timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$name -o p -- python3 $R/bench.py --no-cpu-baseline "$@" > /tmp/kt_$name.log 2>&1 || { tail -5 /tmp/kt_$name.log; exit 1; }
  python3 $R/tools/summarize_profile.py --tag ${T}_bench_$name --stats /tmp/kt_$name --out-dir $O > /dev/null || exit 1
  echo "traced $name"
}
summ() {  # the summary of one workload from whatever passes of it this call has made so far
  local name=$1 more=""
  [ -d /tmp/pmc_${name}_FETCH_SIZE ] && more="$more --fetch /tmp/pmc_${name}_FETCH_SIZE"
  [ -d /tmp/pmc_${name}_WRITE_SIZE ] && more="$more --write /tmp/pmc_${name}_WRITE_SIZE"
  [ -d /tmp/pmc_${name}_insts ] && more="$more --insts /tmp/pmc_${name}_insts"
  python3 $R/tools/summarize_profile.py --tag ${T}_bench_$name --stats /tmp/kt_$name $more --out-dir $O > /dev/null
}
pmc() {
  local name=$1; shift
  [[ $name =~ ^(${ONLY:-.*})$ ]] || return 0
  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${name}_$c
    timeout -s KILL 120 rocprofv3 --pmc $c --kernel-include-regex "mcull" --output-format csv -d /tmp/pmc_${name}_$c -o p -- python3 $R/bench.py --steps 3 --warmup 1 --ramp-ms 0 --no-cpu-baseline "$@" > /tmp/pmc_$name.log 2>&1 || { tail -5 /tmp/pmc_$name.log; exit 1; }
    echo "counted $c for $name"
  done
  summ $name || exit 1
}
insts() {
  local name=$1; shift
  [[ $name =~ ^(${ONLY:-.*})$ ]] || return 0
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/pmc_${name}_insts
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-include-regex "mcull" --output-format csv -d /tmp/pmc_${name}_insts -o p -- python3 $R/bench.py --steps 3 --warmup 1 --ramp-ms 0 --no-cpu-baseline "$@" > /tmp/pmc_$name.log 2>&1 || { tail -5 /tmp/pmc_$name.log; exit 1; }
  echo "counted instructions for $name"
  summ $name || exit 1
}
for w in $WHAT; do
  case $w in
    lines)
      # bench.py quotes the counter passes of the committed summaries: after `traces pmc insts` of the same call they are these
      cp $O/${T}_bench_config[345]_summary.json $O/${T}_bench_hiz_summary.json $O/${T}_bench_depth_summary.json $R/profiles/ 2>/dev/null
      line config4 || exit 1
      line config4_hiz_flipped --hiz || exit 1
      line config4_hiz_unflipped --hiz --unflipped || exit 1
      line config4_no_precull --no-precull || exit 1
      line config4_force_dist --force-dist --no-strong-leg || exit 1
      line config4_shard --force-dist --no-strong-leg --instances 97656 || exit 1
      line config4_moving_camera --moving-camera || exit 1
      line config3 --workload config3 --steps 20 --warmup 3 || exit 1
      line config3_random --workload config3 --c3-mesh random --steps 20 --warmup 3 || exit 1
      line config5 --workload config5 --steps 20 --warmup 3 || exit 1
      line hiz --workload hiz || exit 1
      line config1 --workload config1 --graph || exit 1
      line depth --workload depth || exit 1
      line bounds --workload bounds || exit 1
      line meshbuild --workload meshbuild || exit 1
      ;;
    traces)
      trace config4 --steps 50 --warmup 10 || exit 1
      trace config4_hiz_flipped --hiz --steps 50 --warmup 10 || exit 1
      trace config3 --workload config3 --steps 5 --warmup 2 || exit 1
      trace config3_random --workload config3 --c3-mesh random --steps 5 --warmup 2 || exit 1
      trace config5 --workload config5 --steps 5 --warmup 2 || exit 1
      trace hiz --workload hiz --steps 50 --warmup 10 || exit 1
      trace depth --workload depth --steps 50 --warmup 10 || exit 1
      trace config1 --workload config1 --steps 50 --warmup 10 || exit 1
      ;;
    pmc)
      pmc config4 || exit 1
      pmc config3 --workload config3 || exit 1
      pmc config5 --workload config5 || exit 1
      pmc hiz --workload hiz || exit 1
      ;;
    insts)  # after `traces` in the same call: the summary is rewritten with the durations beside the counters
      insts config3 --workload config3 || exit 1
      insts config5 --workload config5 || exit 1
      insts depth --workload depth || exit 1
      ;;
  esac
done
ls $O
