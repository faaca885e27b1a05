#!/bin/bash
# per-launch memory-side read requests of the mask pass: does a cold launch fetch more than a repeated one?
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_run
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "cull_mask" --output-format csv -d /tmp/pmc_run -o p -- python3 $R/tools/cold_probe.py > /tmp/pmc_run.log 2>&1 || { tail -5 /tmp/pmc_run.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(glob.glob("/tmp/pmc_run/**/*counter_collection.csv", recursive=True)[0])):
    if "cull_mask" in r["Kernel_Name"]:
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for d in sorted(rows):
    print(d, {k: int(v) for k, v in sorted(rows[d].items())})
PY
