#!/bin/bash
# per-dispatch durations of the kernels matching a pattern (a regular expression; ROWS = how many of the last to print), in launch order, for one bench command:
#   tools/trace_dispatches.sh <pattern> <bench.py arguments...>
R=${GRAFT_REPO_ROOT:-/root/repo}
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_disp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_disp -o p -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > /tmp/kt_disp.log 2>&1 || { tail -5 /tmp/kt_disp.log; exit 1; }
python3 - "$PAT" <<'PY'
import csv, glob, re, sys
pat = sys.argv[1]
f = glob.glob("/tmp/kt_disp/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if re.search(pat, r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows[-int(__import__("os").environ.get("ROWS", "24")):]:
    b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (b - prev) / 1e3 if prev is not None else 0.0
    prev = e
    print("%-50s %8.1f us  gap %8.1f us  grid %s" % (r["Kernel_Name"].split("(")[0][-50:], (e - b) / 1e3, gap, r.get("Grid_Size_X", "?")))
PY
