#!/usr/bin/env python3
"""Median kernel durations per phase of tools/precull_probe.py from a rocprofv3 rocpd database (80 culls per phase)."""
import collections
import glob
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sorted(glob.glob(sys.argv[1] + "/*.db"))[-1])
per = int(sys.argv[2]) if len(sys.argv) > 2 else 80
rows = db.execute("select name, duration from kernels where name like '%mcull::%' order by start").fetchall()
phase = collections.defaultdict(lambda: collections.defaultdict(list))
n = 0
for name, d in rows:
    short = name.split("(")[0].replace("void ", "").split("<")[0].replace("mcull::", "")
    phase[n // per][short].append(d / 1e3)
    if "emit_commands" in short:
        n += 1
for p in sorted(phase):
    print(p, {k: round(float(np.median(v)), 1) for k, v in phase[p].items() if len(v) > 5})
