#!/usr/bin/env python3
"""Per-kernel averages from rocprofv3 CSV output: counters per wave (a --pmc run's *_counter_collection.csv) or
durations (a --kernel-trace --stats run's *_kernel_stats.csv).  Prints a few lines; the raw files can then go."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
ks = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
if cc:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(cc[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    if "--per-launch" in sys.argv:  # raw counter values per launch (memory-side request counters)
        for k, c in acc.items():
            print("%-46s launches %3d  " % (k[:46], len(n[k])) + "  ".join("%s %.0f" % (x, v / len(n[k])) for x, v in sorted(c.items())))
        acc = {}
    for k, c in acc.items():
        w = c.get("SQ_WAVES", 0.0) or 1.0
        print("%-46s launches %3d waves/launch %9d " % (k[:46], len(n[k]), w / len(n[k])) +
              "  ".join("%s/wave %.1f" % (x.replace("SQ_INSTS_", ""), v / w) for x, v in sorted(c.items()) if x != "SQ_WAVES"))
for f in ks:
    for r in csv.DictReader(open(f)):
        if "mcull" in r["Name"]:
            print("%-60s calls %4s avg_us %10.1f" % (r["Name"].split("(")[0].replace("void ", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
