#!/usr/bin/env python3
"""Condenses rocprofv3 output (kernel-trace stats and PMC passes) into small files under profiles/.

    python tools/summarize_profile.py --tag r01 --stats gpurun_out/prof4 --fetch gpurun_out/pmc/fetch --write gpurun_out/pmc/write

Counter handling follows the MI355X guide: FETCH_SIZE and WRITE_SIZE are collected in separate
passes (KiB per dispatch); on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so
it is doubled; WRITE_SIZE is taken as is.
"""
import argparse
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def stats_from_db(args, out_dir, summary):
    import sqlite3

    db = sqlite3.connect(sorted(glob.glob(os.path.join(args.stats, "**", "*_results.db"), recursive=True))[-1])
    q = ("select name, count(*), avg(duration), min(duration), max(duration), sum(duration), max(vgpr_count), "
         "max(sgpr_count), max(lds_size) from kernels where name like ? group by name order by sum(duration) desc")
    with open(os.path.join(out_dir, "%s_kernel_stats.csv" % args.tag), "w") as o:
        o.write("kernel,calls,avg_us,min_us,max_us,total_us,vgprs,sgprs,lds_bytes\n")
        for name, calls, avg, mn, mx, tot, vg, sg, lds in db.execute(q, ("%" + args.prefix + "%",)):
            o.write("%s,%d,%.3f,%.3f,%.3f,%.3f,%d,%d,%d\n" % (short(name), calls, avg / 1e3, mn / 1e3, mx / 1e3, tot / 1e3, vg, sg, lds))
            summary.setdefault(short(name), {})["avg_us"] = round(avg / 1e3, 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--insts", help="a --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE pass of the same command")
    ap.add_argument("--prefix", default="mcull::")
    ap.add_argument("--out-dir", default=os.path.join(ROOT, "profiles"))
    args = ap.parse_args()
    out_dir = args.out_dir
    os.makedirs(out_dir, exist_ok=True)
    summary = {}
    if args.stats and not glob.glob(os.path.join(args.stats, "**", "*kernel_stats.csv"), recursive=True):
        stats_from_db(args, out_dir, summary)  # rocprofv3's default output is a rocpd sqlite database
    elif args.stats:
        f = sorted(glob.glob(os.path.join(args.stats, "**", "*kernel_stats.csv"), recursive=True))[-1]
        rows = [r for r in csv.DictReader(open(f)) if args.prefix in r["Name"]]
        with open(os.path.join(out_dir, "%s_kernel_stats.csv" % args.tag), "w") as o:
            o.write("kernel,calls,avg_us,min_us,max_us,total_us\n")
            for r in rows:
                o.write("%s,%s,%.3f,%.3f,%.3f,%.3f\n" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                       float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                       float(r["TotalDurationNs"]) / 1e3))
                summary.setdefault(short(r["Name"]), {})["avg_us"] = round(float(r["AverageNs"]) / 1e3, 3)
    for kind, path, factor in (("fetch", args.fetch, 2.0), ("write", args.write, 1.0)):
        if not path:
            continue
        acc = {}
        csvs = sorted(glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True))
        if csvs:
            for r in csv.DictReader(open(csvs[-1])):
                if args.prefix in r["Kernel_Name"]:
                    acc.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
        else:
            import sqlite3

            db = sqlite3.connect(sorted(glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True))[-1])
            for name, value in db.execute("select kernel_name, value from counters_collection where kernel_name like ?",
                                          ("%" + args.prefix + "%",)):
                acc.setdefault(short(name), []).append(float(value))
        for k, v in acc.items():
            kib = sum(v) / len(v)
            summary.setdefault(k, {})["%s_size_kib_raw" % kind] = round(kib, 1)
            summary[k]["%s_bytes_corrected" % kind] = int(kib * 1024 * factor)
    if args.insts:
        # wave-instructions per launch, and the clock while the kernel ran: GRBM_GUI_ACTIVE sums the busy cycles of the 8
        # XCDs, the timestamps of the same dispatch give its duration in that (serialised) pass
        acc = {}
        f = sorted(glob.glob(os.path.join(args.insts, "**", "*counter_collection.csv"), recursive=True))[-1]
        for r in csv.DictReader(open(f)):
            if args.prefix not in r["Kernel_Name"]:
                continue
            k = acc.setdefault(short(r["Kernel_Name"]), {"launches": set(), "ns": {}})
            k["launches"].add(r["Dispatch_Id"])
            k[r["Counter_Name"]] = k.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            k["ns"][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for name, k in acc.items():
            n = len(k["launches"])
            row = summary.setdefault(name, {})
            row["waves_per_launch"] = int(k.get("SQ_WAVES", 0.0) / n)
            row["valu_insts_per_launch"] = int(k.get("SQ_INSTS_VALU", 0.0) / n)
            row["salu_insts_per_launch"] = int(k.get("SQ_INSTS_SALU", 0.0) / n)
            if row["waves_per_launch"]:
                row["valu_per_wave"] = round(row["valu_insts_per_launch"] / row["waves_per_launch"], 1)
                row["salu_per_wave"] = round(row["salu_insts_per_launch"] / row["waves_per_launch"], 1)
            ns = sum(k["ns"].values())
            if ns and k.get("GRBM_GUI_ACTIVE"):
                row["counter_pass_avg_us"] = round(ns / n / 1e3, 3)
                row["clock_mhz"] = round(k["GRBM_GUI_ACTIVE"] / 8.0 / ns * 1e3, 1)
    kept = os.path.join(ROOT, "profiles", "%s_summary.json" % args.tag)
    if os.path.exists(kept):  # hand-written notes of the committed summary ("_..." keys) stay
        with open(kept) as f:
            summary.update({k: v for k, v in json.load(f).items() if k.startswith("_")})
    with open(os.path.join(out_dir, "%s_summary.json" % args.tag), "w") as o:
        json.dump(summary, o, indent=1, sort_keys=True)
    print(json.dumps(summary, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
