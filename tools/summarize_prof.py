#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output into small text tables (kept under profiles/)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name, n=90):
    name = name.replace("riiamd::", "")
    return name if len(name) <= n else name[:n - 3] + "..."


def stats(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        print("no kernel_stats.csv under", d)
        return
    rows = list(csv.DictReader(open(files[0])))
    print("# rocprofv3 --kernel-trace --stats : per-kernel summary (durations in ns)")
    print("%-92s %8s %14s %14s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
    for r in rows[:25]:
        print("%-92s %8s %14s %14.0f %8s" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"],
                                            float(r["AverageNs"]), r["Percentage"]))
    tr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if tr:
        per = defaultdict(list)
        for r in csv.DictReader(open(tr[0])):
            if "riiamd" in r["Kernel_Name"]:
                per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r))
        print("\n# riiamd kernels: launch geometry / resources (from kernel_trace.csv, first launch)")
        for k, v in per.items():
            r = v[0][1]
            durs = sorted(x[0] for x in v)
            print("%s\n    calls=%d  min/med/max ns = %d / %d / %d  grid=%s wg=%s lds=%s scratch=%s vgpr=%s accum_vgpr=%s sgpr=%s"
                  % (short(k, 140), len(v), durs[0], durs[len(durs) // 2], durs[-1], r.get("Grid_Size"),
                     r.get("Workgroup_Size"), r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("VGPR_Count"),
                     r.get("Accum_VGPR_Count"), r.get("SGPR_Count")))


def pmc(dirs):
    print("# rocprofv3 --pmc passes: per-kernel counter values averaged over dispatches (sum over XCD/SE instances)")
    for d in dirs:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            print("no counter_collection.csv under", d)
            continue
        acc = defaultdict(lambda: defaultdict(list))
        disp = defaultdict(lambda: defaultdict(float))
        for r in csv.DictReader(open(files[0])):
            disp[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for (k, _), cs in disp.items():
            for c, v in cs.items():
                acc[k][c].append(v)
        for k, cs in acc.items():
            print(short(k, 140))
            for c, vals in sorted(cs.items()):
                print("    %-28s avg %.6g  (n=%d, min %.6g, max %.6g)" % (c, sum(vals) / len(vals), len(vals), min(vals), max(vals)))


def pmcjson(dirs):
    """The same averages as machine-readable JSON: {kernel: {counter: avg per dispatch}} (feeds profiles/pmc.json)."""
    import json
    out = {}
    for d in dirs:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        disp = defaultdict(lambda: defaultdict(float))
        for r in csv.DictReader(open(files[0])):
            disp[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        acc = defaultdict(lambda: defaultdict(list))
        for (k, _), cs in disp.items():
            for c, v in cs.items():
                acc[k][c].append(v)
        for k, cs in acc.items():
            o = out.setdefault(short(k, 200), {})
            for c, vals in cs.items():
                o[c] = sum(vals) / len(vals)
                o["_dispatches"] = len(vals)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "pmcjson":
        pmcjson(sys.argv[2:])
    else:
        pmc(sys.argv[2:])
