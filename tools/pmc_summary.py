"""Aggregate a rocprofv3 --pmc CSV directory per (kernel name, grid size): mean counter value per
dispatch and mean duration (kernel-trace), dcv kernels only.  Optional 2nd arg: JSON output."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1]
cnt_files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
kt_files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)


def short(name):
    name = re.sub(r"dcv::TileCfg<(\d), (\d), (\d), (\d), (\d+), (\d), true, 0>", r"Cfg\1\2\3\4k\5b\6s", name)   # trailing 0: no pre-split (plane) operands
    name = re.sub(r"dcv::TileCfg<(\d), (\d), (\d), (\d), (\d+), (\d), false, 0>", r"Cfg\1\2\3\4k\5b\6", name)
    name = re.sub(r"dcv::TileCfg<(\d), (\d), (\d), (\d), (\d+), (\d), true>", r"Cfg\1\2\3\4k\5b\6s", name)    # s: split arithmetic
    name = re.sub(r"dcv::TileCfg<(\d), (\d), (\d), (\d), (\d+), (\d), false>", r"Cfg\1\2\3\4k\5b\6", name)
    name = re.sub(r"dcv::TileCfg<(\d), (\d), (\d), (\d), (\d+), (\d)>", r"Cfg\1\2\3\4k\5b\6", name)
    name = re.sub(r"dcv::TileCfg<(\d), (\d), (\d), (\d), (\d+)>", r"Cfg\1\2\3\4k\5", name)
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "")[:100]


# the counter CSV carries only the total grid size: take x/y/z from the kernel trace of the same run
grid_xyz = {}
for f in kt_files:
    for r in csv.DictReader(open(f)):
        grid_xyz[r["Dispatch_Id"]] = "x".join(str(r[k]) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z") if k in r)


def grid_of(r):
    if "Grid_Size_X" in r:
        return "x".join(str(r[k]) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z") if k in r)
    return grid_xyz.get(r.get("Dispatch_Id"), str(r.get("Grid_Size", "")))


vals = defaultdict(lambda: defaultdict(list))
for f in cnt_files:
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "dcv::" not in k:
            continue
        vals[(short(k), grid_of(r))][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
for f in ([] if cnt_files else kt_files):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "dcv::" not in k:
            continue
        dur[(short(k), grid_of(r))].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
seen = set()
for f in cnt_files:
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "dcv::" not in k or r["Dispatch_Id"] in seen:
            continue
        seen.add(r["Dispatch_Id"])
        dur[(short(k), grid_of(r))].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
out = {}
keys = sorted(set(vals) | set(dur), key=lambda k: -sum(dur.get(k, [0])))
for k in keys:
    n = len(dur.get(k, []))
    avg_us = sum(dur[k]) / n if n else float("nan")
    print(f"{k[0]}  grid={k[1]}\n   dispatches={n} avg_us={avg_us:.1f} total_ms={sum(dur.get(k, [0])) / 1e3:.2f}")
    rec = {"dispatches": n, "avg_us": avg_us}
    for c, v in sorted(vals.get(k, {}).items()):
        print(f"   {c:28s} mean/dispatch={sum(v) / len(v):.6g}")
        rec[c] = sum(v) / len(v)
    out[f"{k[0]} grid={k[1]}"] = rec
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
