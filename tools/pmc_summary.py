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
    name = re.sub(r"dcv::TileCfg<(\d), (\d), (\d), (\d), (\d+)>", r"Cfg\1\2\3\4k\5", name)
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "")[:100]


def grid_of(r):
    if "Grid_Size" in r and r["Grid_Size"]:
        return str(r["Grid_Size"])
    keys = [k for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z") if k in r]
    return "x".join(str(r[k]) for k in keys)


vals = defaultdict(lambda: defaultdict(list))
for f in cnt_files:
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "dcv::" not in k:
            continue
        vals[(short(k), grid_of(r))][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
for f in kt_files:
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "dcv::" not in k:
            continue
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
