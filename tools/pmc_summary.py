"""Aggregate a rocprofv3 --pmc CSV directory per kernel name: mean counter value per dispatch
and mean duration (kernel-trace), dcv kernels only."""
import csv
import glob
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


vals = defaultdict(lambda: defaultdict(list))
for f in cnt_files:
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "dcv::" not in k:
            continue
        vals[short(k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
for f in kt_files:
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "dcv::" not in k:
            continue
        dur[short(k)].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for k in sorted(vals, key=lambda k: -sum(dur.get(k, [0]))):
    n = len(dur.get(k, []))
    avg_us = sum(dur[k]) / n if n else float("nan")
    print(f"{k}\n   dispatches={n} avg_us={avg_us:.1f} total_ms={sum(dur.get(k, [0])) / 1e3:.2f}")
    for c, v in sorted(vals[k].items()):
        print(f"   {c:28s} mean/dispatch={sum(v) / len(v):.4g}")
