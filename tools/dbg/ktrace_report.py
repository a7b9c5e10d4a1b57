"""Durations of the dispatches of kernels matching a substring in a rocprofv3 kernel_trace.csv, grouped by grid size."""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        d[r.get("Grid_Size_X") or r.get("Grid_Size")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    print("grid", k, "n", len(v), "avg_us=%.1f min_us=%.1f" % (sum(v) / len(v), min(v)))
