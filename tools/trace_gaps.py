"""Per-step timeline from a rocprofv3 --kernel-trace CSV dir: busy time, idle gaps between consecutive
dcv kernels, per-kernel share; steady-state window = the last `nsteps` reduce_grads_kernel launches (one per optimiser step)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
adam = [i for i, r in enumerate(rows) if "reduce_grads" in r[2]]   # one per optimiser step (the update is fused into it)
lo, hi = adam[-nsteps - 1], adam[-1]
win = rows[lo + 1: hi + 1]
span = win[-1][1] - win[0][0]
busy = sum(e - s for s, e, _ in win)
gaps = sum(max(0, win[i + 1][0] - win[i][1]) for i in range(len(win) - 1))
per = defaultdict(lambda: [0, 0])
for s, e, n in win:
    n = re.sub(r"dcv::TileCfg<(\d), (\d), (\d), (\d), (\d+), (\d), true, 0>", r"Cfg\1\2\3\4k\5b\6s", n)   # trailing 0: no pre-split (plane) operands
    n = re.sub(r"dcv::TileCfg<(\d), (\d), (\d), (\d), (\d+), (\d), false, 0>", r"Cfg\1\2\3\4k\5b\6", n)
    n = re.sub(r"dcv::TileCfg<(\d), (\d), (\d), (\d), (\d+), (\d), true>", r"Cfg\1\2\3\4k\5b\6s", n)
    n = re.sub(r"dcv::TileCfg<(\d), (\d), (\d), (\d), (\d+), (\d), false>", r"Cfg\1\2\3\4k\5b\6", n)
    n = re.sub(r"dcv::TileCfg<(\d), (\d), (\d), (\d), (\d+), (\d)>", r"Cfg\1\2\3\4k\5b\6", n)
    n = re.sub(r"\(.*", "", n).replace("void ", "")[:80]
    per[n][0] += e - s
    per[n][1] += 1
print(f"steps={nsteps} span/step={span / nsteps / 1e3:.1f} us  busy/step={busy / nsteps / 1e3:.1f} us  gaps/step={gaps / nsteps / 1e3:.1f} us  launches/step={len(win) / nsteps:.1f}")
for n, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    print(f"  {t / nsteps / 1e3:8.1f} us/step  {c / nsteps:5.1f} launches/step  {n}")
