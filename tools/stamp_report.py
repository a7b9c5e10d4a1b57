"""Per-CU timeline report of the DCV_STAMP dumps written by tools/gemm_bench (developer tool)."""
import sys
import numpy as np

for path in sys.argv[1:]:
    h = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
    n = len(h)
    rt0 = h[:, 5].astype(np.int64)
    rt1 = h[:, 6].astype(np.int64)
    base = rt0.min()
    s = (rt0 - base) * 0.01
    e = (rt1 - base) * 0.01
    hw = h[:, 7] & 0xFFFFFFFF
    xcc = (h[:, 7] >> 32) & 0xF
    cuid = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 10 + ((hw >> 8) & 0xF)
    ml = (h[:, 2] - h[:, 1]).astype(np.int64)
    pro = (h[:, 1] - h[:, 0]).astype(np.int64)
    epi = (h[:, 4] - h[:, 2]).astype(np.int64)
    clk = (ml / np.maximum(rt1 - rt0, 1) * 0.1).mean()
    # order of each WG on its CU
    order = np.zeros(n, dtype=np.int64)
    for c in np.unique(cuid):
        idx = np.where(cuid == c)[0]
        order[idx[np.argsort(s[idx], kind="stable")]] = np.arange(len(idx))
    print(f"{path}: {n} WGs on {len(np.unique(cuid))} CUs, clock {clk:.2f} GHz, last mainloop end {e.max():.1f} us")
    for k in range(int(order.max()) + 1):
        m = order == k
        print(f"   WG #{k} on its CU: start {s[m].mean():7.1f} us  mainloop {ml[m].mean():8.0f} cyc ({(e - s)[m].mean():5.1f} us)"
              f"  prologue {pro[m].mean():6.0f}  epilogue {epi[m].mean():6.0f}")
