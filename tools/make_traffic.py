"""profiles/traffic_<tag>.json from the FETCH_SIZE / WRITE_SIZE passes of tools/prof_cmd.sh over bench.py:
HBM bytes per launch of the two layer-0 products of the Deep-TICA step (the kernels bench.py's roofline picks from),
corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts half the bytes of wide streaming reads,
WRITE_SIZE is exact, both in KiB).
usage: make_traffic.py gpurun_out/prof_<tag> rows_per_launch gemm_mode out.json [more `prof_dir rows mode` triples ...]"""
import json
import sys

args = sys.argv[1:]
out = args[-1]
triples = [args[i:i + 3] for i in range(0, len(args) - 1, 3)]
tables = []
for d, rows, mode in triples:
    rows = int(rows)
    p3 = json.load(open(f"{d}/p3.summary.json"))
    p4 = json.load(open(f"{d}/p4.summary.json"))
    tag = "s," if mode == "split" else ","   # Cfg....b2s = split arithmetic

    def pick(table, gemm, epi):
        # the launch of this kind with the largest total time (layer 0 dominates its kind)
        best = None
        for k, v in table.items():
            if f"gemm_kernel<{gemm}, Cfg" in k and epi in k and (("b2s," in k) == (mode == "split")):
                if best is None or v["avg_us"] * v["dispatches"] > best[1]["avg_us"] * best[1]["dispatches"]:
                    best = (k, v)
        if best is None:
            raise KeyError((gemm, epi))
        return best

    res = {}
    for name, (gemm, epi) in {"layer0.fwd": (0, "EpiBiasAct>"), "layer0.wgrad": (2, "EpiSlab>")}.items():
        (k3, f), (k4, w) = pick(p3, gemm, epi), pick(p4, gemm, epi)
        res[name] = {
            "kernel": k3, "fetch_size_kib": f["FETCH_SIZE"], "write_size_kib": w["WRITE_SIZE"],
            "hbm_bytes_per_launch": (2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024,
            "avg_us_in_fetch_pass": f["avg_us"],
            # no clock field: GRBM_GUI_ACTIVE / 8 / dispatch time reads HIGH on dispatches shorter than ~0.3 ms (the counter
            # window is wider than the dispatch; MI355X_MICROARCH.md, DVFS give-back) -- round 3 printed 2.76-2.80 "GHz" for
            # 24 us kernels on a 2.4 GHz part.  The in-kernel clock comes from s_memtime / s_memrealtime stamps (tools/gemm_bench).
        }
    tables.append({
        "source": f"rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE / WRITE_SIZE passes (tools/prof_cmd.sh) of bench.py, {d}",
        "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM section)",
        "rows_per_launch": rows, "gemm_mode": mode, "kernels": res})
json.dump(tables, open(out, "w"), indent=1)
print(json.dumps(tables, indent=1))
