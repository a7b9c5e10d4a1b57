"""Headline numbers of a bench.py JSON line."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("contract", round(d["value"] / 1e6, 2), "M frames/s", round(d["ms_per_step"], 4), "ms; frac", round(d["roofline"]["frac"], 3))
if d.get("large_batch"): print("large", round(d["large_batch"]["value"] / 1e6, 2), round(d["large_batch"]["ms_per_step"], 4), "frac", round(d["large_batch"]["roofline"]["frac"], 3))
if d.get("other_gemm_mode"): print("other", d["other_gemm_mode"]["gemm_mode"], round(d["other_gemm_mode"]["value"] / 1e6, 2))
if d.get("cpu_baseline"): print("cpu", d["cpu_baseline"]["value"])
