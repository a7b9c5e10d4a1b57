import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d["value"] / 1e6, 2), round(d["ms_per_step"], 4), round(d["roofline"]["frac"], 3), "c2", round(d["c2"]["value"] / 1e6, 1),
      "ref_small", [round(r["us_per_step"], 1) for r in d["ref_small"]["runs"]],
      "fit", [round(r["us_per_training_step_all_inclusive"], 1) for r in d["calculator_fit"]["runs"]])
