for rep in 1 2; do for tr in 0 16; do
echo "== DCV_SNET_TR=$tr (rep $rep)"
DCV_SNET_TR=$tr python - <<'PY'
import os, sys, json
sys.argv=["bench.py"]
import torch
sys.path.insert(0,".")
import bench
r=bench.run_calculator_fit(10)
print(" ".join(f"{x['cv']}/{x['batch']}:{x['us_per_training_step_all_inclusive']:.1f}" for x in r["runs"]))
PY
done; done
