"""Developer probe: bench.run_calculator_fit twice in one process (the second run is warm)."""
import sys
sys.argv = ["bench.py"]
sys.path.insert(0, ".")
import bench
for rep in range(2):
    r = bench.run_calculator_fit(10)
    print(" ".join("%s/%d:%.1f" % (x["cv"], x["batch"], x["us_per_training_step_all_inclusive"]) for x in r["runs"]))
