"""Developer probe: cProfile of tools/bench_configs.py entries (host time around the kernels)."""
import cProfile, pstats, sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import bench_configs as bc
which = sys.argv[1]
fn = {"c3": bc.c3, "c5p": bc.c5_pipeline, "c5": bc.c5_parts, "f4": bc.f4_scores}[which]
pr = cProfile.Profile(); pr.enable(); out = fn(); pr.disable()
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items() if not isinstance(v, (list, dict))})
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
