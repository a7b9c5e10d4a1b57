"""Print the top rows of a rocprofv3 kernel_stats.csv (names contain commas: proper CSV parse)."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
