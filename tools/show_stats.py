import csv, glob, sys
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/prof*/*/*kernel_stats.csv"))[-1]
print(f)
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>7s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={float(r['Percentage']):6.2f}")
