"""Per-kernel averages of a rocprofv3 --pmc CSV: python tools/pmc_kernels.py <counter_collection.csv>"""
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:44]
    a = agg[k][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
for k, cs in sorted(agg.items()):
    print(f"{k:44s} " + "  ".join(f"{c}: n={n} avg={tot/n:.4g}" for c, (n, tot) in sorted(cs.items())))
