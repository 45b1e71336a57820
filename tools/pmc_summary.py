"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs per kernel (KiB units -> bytes).
gfx950: FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming read
(MI355X_MICROARCH.md, HBM section): the corrected column doubles it."""
import csv, glob, sys, collections
out = []
for name, pat in (("FETCH_SIZE", "gpurun_out/pmc_fetch/*/*counter_collection.csv"),
                  ("WRITE_SIZE", "gpurun_out/pmc_write/*/*counter_collection.csv")):
    files = glob.glob(pat)
    if not files:
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].split("(")[0][:48]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    for k, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        per = tot / n * 1024.0
        corr = per * (2.0 if name == "FETCH_SIZE" else 1.0)
        out.append(f"{name:10s} {k:48s} launches={n:5d} per_launch={per/1e6:10.2f} MB  gfx950-corrected={corr/1e6:10.2f} MB")
print("\n".join(out))
