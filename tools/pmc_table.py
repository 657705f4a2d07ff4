#!/usr/bin/env python3
"""Per-kernel sums of the counters collected by tools/pmc_passes.sh (rocprofv3 counter_collection.csv files)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
tab = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].split("(")[0][:150]
            tab[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
names = sorted({c for k in tab for c in tab[k]})
for k in sorted(tab, key=lambda k: -tab[k].get("SQ_WAVE_CYCLES", tab[k].get(names[0], 0))):
    print(k)
    for c in names:
        if c in tab[k]:
            print(f"    {c:36s} {tab[k][c]:16.0f}   ({cnt[k][c]} dispatches)")
