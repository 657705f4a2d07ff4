"""Per-kernel totals of a rocprofv3 --kernel-trace --stats run.  usage: python tools/kstats.py <dir> [top]"""
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:top]:
    print(f"{float(r['TotalDurationNs']) / 1e6:10.3f} ms  {int(r['Calls']):5d} calls  {float(r['AverageNs']) / 1e3:10.1f} us  {r['Name'][:110]}")
