"""Idle gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV (one process, one stream): where the host round trips of a
step sit.  usage: python tools/step_gaps.py <dir with *_kernel_trace.csv> [min_gap_us] [all]   (all: also one line per kernel of the last step)"""
import csv
import glob
import sys

root = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last step = everything after the last k_sample_key_range / k_sample_descents launch pair
starts = [i for i, r in enumerate(rows) if "k_sample_descents" in r[2] or ("k_sample_key_range" in r[2] and not (i and "k_sample_descents" in rows[i - 1][2]))]
first = starts[-1] if starts else 0
last = rows[first:]
t0 = last[0][0]
total_gap = 0.0
busy = 0.0
show_all = len(sys.argv) > 3 and sys.argv[3] == "all"
for (s0, e0, n0), (s1, e1, n1) in zip(last, last[1:]):
    gap = (s1 - e0) / 1e3
    busy += (e0 - s0) / 1e3
    if show_all:
        print(f"{(s0 - t0) / 1e3:10.1f} us  run {(e0 - s0) / 1e3:9.1f} us  {n0[:110]}")
    if gap > 0:
        total_gap += gap
    if gap >= min_gap:
        print(f"{(e0 - t0) / 1e3:10.1f} us  gap {gap:8.1f} us  after {n0[:70]}  before {n1[:70]}")
busy += (last[-1][1] - last[-1][0]) / 1e3
print(f"kernels {len(last)}  span {(last[-1][1] - t0) / 1e3:.1f} us  busy {busy:.1f} us  idle {total_gap:.1f} us")
