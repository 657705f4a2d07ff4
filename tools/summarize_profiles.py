#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (written by tools/collect_profiles.sh on the GPU box) into tracked files under profiles/:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `bench.py`
  profiles/<tag>_pmc_traffic.json   per-kernel HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (KiB units; FETCH_SIZE doubled:
                                    on gfx950 it reports half of a wide coalesced read stream -- MI355X_MICROARCH.md, HBM section)
  profiles/<tag>_bench.json         the bench line of the same build
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
def newest(pattern):
    """gpurun merges each call's files into gpurun_out/ without deleting older ones: take the latest run's file"""
    return max(glob.glob(pattern), key=os.path.getmtime)


stats = [newest(os.path.join(src, "kt", "*", "*kernel_stats.csv"))]
shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
if glob.glob(os.path.join(src, "kt_secondary", "*", "*kernel_stats.csv")):  # the default run: headline + the secondary configs
    shutil.copy(newest(os.path.join(src, "kt_secondary", "*", "*kernel_stats.csv")), os.path.join(dst, f"{tag}_kernel_stats_with_secondary.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))
if os.path.exists(os.path.join(src, "bench_general_keys.json")):
    shutil.copy(os.path.join(src, "bench_general_keys.json"), os.path.join(dst, f"{tag}_bench_general_keys.json"))


def short(name):
    m = re.search(r"pdx::(k_[a-z_0-9]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


sys.path.insert(0, root)
import bench  # noqa: E402  (source_hash: ties these counters to the library sources they were measured on)

# bench.py tags -> kernels (large-payload instantiations only)
TAGS = {"radix_scatter": r"k_radix_scatter(_occ4)?<\d+, unsigned long", "radix_hist": r"k_radix_hist<", "dense_slots": r"k_dense_slots_tail", "hash_insert": r"k_hash_insert",
        "seg_reduce": r"k_seg_reduce<", "key_minmax": r"k_minmax_partial<long long>", "fused_last_digit_reduce": r"k_flr_(reduce|wave)<",
        "hash_probe_lds": r"k_hash_probe_lds", "hash_bucket_hist": r"k_hash_bucket_hist", "acc_reduce": r"k_acc<", "acc_partition_keys": r"k_acc_part_keys<"}


def traffic_file(prefix, slots, out_name):
    """pmc_<prefix>FETCH_SIZE / pmc_<prefix>WRITE_SIZE -> profiles/<out_name>: HBM bytes per launch and per step of ONE plan (slots)"""
    traffic = collections.defaultdict(lambda: {"launches": 0, "fetch_kib": 0.0, "write_kib": 0.0})
    for counter, field in (("FETCH_SIZE", "fetch_kib"), ("WRITE_SIZE", "write_kib")):
        f = newest(os.path.join(src, f"pmc_{prefix}{counter}", "*", "*counter_collection.csv"))
        seen = collections.Counter()
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            traffic[k][field] += float(r["Counter_Value"])
            seen[k] += 1
        for k, n in seen.items():
            traffic[k]["launches"] = max(traffic[k]["launches"], n)
    out = {}
    for k, t in traffic.items():
        n = max(t["launches"], 1)
        rd, wr = 2.0 * t["fetch_kib"] * 1024 / n, t["write_kib"] * 1024 / n
        if rd + wr < 1e8:
            continue
        out[k] = {"launches_in_profiled_run": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
    by_tag = {}
    for tg, pat in TAGS.items():
        ks = [(k, v) for k, v in out.items() if re.search(pat, k)]
        if ks:
            n = sum(v["launches_in_profiled_run"] for _, v in ks)
            by_tag[tg] = sum(v["hbm_bytes_per_launch"] * v["launches_in_profiled_run"] for _, v in ks) / n
    # one step's HBM traffic: every kernel of the profiled run except the input generators, over its (warmup + timed) steps
    steps_in_run = 2
    step_bytes = sum(v["hbm_bytes_per_launch"] * v["launches_in_profiled_run"] for k, v in out.items() if "synth" not in k) / steps_in_run
    json.dump({"by_bench_tag_hbm_bytes_per_launch": by_tag, "step_hbm_bytes": step_bytes, "source_hash": bench.source_hash(), "rows": 1000000000, "n_gpus": 1, "slots": slots,
               "note": "bytes per launch at 1e9 rows / 1e6 keys, 1 GPU; read = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB; slots = the key -> slot "
                       "plan of the profiled step (bench.py attaches the file to a run only when its plan, rows, GPU count and sources match)",
               "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]))},
              open(os.path.join(dst, out_name), "w"), indent=1)
    return out


out = traffic_file("", "dense", f"{tag}_pmc_traffic.json")
if glob.glob(os.path.join(src, "pmc_general_FETCH_SIZE", "*", "*counter_collection.csv")):
    traffic_file("general_", "hash_lds", f"{tag}_pmc_traffic_general.json")
if glob.glob(os.path.join(src, "kt_general", "*", "*kernel_stats.csv")):
    shutil.copy(newest(os.path.join(src, "kt_general", "*", "*kernel_stats.csv")), os.path.join(dst, f"{tag}_kernel_stats_general_keys.csv"))
print(open(os.path.join(dst, f"{tag}_bench.json")).read()[:600])
print(json.dumps(out, indent=1)[:1500])
