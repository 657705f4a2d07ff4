"""GPU: bench.py's output contract on a small workload (one JSON line with roofline + cpu_baseline), and a 2-rank rehearsal of
its N > 1 path (the sharded partial-tree group-by) with both ranks on the test box's single GPU over gloo."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline", "cpu_baseline")


def _last_json(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_contract():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "3e6", "--keys", "5e3", "--steps", "2", "--warmup", "1",
                        "--cpu-sample-rows", "1e6"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    assert all(k in d for k in KEYS)
    assert d["n_gpus"] == 1 and d["unit"] == "Grows/s" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["value"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # frac prices the WHOLE step: 16 B/row x rows / ms_per_step / peak
    assert abs(r["achieved"] - 16.0 * 3e6 / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    dk = r["dominant_kernel"]
    assert dk["kernel"] and dk["avg_launch_ms"] > 0 and dk["frac"] >= r["frac"] and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= c["threads"] >= 1 and c["gpu_matches_oracle_bit_exact"] is True
    if "arrow_matches_oracle_bit_exact" in c:  # Arrow C++ itself timed (the pyarrow wheel's libarrow is on this box)
        assert c["arrow_matches_oracle_bit_exact"] is True and "Arrow C++" in c["engine"]
    assert all(v for v in d["check"].values() if isinstance(v, bool))
    ch = d["chain_check"]
    assert ch["gpu_matches_oracle_bit_exact"] is True and ch["same_path_as_timed_step"] is True and ch["plan"]["layout"] in ("fused", "full")
    assert d["general_keys_hash_path"]["ms"] > 0 and d["general_keys_hash_path"]["plan"]["slots"].startswith("hash")
    sec = d["secondary"]
    t3 = sec["groupby_reference_api_three_calls"]
    assert t3["ms"] > 0 and t3["ratio_to_fused_call"] < 1.6  # (small workload: launch overheads weigh more than at 1e9 rows)
    for k in ("groupby_min_max", "groupby_count", "groupby_int64_sum"):  # the order-free kinds skip the value sort (gb_acc.hpp)
        assert sec[k]["plan"]["reducer"] == "lds_acc", (k, sec[k])
    for k in ("groupby_min_max", "groupby_count", "groupby_int64_sum", "groupby_general_keys_hash_path", "groupby_5pct_null_values", "groupby_one_hot_key_5pct", "C1_add_f64[1e+06]", "C2_filter_8cols+index", "C2_take_8cols+index",
              "C5_resample_1min_mean", "a12_round_temporal_minute", "a12_downsample_1T_mean", "groupby_sorted_keys", "8f3_argsort_f64"):
        assert sec[k]["ms"] > 0 and 0 < sec[k]["frac"] < 1, k


def test_bench_two_rank_rehearsal():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PDX_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "4e6", "--keys", "2e4", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["rows_per_gpu"] == 2_000_000 and d["cpu_baseline"] is None
    assert d["check"]["groups"] == 20_000 and all(v for v in d["check"].values() if isinstance(v, bool))
    assert d["config"]["path"] == "sharded-c-abi"  # pdx_dist_* inside the library; the custom transport rides on the gloo group here
    assert d["check"]["c_abi_matches_torch_orchestration_all_ranks"] is True  # both orchestrations, same shards, bit for bit


def test_bench_two_ranks_without_a_launcher():
    """`python bench.py --gpus 2` as a plain command (no WORLD_SIZE in the environment): bench.py starts its two ranks itself as fresh
    child processes and rank 0 prints the one JSON line (gloo rehearsal: both ranks share this box's GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PDX_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "4e6", "--keys", "2e4", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["config"]["rows_per_gpu"] == 2_000_000 and d["config"]["path"] == "sharded-c-abi"
    assert d["check"]["groups"] == 20_000 and all(v for v in d["check"].values() if isinstance(v, bool))
