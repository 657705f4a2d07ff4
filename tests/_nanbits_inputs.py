"""Seeded inputs of the NaN-bits golden cases (tests/golden/nan_bits_golden.npz holds only Arrow's OUTPUTS; the inputs are regenerated
from these seeds by the generator oracle/gen_golden_nanbits.py and by the tests, numpy only)."""
import numpy as np

WHOLE_LENGTHS = (1, 15, 16, 17, 31, 33, 100, 257, 1000, 4096, 5000, 70_001, 300_000)
WHOLE_MIX = ((0.5, 0.0), (0.02, 0.0), (0.0, 0.3), (0.05, 0.05), (None, 0.0))  # (share of NaNs or None = two per array, share of +-inf)
GROUP_SHAPES = ((2_000, 7), (50_000, 40), (200_000, 3), (400_000, 5000), (1_500_000, 20_000))
GROUP_MIX = ((0.01, 0.0), (0.0005, 0.0005), (0.2, 0.02))


def special_values(rng, n, p_nan, p_inf):
    """random magnitudes with quiet and signalling NaNs of both signs and distinct payloads, and +-inf"""
    v = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4, n)
    m = rng.random(n) < p_nan
    k = int(m.sum())
    bits = (rng.integers(1, 2**51, k).astype(np.uint64) | np.uint64(0x7FF0000000000000) | (rng.integers(0, 2, k).astype(np.uint64) << np.uint64(63))
            | (rng.integers(0, 2, k).astype(np.uint64) << np.uint64(51)))
    v[m] = bits.view(np.float64)
    mi = rng.random(n) < p_inf
    v[mi] = rng.choice(np.array([np.inf, -np.inf]), int(mi.sum()))
    return v


def whole_cases():
    """-> [(name, values, valid | None)]"""
    out, ci = [], 0
    for n in WHOLE_LENGTHS:
        for p_nan, p_inf in WHOLE_MIX:
            for nulls in (False, True):
                rng = np.random.default_rng(770000 + ci)
                v = special_values(rng, n, min(2.0 / max(n, 2) if p_nan is None else p_nan, 1.0), p_inf)
                valid = (rng.random(n) > 0.15) if nulls else None
                out.append((f"w{ci}", v, valid))
                ci += 1
    return out


def group_cases():
    """-> [(name, keys, values, valid | None)]"""
    out, gi = [], 0
    for n, nk in GROUP_SHAPES:
        for p_nan, p_inf in GROUP_MIX:
            for nulls in (False, True):
                rng = np.random.default_rng(880000 + gi)
                keys = rng.integers(0, nk, n).astype(np.int64)
                v = special_values(rng, n, p_nan, p_inf)
                valid = (rng.random(n) > 0.1) if nulls else None
                out.append((f"g{gi}", keys, v, valid))
                gi += 1
    return out
