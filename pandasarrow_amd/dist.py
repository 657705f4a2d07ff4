"""Row-range sharded group-by across the GPUs of one node (one process per GPU, torch.distributed over RCCL/xGMI).

The reference has no distributed code; this is the multi-GPU form of
``df.group_by(key).{sum,mean,min,max,count}(col)`` (src/group_by.h:22-299, src/pd_core_macros.h:5-147) whose result is
bit-identical to the single-GPU / reference result:

  1. local hash group-by of the shard's keys  -> local uniques + first rows (global row index = row_offset + local)
  2. all-gather(v) of the local uniques in rank order; every rank dedupes the concatenation keeping the FIRST occurrence,
     which is exactly the reference's first-occurrence group order over the whole column  -> global group ids
  3. groups are owned by contiguous global-id ranges; every row is routed to its group's owner with ONE all-to-all(v)
     of (global id, value) pairs.  Shards are row ranges in rank order, so the received stream (source-major) is in
     global row order and a stable local group-by reproduces Arrow's per-group pairwise sum in row order.  A plain
     reduce-by-key of per-shard partial sums would NOT be bit-exact for fp64 (hard part 1).
  4. owners aggregate, place results by global id and all-gather(v) them: concat in rank order == global id order.

All compute is delegated to an ``engine``; the product engine is ``HipEngine`` (the C ABI).  There is no default CPU
engine: tests inject an oracle-backed engine to exercise this orchestration over ``gloo`` without a GPU.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _lib as L


# ---------------------------------------------------------------- optional stage timing (PDX_DIST_TIMING=1): syncs per stage
import os as _os
import time as _time

TIMING = {}


class _Stage:
    def __init__(self, name):
        self.name = name
        self.on = _os.environ.get("PDX_DIST_TIMING") == "1"

    def __enter__(self):
        if self.on:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            self.t0 = _time.perf_counter()

    def __exit__(self, *a):
        if self.on:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            TIMING[self.name] = TIMING.get(self.name, 0.0) + (_time.perf_counter() - self.t0) * 1e3


# ---------------------------------------------------------------- collectives (variable-length helpers)
def _world():
    return (dist.get_world_size(), dist.get_rank()) if dist.is_initialized() else (1, 0)


def all_gather_sizes(n, device):
    W, _ = _world()
    if W == 1:
        return [int(n)]
    t = torch.tensor([int(n)], dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(W)]
    dist.all_gather(out, t)
    return [int(x.item()) for x in out]


def all_gather_v(t: torch.Tensor, sizes):
    """Concatenation of every rank's 1-D tensor in rank order (all-gatherv: pad to the largest, gather, trim)."""
    W, _ = _world()
    if W == 1:
        return t.clone()
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
    pad[: t.numel()] = t
    out = [torch.empty_like(pad) for _ in range(W)]
    dist.all_gather(out, pad)
    return torch.cat([o[:s] for o, s in zip(out, sizes)])


def all_to_all_v(chunks):
    """chunks[d] goes to rank d; returns the list received from every source, in source (rank) order."""
    W, r = _world()
    if W == 1:
        return [chunks[0]]
    dev, dt = chunks[0].device, chunks[0].dtype
    send_counts = torch.tensor([c.numel() for c in chunks], dtype=torch.int64, device=dev)
    recv_counts = torch.empty_like(send_counts)
    if dist.get_backend() == "gloo":  # gloo has no all-to-all: emulate with W all-gathers (CPU tests only)
        all_counts = [torch.empty_like(send_counts) for _ in range(W)]
        dist.all_gather(all_counts, send_counts)
        recv = []
        for src in range(W):
            sizes = [int(all_counts[src][d].item()) for d in range(W)]
            buf = torch.cat(chunks) if src == r else torch.empty(sum(sizes), dtype=dt, device=dev)
            dist.broadcast(buf, src)
            off = sum(sizes[:r])
            recv.append(buf[off:off + sizes[r]].clone())
        return recv
    dist.all_to_all_single(recv_counts, send_counts)
    rc = [int(x) for x in recv_counts.tolist()]
    sc = [int(x) for x in send_counts.tolist()]
    out = torch.empty(sum(rc), dtype=dt, device=dev)
    dist.all_to_all_single(out, torch.cat(chunks), output_split_sizes=rc, input_split_sizes=sc)
    return list(torch.split(out, rc))


# ---------------------------------------------------------------- the product engine: HIP kernels through the C ABI
class HipEngine:
    """Engine over pandasarrow_amd.column (device-resident Arrow-layout columns, libpdx_hip.so)."""

    def __init__(self):
        from . import column as K

        L.load()  # fail loudly without the HIP library
        self.K = K
        self.device = K._device()

    # columns <-> tensors (communication buffers)
    def col(self, t: torch.Tensor, dtype, ok: torch.Tensor | None = None):
        K = self.K
        vb = None
        if ok is not None and not bool(ok.all()):
            import numpy as np

            vb = torch.from_numpy(K.pack_bits_host(ok.cpu().numpy().astype(bool))).to(self.device)
        vals = t if t.numel() else torch.zeros(1, dtype=t.dtype, device=self.device)
        return K.Column(dtype, t.numel(), vals.contiguous(), vb)

    def values(self, col) -> torch.Tensor:
        return col.values[col.offset:col.offset + col.length]

    def group(self, key_col):
        return self.K.GroupByHandle.create(key_col)

    def unique_keys(self, gb):
        c = gb.unique_keys()
        vals, ok = c.to_numpy()
        return self.values(c), torch.from_numpy(ok).to(self.device)

    def first_rows(self, gb):
        return gb.first_rows()

    def group_ids(self, gb):
        return gb.group_ids().to(torch.int64)

    def map_ids(self, gb, mapping: torch.Tensor):
        return gb.map_ids(mapping.contiguous())

    def agg(self, gb, values_col, kinds):
        return gb.agg(values_col, kinds)

    def select_eq(self, cols, by_col, value):
        """rows of `cols` where by_col == value (stable)."""
        mask = self.K.compare(L.EQ, by_col, int(value))
        return self.K.filter(cols, mask, emit_null=False)

    def place(self, cols, positions: torch.Tensor, n):
        """dense columns of n rows with cols[c][j] stored at positions[j]."""
        outs = [self.K.Column.empty(c.dtype, n, with_validity=c.has_nulls()) for c in cols]
        if n:
            self.K.scatter(cols, self.col(positions, L.INT64), outs)
        return outs

    def dtype_of(self, col):
        return col.dtype

    def valid_bools(self, col):
        if not col.has_nulls():
            return None
        return torch.from_numpy(col.to_numpy()[1]).to(self.device)

    # ---- partial-tree exchange primitives (exact fp64 sum without shipping rows)
    def group_values(self, gb, values_col):
        return self.K.GroupedValues(gb, values_col)

    def grouped_counts(self, gv):
        return gv.counts()

    def partial_records(self, gv, prefix: torch.Tensor, gid_map: torch.Tensor, order: torch.Tensor | None = None):
        gv.partial_plan(prefix, order)
        return gv.partial_fill(gid_map)

    def replay(self, rec_key, rec_val, gid_lo, n_own):
        return self.K.replay_partials(rec_key, rec_val, gid_lo, n_own)

    def select_tensor_eq(self, tensors, by: torch.Tensor, value):
        """stable selection of rows of 1-D int64/float64 tensors where by == value (through the filter kernels)."""
        by_col = self.col(by, L.INT64)
        cols = [self.col(t, L.FLOAT64 if t.dtype == torch.float64 else L.INT64) for t in tensors]
        return [self.values(c) for c in self.select_eq(cols, by_col, value)]


# ---------------------------------------------------------------- sharded group-by
def groupby_agg_sharded(engine, keys, vals, kinds, row_offset=0):
    """Every rank passes its row-range shard (engine columns).  Returns a dict with the FULL result on every rank:
    keys/keys_ok (G unique keys, first-occurrence order), first_rows, outs (one (values tensor, valid|None) per kind)."""
    W, r = _world()
    dev = engine.device
    gb = engine.group(keys)
    uk, uok = engine.unique_keys(gb)
    fr = engine.first_rows(gb) + int(row_offset)
    Gl = int(uk.numel())
    # 2. global dictionary
    sizes = all_gather_sizes(Gl, dev)
    cat_keys = all_gather_v(uk, sizes)
    cat_ok = all_gather_v(uok.to(torch.uint8), sizes).to(torch.bool)
    cat_first = all_gather_v(fr, sizes)
    gb_cat = engine.group(engine.col(cat_keys, L.INT64, cat_ok))
    glob_keys, glob_ok = engine.unique_keys(gb_cat)
    G = int(glob_keys.numel())
    gid_cat = engine.group_ids(gb_cat)
    glob_first = cat_first[engine.first_rows(gb_cat)] if G else cat_first[:0]
    off = sum(sizes[:r])
    my_map = gid_cat[off:off + Gl].contiguous()  # local group id -> global group id
    # 3. route rows to the owners of contiguous global-id ranges
    bounds = [G * d // W for d in range(W + 1)]
    owner_l = torch.bucketize(my_map, torch.tensor(bounds[1:], dtype=torch.int64, device=dev), right=True)
    row_gid = engine.map_ids(gb, my_map)
    if W == 1:
        recv_gid, recv_val = row_gid, vals
    else:
        row_owner = engine.map_ids(gb, owner_l.contiguous())
        vok = engine.valid_bools(vals)  # value nulls travel as a third routed column (only when present anywhere)
        has_nulls = any(all_gather_sizes(0 if vok is None else 1, dev))
        cols = [row_gid, vals]
        if has_nulls:
            n_loc = int(engine.values(row_gid).numel())
            vok = torch.ones(n_loc, dtype=torch.bool, device=dev) if vok is None else vok
            cols.append(engine.col(vok.to(torch.int64), L.INT64))
        send = [[] for _ in cols]
        for d in range(W):
            for j, c in enumerate(engine.select_eq(cols, row_owner, d)):
                send[j].append(engine.values(c))
        recv = [torch.cat(all_to_all_v(chunks)) for chunks in send]
        recv_gid = engine.col(recv[0], L.INT64)
        recv_val = engine.col(recv[1], engine.dtype_of(vals), recv[2].to(torch.bool) if has_nulls else None)
    # 4. owners aggregate (stable: source-major order == global row order) and place results by global id
    gb2 = engine.group(recv_gid)
    outs2 = engine.agg(gb2, recv_val, list(kinds))
    u2, _ = engine.unique_keys(gb2)
    n_own = bounds[r + 1] - bounds[r]
    dense = engine.place(outs2, (u2 - bounds[r]).contiguous(), n_own)
    own_sizes = [bounds[d + 1] - bounds[d] for d in range(W)]
    outs = []
    for c in dense:
        v = all_gather_v(engine.values(c), own_sizes)
        okb = engine.valid_bools(c)
        any_nulls = all_gather_sizes(0 if okb is None else 1, dev)
        ok = None
        if any(any_nulls):
            okb = torch.ones(n_own, dtype=torch.bool, device=dev) if okb is None else okb
            ok = all_gather_v(okb.to(torch.uint8), own_sizes).to(torch.bool)
        outs.append((v, ok))
    return {"G": G, "keys": glob_keys, "keys_ok": glob_ok, "first_rows": glob_first, "outs": outs, "kinds": list(kinds)}


def groupby_sum_mean_count_sharded(engine, keys, vals, row_offset=0):
    """Headline query (sum, mean, count of a non-null float64 column) with the PARTIAL-TREE exchange: instead of routing every
    row to its group's owner, each rank ships the boundary-leaf fragments and the aligned subtree nodes of its share of every
    group (include/pdx/abi.h, "exact multi-GPU fp64 sum").  Same result dict as groupby_agg_sharded with kinds [SUM, MEAN, COUNT]."""
    W, r = _world()
    dev = engine.device
    with _Stage("1_local_group"):
        gb = engine.group(keys)
        uk, uok = engine.unique_keys(gb)
        fr = engine.first_rows(gb) + int(row_offset)
        Gl = int(uk.numel())
    with _Stage("2_dictionary"):
        sizes = all_gather_sizes(Gl, dev)
        cat_keys = all_gather_v(uk, sizes)
        cat_ok = all_gather_v(uok.to(torch.uint8), sizes).to(torch.bool)
        cat_first = all_gather_v(fr, sizes)
        if W == 1:
            glob_keys, glob_ok, glob_first, G = uk, uok, fr, Gl
            my_map = torch.arange(Gl, dtype=torch.int64, device=dev)
        else:
            gb_cat = engine.group(engine.col(cat_keys, L.INT64, cat_ok))
            glob_keys, glob_ok = engine.unique_keys(gb_cat)
            G = int(glob_keys.numel())
            gid_cat = engine.group_ids(gb_cat)
            glob_first = cat_first[engine.first_rows(gb_cat)] if G else cat_first[:0]
            off = sum(sizes[:r])
            my_map = gid_cat[off:off + Gl].contiguous()
    # rows per (global group, rank): dense count vectors, all-gathered; prefix over lower ranks
    with _Stage("3_group_values"):
        gv = engine.group_values(gb, vals)
        cnt_local = engine.grouped_counts(gv)
    with _Stage("4_counts_exchange"):
        dense = torch.zeros(max(G, 1), dtype=torch.int64, device=dev)
        dense[my_map] = cnt_local
        if W > 1:
            allc = [torch.empty_like(dense) for _ in range(W)]
            dist.all_gather(allc, dense)
            allc = torch.stack(allc)  # [W, G]
            prefix_g = allc[:r].sum(dim=0) if r > 0 else torch.zeros_like(dense)
            count_g = allc.sum(dim=0)
        else:
            prefix_g, count_g = torch.zeros_like(dense), dense
    with _Stage("5_partial_records"):
        # records are emitted group by group in GLOBAL-id order, so they leave the kernel already partitioned by owner rank
        order = torch.argsort(my_map) if W > 1 else None
        rec_key, rec_val = engine.partial_records(gv, prefix_g[my_map].contiguous(), my_map, order)
    # owners of contiguous global-id ranges: one all-to-all(v) of contiguous slices
    bounds = [G * d // W for d in range(W + 1)]
    with _Stage("6_all_to_all"):
        if W > 1:
            cuts = torch.searchsorted(rec_key, torch.tensor([b * 64 for b in bounds], dtype=torch.int64, device=dev)).tolist()
            rk = torch.cat(all_to_all_v([rec_key[cuts[d]:cuts[d + 1]] for d in range(W)]))
            rv = torch.cat(all_to_all_v([rec_val[cuts[d]:cuts[d + 1]] for d in range(W)]))
        else:
            rk, rv = rec_key, rec_val
    with _Stage("7_replay"):
        n_own = bounds[r + 1] - bounds[r]
        sums_own = engine.replay(rk, rv, bounds[r], n_own)
    with _Stage("8_gather_results"):
        own_sizes = [bounds[d + 1] - bounds[d] for d in range(W)]
        sums = all_gather_v(sums_own, own_sizes)
        counts = count_g[:G]
        means = sums / counts.to(torch.float64)
    return {"G": G, "keys": glob_keys, "keys_ok": glob_ok, "first_rows": glob_first, "kinds": [L.AGG_SUM, L.AGG_MEAN, L.AGG_COUNT],
            "outs": [(sums, None), (means, None), (counts, None)], "records": int(rec_key.numel())}


def check_result(res, n_total):
    """Size-independent properties of a sharded sum/mean/count result (bench.py)."""
    out = {"groups": int(res["G"])}
    kinds = res["kinds"]
    if L.AGG_COUNT in kinds:
        cnt = res["outs"][kinds.index(L.AGG_COUNT)][0]
        out["counts_sum_to_rows"] = int(cnt.sum().item()) == int(n_total)
        if L.AGG_SUM in kinds and L.AGG_MEAN in kinds:
            s = res["outs"][kinds.index(L.AGG_SUM)][0]
            m = res["outs"][kinds.index(L.AGG_MEAN)][0]
            out["mean_is_sum_over_count"] = bool(torch.equal(m, s / cnt.to(torch.float64)))
    fr = res["first_rows"]
    out["first_occurrence_order"] = bool((fr[1:] > fr[:-1]).all().item()) if fr.numel() > 1 else True
    return out
