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


def _solo():
    """True when the collectives may be skipped (one rank).  PDX_DIST_FORCE_COLLECTIVES=1 (tests) keeps every collective on the
    wire even at world size 1, so a single-GPU box exercises the RCCL branches (all_gather, all_to_all_single with split sizes)."""
    W, _ = _world()
    return W == 1 and not (dist.is_initialized() and _os.environ.get("PDX_DIST_FORCE_COLLECTIVES") == "1")


def all_gather_sizes(n, device):
    W, _ = _world()
    if _solo():
        return [int(n)]
    t = torch.tensor([int(n)], dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(W)]
    dist.all_gather(out, t)
    return [int(x.item()) for x in out]


def all_gather_v(t: torch.Tensor, sizes):
    """Concatenation of every rank's 1-D tensor in rank order (all-gatherv: pad to the largest, gather, trim)."""
    W, _ = _world()
    if _solo():
        return t.clone()
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
    pad[: t.numel()] = t
    out = [torch.empty_like(pad) for _ in range(W)]
    dist.all_gather(out, pad)
    return torch.cat([o[:s] for o, s in zip(out, sizes)])


def all_gather_v_multi(ts, sizes):
    """all_gather_v of several equally long 8-byte (or bool / uint8) 1-D tensors in ONE collective: the tensors travel as the rows
    of one padded int64 matrix.  Returns the per-tensor concatenations in rank order, in the input dtypes."""
    W, _ = _world()
    if _solo():
        return [t.clone() for t in ts]
    k, n = len(ts), int(ts[0].numel())
    mx = max(max(sizes), 1)
    pad = torch.zeros((k, mx), dtype=torch.int64, device=ts[0].device)
    for i, t in enumerate(ts):
        pad[i, :n] = t.view(torch.int64) if t.dtype == torch.float64 else t.to(torch.int64)
    out = [torch.empty_like(pad) for _ in range(W)]
    dist.all_gather(out, pad)
    res = []
    for i, t in enumerate(ts):
        cat = torch.cat([o[i, :sz] for o, sz in zip(out, sizes)])
        res.append(cat.view(torch.float64) if t.dtype == torch.float64 else cat.to(t.dtype))
    return res


def all_to_all_v_pairs(chunks_a, chunks_b, all_counts=None):
    """all_to_all_v of two aligned lists (chunks_a[d] and chunks_b[d] have the same length, 8-byte dtypes) in ONE exchange: every
    destination receives [a | b] of every source.  Returns (recv_a, recv_b), each the concatenation in source order.
    all_counts (optional): [W][W] python ints, all_counts[s][d] = ELEMENTS of a (== of b) that s sends to d, when the caller has
    already exchanged them -- saves the count exchange and its host sync."""
    W, _ = _world()
    if _solo():
        return chunks_a[0], chunks_b[0]
    da, db = chunks_a[0].dtype, chunks_b[0].dtype
    as64 = lambda t: t.view(torch.int64) if t.dtype == torch.float64 else t.to(torch.int64)
    doubled = None if all_counts is None else [[2 * c for c in row] for row in all_counts]
    recv = all_to_all_v([torch.cat([as64(a), as64(b)]) for a, b in zip(chunks_a, chunks_b)], doubled)
    ra, rb = [], []
    for piece in recv:
        h = piece.numel() // 2
        ra.append(piece[:h])
        rb.append(piece[h:])
    back = lambda t, dt: t.view(torch.float64) if dt == torch.float64 else t.to(dt)
    return back(torch.cat(ra), da), back(torch.cat(rb), db)


def all_to_all_v(chunks, all_counts=None):
    """chunks[d] goes to rank d; returns the list received from every source, in source (rank) order.
    all_counts (optional): [W][W] python ints with all_counts[s][d] = elements s sends to d (already known to every rank)."""
    W, r = _world()
    if _solo():
        return [chunks[0]]
    dev, dt = chunks[0].device, chunks[0].dtype
    sc = [int(c.numel()) for c in chunks]
    if all_counts is not None and dist.get_backend() != "gloo":
        rc = [int(all_counts[s][r]) for s in range(W)]
        out = torch.empty(sum(rc), dtype=dt, device=dev)
        dist.all_to_all_single(out, torch.cat(chunks), output_split_sizes=rc, input_split_sizes=sc)
        return list(torch.split(out, rc))
    send_counts = torch.tensor(sc, dtype=torch.int64, device=dev)
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
    rc = [int(x) for x in recv_counts.tolist()]  # the one host sync of this exchange (send counts are host values already)
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
    # Arrow validity bitmaps <-> bool tensors, on the device (communication buffers carry bools)
    def _pack_bits(self, ok: torch.Tensor) -> torch.Tensor:
        n = ok.numel()
        pad = torch.zeros(((n + 7) // 8 + 16) * 8, dtype=torch.uint8, device=self.device)  # 16 bytes of slack like Column.empty
        pad[:n] = ok.to(torch.uint8)
        w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=self.device)
        return (pad.view(-1, 8) * w).sum(dim=1, dtype=torch.uint8)

    def _unpack_bits(self, bits: torch.Tensor, offset: int, n: int) -> torch.Tensor:
        # one byte of scratch per row (the bytes that hold the window, spread over 8 lanes), not an 8-byte row index
        b0, b1 = offset >> 3, (offset + n + 7) >> 3
        sh = torch.arange(8, dtype=torch.uint8, device=self.device)
        flat = ((bits[b0:b1, None] >> sh) & 1).reshape(-1)
        return flat[offset - 8 * b0: offset - 8 * b0 + n].to(torch.bool)

    def col(self, t: torch.Tensor, dtype, ok: torch.Tensor | None = None):
        K = self.K
        vb = None
        if ok is not None and not bool(ok.all()):
            vb = self._pack_bits(ok)
        vals = t if t.numel() else torch.zeros(1, dtype=t.dtype, device=self.device)
        return K.Column(dtype, t.numel(), vals.contiguous(), vb)

    def values(self, col) -> torch.Tensor:
        return col.values[col.offset:col.offset + col.length]

    def group(self, key_col):
        return self.K.GroupByHandle.create(key_col)

    def unique_keys(self, gb):
        c = gb.unique_keys()
        if c.validity is None or c.null_count == 0:
            return self.values(c), torch.ones(c.length, dtype=torch.bool, device=self.device)
        return self.values(c), self._unpack_bits(c.validity, c.offset, c.length)

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
        return self._unpack_bits(col.validity, col.offset, col.length)

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

    # ---- whole-column aggregates / resample / concat over shards
    def length(self, col):
        return col.length

    def aggregate(self, kind, col):
        return self.K.aggregate(kind, col)

    def to_f64(self, col):
        return self.K.cast_f64(col, checked=False)  # Arrow's mean over int64 input: static_cast<double> per value

    def count_below(self, ts_col, edge, inclusive):
        """rows with ts < edge (<= when inclusive)."""
        as_int = self.K.Column(L.INT64, ts_col.length, ts_col.values, None, ts_col.offset)
        return self.K.filter_count(self.K.compare(L.LE if inclusive else L.LT, as_int, int(edge)))

    def resample(self, ts_col, freq_ns, closed_right, label_right, origin, origin_custom_ns, offset_ns):
        return self.K.GroupByHandle.resample(ts_col, freq_ns, closed_right, label_right, origin, origin_custom_ns, offset_ns)

    def select_tensor_eq(self, tensors, by: torch.Tensor, value):
        """stable selection of rows of 1-D int64/float64 tensors where by == value (through the filter kernels)."""
        by_col = self.col(by, L.INT64)
        cols = [self.col(t, L.FLOAT64 if t.dtype == torch.float64 else L.INT64) for t in tensors]
        return [self.values(c) for c in self.select_eq(cols, by_col, value)]


# ---------------------------------------------------------------- the sharded path behind the C ABI (pdx_dist_*, csrc/dist.hip)
class _RawDeviceBytes:
    """`nbytes` of device memory at `ptr` as a CUDA-array-interface object (the custom transport wraps the library's buffers)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def _unpack_bits_device(bits, n):
    """Arrow validity bitmap (uint8 tensor on the device, LSB first) -> bool tensor of n rows, without leaving the device."""
    b = bits[: (n + 7) // 8].to(torch.uint8)
    shifts = torch.arange(8, dtype=torch.uint8, device=b.device)
    return ((b[:, None] >> shifts[None, :]) & 1).reshape(-1)[:n].to(torch.bool)


class _LazyResult(dict):
    """result dict whose "keys_ok" (the null key's flag as a bool tensor) is unpacked from the key column's validity bitmap on first use:
    the headline step never looks at it, and the unpacking is four small torch launches per step"""

    def __init__(self, items, kcol, G):
        super().__init__(items)
        self._kcol, self._G = kcol, G

    def __missing__(self, name):
        if name != "keys_ok":
            raise KeyError(name)
        from . import column as K

        ok = (_unpack_bits_device(self._kcol.validity, self._G) if self._kcol.validity is not None
              else torch.ones(self._G, dtype=torch.bool, device=K._device()))
        self[name] = ok
        return ok

    def __contains__(self, name):
        return name == "keys_ok" or super().__contains__(name)

    def get(self, name, default=None):
        return self[name] if name in self else default


def _fetch_dist_result(lib, h, key_dtype):
    """pdx_dist_groupby* -> the result dict of the sharded entry points (device tensors); destroys the handle."""
    import ctypes as C

    from . import column as K

    try:
        G = int(lib.pdx_dist_groupby_num_groups(h))
        kcol = K.Column.empty(key_dtype, G, with_validity=True)
        m = kcol.mut()
        dev = K._device()
        first = torch.empty(max(G, 1), dtype=torch.int64, device=dev)
        sums = torch.empty(max(G, 1), dtype=torch.float64, device=dev)
        means = torch.empty(max(G, 1), dtype=torch.float64, device=dev)
        counts = torch.empty(max(G, 1), dtype=torch.int64, device=dev)
        L.check(lib.pdx_dist_groupby_fetch(h, C.byref(m), first.data_ptr(), sums.data_ptr(), means.data_ptr(), counts.data_ptr(), K._stream()))
        kcol._adopt(m)
        records = int(lib.pdx_dist_groupby_num_records(h))
    finally:
        lib.pdx_dist_groupby_destroy(h)
    # (the null key's flag is unpacked on the device, and only when somebody asks for it)
    return _LazyResult({"G": G, "keys": kcol.values[:G], "first_rows": first[:G], "kinds": [L.AGG_SUM, L.AGG_MEAN, L.AGG_COUNT],
                        "outs": [(sums[:G], None), (means[:G], None), (counts[:G], None)], "records": records}, kcol, G)


def _fetch_agg_result(lib, h, key_dtype, val_dtype, kinds):
    """pdx_dist_agg* -> the result dict of the order-free sharded / chunked entry points (device tensors); destroys the handle."""
    import ctypes as C

    from . import column as K

    try:
        G = int(lib.pdx_dist_agg_num_groups(h))
        dev = K._device()
        kcol = K.Column.empty(key_dtype, G, with_validity=True)
        first = torch.empty(max(G, 1), dtype=torch.int64, device=dev)
        outs = [K.Column.empty(K._AGG_OUT_DT[k](val_dtype), G, with_validity=True) for k in kinds]
        km, marr = kcol.mut(), K._mut_array(outs)
        L.check(lib.pdx_dist_agg_fetch(h, C.byref(km), first.data_ptr(), marr, K._stream()))
        kcol._adopt(km)
        for i, o in enumerate(outs):
            o._adopt(marr[i])
    finally:
        lib.pdx_dist_agg_destroy(h)
    res = [(o.values[:G], _unpack_bits_device(o.validity, G) if (o.null_count != 0 and o.validity is not None) else None) for o in outs]
    ok_t = _unpack_bits_device(kcol.validity, G) if kcol.validity is not None else torch.ones(G, dtype=torch.bool, device=dev)
    return {"G": G, "keys": kcol.values[:G], "keys_ok": ok_t, "first_rows": first[:G], "kinds": list(kinds), "outs": res}


def groupby_order_free_chunked(keys, vals, kinds, chunk_rows=0):
    """min / max / count (int64 sum) on one GPU for inputs beyond 2^31 - 1 rows (pdx_groupby_order_free_chunked): chunks of `chunk_rows`
    rows (0 = the largest allowed) reduced without a value sort, their dense per-group partials folded in chunk order."""
    import ctypes as C

    from . import column as K

    lib = L.load()
    kinds = list(kinds)
    h = C.c_void_p()
    ck, cv = keys.c(), vals.c()
    karr = (C.c_int * len(kinds))(*kinds)
    L.check(lib.pdx_groupby_order_free_chunked(C.byref(ck), C.byref(cv), karr, len(kinds), int(chunk_rows), K._stream(), C.byref(h)))
    return _fetch_agg_result(lib, h, keys.dtype, vals.dtype, kinds)


def groupby_sum_mean_count_chunked(keys, vals, chunk_rows=0):
    """The headline query on one GPU for inputs beyond 2^31 - 1 rows (pdx_groupby_sum_mean_count_chunked): chunks of `chunk_rows` rows
    (0 = the largest allowed) merged exactly through the partial-tree records.  Same result dict as the sharded entry points."""
    import ctypes as C

    from . import column as K

    lib = L.load()
    h = C.c_void_p()
    ck, cv = keys.c(), vals.c()
    L.check(lib.pdx_groupby_sum_mean_count_chunked(C.byref(ck), C.byref(cv), int(chunk_rows), K._stream(), C.byref(h)))
    return _fetch_dist_result(lib, h, keys.dtype)


class CDist:
    """Binding over the C ABI's sharded path: the orchestration, the glue kernels and the collectives all live in libpdx_hip.so
    (csrc/dist.hip); python only creates the communicator.

    transport="rccl":  the library's built-in transport -- ncclCommInitRank with an id that rank 0 creates and the process group
                       broadcasts; every collective is an RCCL call made by the library itself (the production path, what a C++
                       host gets from pdx_dist_init).
    transport="torch": pdx_dist_init_custom with callbacks over torch.distributed (host staging; any backend, e.g. gloo).  For
                       rehearsals with several ranks on ONE GPU, where RCCL refuses duplicate devices -- never a measurement."""

    def __init__(self, transport="rccl"):
        import ctypes as C

        self.W, self.r = _world()
        self.lib = L.load()
        self._h = C.c_void_p()
        self.transport = transport
        if transport == "rccl":
            # Every rank-local step that can fail is made collective BEFORE the ranks meet inside ncclCommInitRank: each rank asks the
            # library for an id (this opens librccl and resolves its symbols; only rank 0's id is used) and the statuses are reduced, so a
            # rank that cannot load RCCL raises on ALL ranks instead of leaving the others blocked in the broadcast / the communicator init.
            ident = (C.c_char * 128)()
            rc = int(self.lib.pdx_dist_unique_id(ident))
            msg = self.lib.pdx_last_error().decode("utf-8", "replace") if rc != 0 else ""
            if dist.is_initialized() and self.W > 1:
                dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
                bad = torch.tensor([1 if rc != 0 else 0], dtype=torch.int64, device=dev)
                dist.all_reduce(bad, op=dist.ReduceOp.MAX)
                if int(bad.item()) != 0:
                    raise RuntimeError(f"pdx_dist: RCCL is not usable on every rank (rank {self.r}: {'status %d %s' % (rc, msg) if rc else 'ok'})")
                t = torch.tensor(list(bytes(ident)), dtype=torch.uint8, device=dev)
                dist.broadcast(t, 0)
                ident = (C.c_char * 128)(*bytes(t.cpu().tolist()))
            elif rc != 0:
                raise RuntimeError(f"pdx_dist_unique_id failed: status {rc} {msg}")
            L.check(self.lib.pdx_dist_init(ident, self.W, self.r, C.byref(self._h)))
        elif transport == "torch":
            AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
            A2A = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t),
                              C.POINTER(C.c_size_t), C.c_void_p)

            class Transport(C.Structure):
                _fields_ = [("ctx", C.c_void_p), ("all_gather", AG), ("all_to_all_v", A2A)]

            dev = torch.device("cuda", torch.cuda.current_device())
            W, r = self.W, self.r

            def view(ptr, nbytes):
                return torch.as_tensor(_RawDeviceBytes(ptr, nbytes), device=dev)

            def all_gather(ctx, send, recv, nbytes, stream):
                try:
                    torch.cuda.synchronize()
                    mine = view(send, nbytes).cpu()
                    outs = [torch.empty_like(mine) for _ in range(W)]
                    dist.all_gather(outs, mine)
                    view(recv, nbytes * W).copy_(torch.cat(outs))
                    torch.cuda.synchronize()
                    return 0
                except Exception:  # noqa: BLE001  (an exception must not unwind through the C frames)
                    import traceback

                    traceback.print_exc()
                    return 4

            def all_to_all_v(ctx, send, soff, sbytes, recv, roff, rbytes, stream):
                try:
                    torch.cuda.synchronize()
                    table = torch.tensor([[soff[p], sbytes[p]] for p in range(W)], dtype=torch.int64)
                    tables = [torch.empty_like(table) for _ in range(W)]
                    dist.all_gather(tables, table)
                    for src in range(W):  # (gloo has no all-to-all: every source broadcasts the extent it sends, receivers slice)
                        ext = int((tables[src][:, 0] + tables[src][:, 1]).max().item())
                        buf = view(send, ext).cpu() if src == r else torch.empty(ext, dtype=torch.uint8)
                        if ext:
                            dist.broadcast(buf, src)
                        o, b = int(tables[src][r, 0]), int(tables[src][r, 1])
                        assert b == rbytes[src], "send and receive counts disagree"
                        if b:
                            view(recv + roff[src], b).copy_(buf[o:o + b])
                    torch.cuda.synchronize()
                    return 0
                except Exception:  # noqa: BLE001
                    import traceback

                    traceback.print_exc()
                    return 4

            self._cb = (AG(all_gather), A2A(all_to_all_v))  # keep the thunks alive
            self._tr = Transport(None, self._cb[0], self._cb[1])
            L.check(self.lib.pdx_dist_init_custom(C.byref(self._tr), self.W, self.r, C.byref(self._h)))
        else:
            raise ValueError(transport)

    def close(self):
        if self._h is not None and self._h.value:
            self.lib.pdx_dist_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def groupby_sum_mean_count(self, keys, vals, row_offset=0):
        """The headline query over row-range shards, entirely inside the library.  Same result dict as
        groupby_sum_mean_count_sharded (the FULL result on every rank)."""
        import ctypes as C

        from . import column as K

        h = C.c_void_p()
        ck, cv = keys.c(), vals.c()
        L.check(self.lib.pdx_dist_groupby_sum_mean_count(self._h, C.byref(ck), C.byref(cv), int(row_offset), K._stream(), C.byref(h)))
        return _fetch_dist_result(self.lib, h, keys.dtype)

    def groupby_order_free(self, keys, vals, kinds, row_offset=0):
        """min / max / count (and the sum of an int64 column) over row-range shards, inside the library (pdx_dist_groupby_order_free:
        dense per-group partials, one all-gather, fold in rank order).  Values may carry nulls.  Same result dict as the other sharded
        group-bys: {"G", "keys", "keys_ok", "first_rows", "kinds", "outs": [(values tensor, valid bool tensor | None)]} on every rank."""
        import ctypes as C

        from . import column as K

        kinds = list(kinds)
        h = C.c_void_p()
        ck, cv = keys.c(), vals.c()
        karr = (C.c_int * len(kinds))(*kinds)
        L.check(self.lib.pdx_dist_groupby_order_free(self._h, C.byref(ck), C.byref(cv), karr, len(kinds), int(row_offset), K._stream(), C.byref(h)))
        return _fetch_agg_result(self.lib, h, keys.dtype, vals.dtype, kinds)

    def resample(self, ts, vals, kinds, freq_ns, closed_right=False, label_right=False, origin=L.ORIGIN_START_DAY, origin_custom_ns=0, offset_ns=0):
        """pd::resample(...).{kinds}(col) over a sorted axis sharded by row ranges, inside the library (pdx_dist_resample).
        -> {"labels": int64 tensor, "outs": [(values tensor, valid bool tensor | None) per kind]} on every rank."""
        import ctypes as C

        from . import column as K

        kinds = list(kinds)
        h = C.c_void_p()
        ct, cv = ts.c(), vals.c()
        karr = (C.c_int * len(kinds))(*kinds)
        L.check(self.lib.pdx_dist_resample(self._h, C.byref(ct), C.byref(cv), karr, len(kinds), int(freq_ns), int(bool(closed_right)), int(bool(label_right)),
                                           int(origin), int(origin_custom_ns), int(offset_ns), K._stream(), C.byref(h)))
        try:
            G = int(self.lib.pdx_dist_resampled_num_bins(h))
            labels = K.Column.empty(L.TIMESTAMP_NS, G)
            outs = [K.Column.empty(K._AGG_OUT_DT[k](vals.dtype), G, with_validity=True) for k in kinds]
            lm, marr = labels.mut(), K._mut_array(outs)
            L.check(self.lib.pdx_dist_resampled_fetch(h, C.byref(lm), marr, K._stream()))
            labels._adopt(lm)
            for i, o in enumerate(outs):
                o._adopt(marr[i])
        finally:
            self.lib.pdx_dist_resampled_destroy(h)
        res = []
        for o in outs:
            _, ok = o.to_numpy() if o.null_count != 0 else (None, None)
            res.append((o.values[:G], None if ok is None else torch.from_numpy(ok).to(K._device())))
        return {"labels": labels.values[:G], "outs": res}

    def concat(self, col):
        """all-gather(v) of one column's shards in rank order (pdx_dist_concat)."""
        import ctypes as C

        from . import column as K

        sizes = all_gather_sizes(col.length, K._device()) if self.transport == "torch" or dist.is_initialized() else [col.length]
        out = K.Column.empty(col.dtype, sum(sizes), with_validity=True)
        m, c = out.mut(), col.c()
        L.check(self.lib.pdx_dist_concat(self._h, C.byref(c), C.byref(m), K._stream()))
        out._adopt(m)
        if out.null_count == 0:
            out.validity = None
        return out


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
    if _solo():
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
        cat_keys, cat_first, cat_ok = all_gather_v_multi([uk, fr, uok], sizes)  # one collective for the three dictionary columns
        solo = _solo()
        if solo:
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
        if not solo:
            allc = [torch.empty_like(dense) for _ in range(W)]
            dist.all_gather(allc, dense)
            allc = torch.stack(allc)  # [W, G]
            prefix_g = allc[:r].sum(dim=0) if r > 0 else torch.zeros_like(dense)
            count_g = allc.sum(dim=0)
        else:
            prefix_g, count_g = torch.zeros_like(dense), dense
    with _Stage("5_partial_records"):
        # records are emitted group by group in GLOBAL-id order, so they leave the kernel already partitioned by owner rank
        order = torch.argsort(my_map) if not solo else None
        rec_key, rec_val = engine.partial_records(gv, prefix_g[my_map].contiguous(), my_map, order)
    # owners of contiguous global-id ranges: one all-to-all(v) of contiguous slices
    bounds = [G * d // W for d in range(W + 1)]
    with _Stage("6_all_to_all"):
        if not solo:
            # every rank's cut points travel in ONE small all-gather: a rank learns its own slices and what it will receive from
            # the others from the same host read (one sync instead of three: cuts, send counts, receive counts)
            cuts_t = torch.searchsorted(rec_key, torch.tensor([b * 64 for b in bounds], dtype=torch.int64, device=dev))
            all_cuts = _gather_scalars(cuts_t).tolist()
            cuts = all_cuts[r]
            counts = [[row[d + 1] - row[d] for d in range(W)] for row in all_cuts]
            rk, rv = all_to_all_v_pairs([rec_key[cuts[d]:cuts[d + 1]] for d in range(W)], [rec_val[cuts[d]:cuts[d + 1]] for d in range(W)],
                                        counts)
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


# ---------------------------------------------------------------- sharded whole-column aggregates ("whole-array sum")
def _gather_scalars(t: torch.Tensor):
    """[W, k] stack of a small per-rank tensor, rank order."""
    W, _ = _world()
    if _solo():
        return t[None]
    out = [torch.empty_like(t) for _ in range(W)]
    dist.all_gather(out, t)
    return torch.stack(out)


def aggregate_sharded(engine, col, kind):
    """``Series::sum/mean/min/max/count`` (src/ndframe.cpp:26-31, 119, 162-166, 220) of a column sharded by row ranges in rank
    order.  Returns (value | None, valid count) on every rank, bit-identical to the single-process result:

      count / min / max / int64 sum  -- order-free: per-shard scalars, all-gathered, folded in rank order (min/max keep the FIRST
                                        of ties, as the kernels do; int64 sum wraps)
      float64 sum, float64/int64 mean -- Arrow's pairwise tree is order sensitive.  Without nulls the whole column is ONE group of
                                        the partial-tree exchange (boundary-leaf fragments + aligned subtree nodes per shard,
                                        replayed by one owner); with nulls the 16-value leaves restart at every run of valid
                                        rows, so the rows themselves are routed to one owner (correctness path)."""
    W, r = _world()
    dev = engine.device
    if _solo():
        return engine.aggregate(kind, col)
    is_f = engine.dtype_of(col) == L.FLOAT64
    if kind == L.AGG_COUNT or kind in (L.AGG_MIN, L.AGG_MAX) or (kind == L.AGG_SUM and not is_f):
        v, cnt = engine.aggregate(kind, col)
        dt = torch.float64 if is_f and kind != L.AGG_COUNT else torch.int64
        vals = _gather_scalars(torch.tensor([0 if v is None else v], dtype=dt, device=dev))[:, 0].tolist()
        meta = _gather_scalars(torch.tensor([0 if v is None else 1, cnt], dtype=torch.int64, device=dev)).tolist()
        total = sum(m[1] for m in meta)
        if kind == L.AGG_COUNT:
            return total, total
        have = [x for x, m in zip(vals, meta) if m[0]]
        if not have:
            return None, total
        if kind == L.AGG_SUM:
            acc = 0
            for x in have:
                acc = (acc + int(x) + (1 << 63)) % (1 << 64) - (1 << 63)  # two's-complement wrap
            return acc, total
        best = None
        for x in have:  # rank order; NaN never replaces a number, a number always replaces NaN
            if best is None or (best != best and x == x) or (x < best if kind == L.AGG_MIN else x > best):
                best = x
        return best, total
    # order-sensitive fp64 tree
    x = col if is_f else engine.to_f64(col)
    n_loc = engine.length(col)
    any_nulls = any(all_gather_sizes(1 if engine.valid_bools(col) is not None else 0, dev))
    one_group = engine.col(torch.zeros(n_loc, dtype=torch.int64, device=dev), L.INT64)
    if not any_nulls:
        res = groupby_sum_mean_count_sharded(engine, one_group, x, row_offset=0)
        if res["G"] == 0:
            return None, 0
        cnt = int(res["outs"][2][0][0].item())
        return float(res["outs"][0 if kind == L.AGG_SUM else 1][0][0].item()), cnt
    res = groupby_agg_sharded(engine, one_group, x, [kind, L.AGG_COUNT])
    if res["G"] == 0:
        return None, 0
    (v, ok), (c, _) = res["outs"]
    cnt = int(c[0].item())
    if ok is not None and not bool(ok[0].item()):
        return None, cnt
    return float(v[0].item()), cnt


# ---------------------------------------------------------------- sharded concat: all-gather(v) in rank order
def concat_sharded(engine, col):
    """``pd::concat`` (rows, src/concat.cpp:116-190) of the per-rank columns in rank order; every rank gets the whole column."""
    dev = engine.device
    sizes = all_gather_sizes(engine.length(col), dev)
    v = all_gather_v(engine.values(col), sizes)
    okb = engine.valid_bools(col)
    ok = None
    if any(all_gather_sizes(0 if okb is None else 1, dev)):
        okb = torch.ones(engine.length(col), dtype=torch.bool, device=dev) if okb is None else okb
        ok = all_gather_v(okb.to(torch.uint8), sizes).to(torch.bool)
    return engine.col(v, engine.dtype_of(col), ok)


# ---------------------------------------------------------------- sharded resample
_DAY_NS = 86400000000000


def _resample_grid(first, last, freq, closed_right, origin_type, origin_custom, offset):
    """adjustDatesAnchored + date_range on the WHOLE axis (src/resample.cpp:85-178, src/core.cpp:308-331), exact Python ints:
    (origin without offset, first edge, number of bins)."""
    if origin_type == L.ORIGIN_EPOCH:
        origin = 0
    elif origin_type == L.ORIGIN_START_DAY:
        origin = first // _DAY_NS * _DAY_NS
    elif origin_type == L.ORIGIN_START:
        origin = first
    elif origin_type == L.ORIGIN_END:
        origin = last
    elif origin_type == L.ORIGIN_END_DAY:
        origin = last // _DAY_NS * _DAY_NS
    else:
        origin = origin_custom
    o = origin + offset

    def cmod(a, b):  # C++ '%' (truncating), as the reference computes the offsets
        m = abs(a) % b
        return m if a >= 0 else -m

    fo, lo = cmod(first - o, freq), cmod(last - o, freq)
    f, l = first, last
    if closed_right:
        f = f - fo if fo > 0 else f - freq
        if lo > 0:
            l += freq - lo
    else:
        if fo > 0:
            f -= fo
        l = l + freq - lo if lo > 0 else l + freq
    if f >= l:
        raise L.PdxError(L.INVALID, "start date has to be less than end date")
    nedges = (l - f) // freq + 1
    return origin, f, nedges - 1


def resample_agg_sharded(engine, ts, vals, kinds, freq_ns, closed_right=False, label_right=False, origin=L.ORIGIN_START_DAY,
                         origin_custom_ns=0, offset_ns=0):
    """``pd::resample(df, rule).{sum,mean,min,max,count}(col)`` (src/resample.h:91-122, src/group_by.h:255-299) over an axis
    sharded by row ranges in rank order (sorted timestamps, so shards are time ranges).  Bins are made whole before any
    arithmetic: the leading rows of a shard that fall into a bin already open on an earlier rank move to that bin's first rank
    (one all-to-all(v); usually a few hundred rows per boundary).  Every rank then resamples its rows on the WHOLE axis' grid
    (PDX_ORIGIN_CUSTOM | PDX_ORIGIN_SHARD) and the (label, value) rows are all-gathered in rank order == label order.
    Returns {"labels": int64 tensor, "outs": [(values, valid|None) per kind]} on every rank."""
    W, r = _world()
    dev = engine.device
    n_loc = engine.length(ts)
    tv = engine.values(ts)
    mine = torch.tensor([n_loc, int(tv[0].item()) if n_loc else 0, int(tv[n_loc - 1].item()) if n_loc else 0], dtype=torch.int64, device=dev)
    info = _gather_scalars(mine).tolist()
    live = [q for q in range(W) if info[q][0] > 0]
    N = sum(i[0] for i in info)
    if N == 0:
        empty = torch.zeros(0, dtype=torch.int64, device=dev)
        return {"labels": empty, "outs": [(torch.zeros(0, dtype=torch.float64, device=dev), None) for _ in kinds]}
    for a, b in zip(live, live[1:]):
        if info[a][2] > info[b][1]:
            raise L.PdxError(L.INVALID, "pdx_resample_create: timestamps must be sorted ascending")
    g_first, g_last = info[live[0]][1], info[live[-1]][2]
    _, first_edge, nbins = _resample_grid(g_first, g_last, int(freq_ns), bool(closed_right), origin, int(origin_custom_ns), int(offset_ns))
    if N < nbins:
        raise L.PdxError(L.INVALID, "upSampling is not implemented.")  # GroupInfo::upsampling on the whole axis (src/resample.h:14-17)

    def bin_of(t):  # closed left: [e_k, e_k + f); closed right: (e_k, e_k + f]
        return (t - first_edge - (1 if closed_right else 0)) // int(freq_ns)

    # the first rank holding rows of my leading bin
    send_to, m = r, 0
    if n_loc:
        b0 = bin_of(info[r][1])
        for q in reversed([q for q in live if q < r]):
            if bin_of(info[q][2]) != b0:
                break
            send_to = q
            if bin_of(info[q][1]) != b0:
                break
        if send_to != r:
            upper = first_edge + (b0 + 1) * int(freq_ns)
            m = engine.count_below(ts, upper, inclusive=bool(closed_right))
    vok = engine.valid_bools(vals)
    has_nulls = any(all_gather_sizes(0 if vok is None else 1, dev))
    if not _solo():
        vv = engine.values(vals)
        head = [tv[:m], vv[:m]]
        if has_nulls:
            vok = torch.ones(n_loc, dtype=torch.bool, device=dev) if vok is None else vok
            head.append(vok[:m].to(torch.int64))
        recv = []
        for h in head:
            chunks = [h if d == send_to and d != r else h[:0] for d in range(W)]
            recv.append(torch.cat(all_to_all_v(chunks)))
        ts2 = engine.col(torch.cat([tv[m:], recv[0]]), engine.dtype_of(ts))
        ok2 = torch.cat([vok[m:], recv[2].to(torch.bool)]) if has_nulls else None
        vals2 = engine.col(torch.cat([vv[m:], recv[1]]), engine.dtype_of(vals), ok2)
    else:
        ts2, vals2 = ts, vals
    # every shard is anchored at the whole axis' FIRST EDGE (not at the origin: with a negative first offset the reference's grid
    # starts at the first timestamp itself, src/resample.cpp:85-178), so all shards bin on one grid
    gb = engine.resample(ts2, int(freq_ns), bool(closed_right), bool(label_right), L.ORIGIN_CUSTOM | L.ORIGIN_SHARD, first_edge, 0)
    labels, _ = engine.unique_keys(gb)
    outs_l = engine.agg(gb, vals2, list(kinds))
    sizes = all_gather_sizes(int(labels.numel()), dev)
    out = {"labels": all_gather_v(labels, sizes), "outs": []}
    for c in outs_l:
        v = all_gather_v(engine.values(c), sizes)
        okb = engine.valid_bools(c)
        ok = None
        if any(all_gather_sizes(0 if okb is None else 1, dev)):
            okb = torch.ones(int(labels.numel()), dtype=torch.bool, device=dev) if okb is None else okb
            ok = all_gather_v(okb.to(torch.uint8), sizes).to(torch.bool)
        out["outs"].append((v, ok))
    return out
