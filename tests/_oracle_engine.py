"""Oracle-backed engine for the multi-rank orchestration tests (CPU, gloo).  TEST INFRASTRUCTURE: it plugs the CPU oracle
into pandasarrow_amd.dist in place of HipEngine so the N>1 path (dictionary merge, routing, all-to-all, placement) can be
exercised without a GPU.  The product never constructs this."""
import numpy as np
import torch

import oracle as orc

INT64, FLOAT64 = 0, 1


class OCol:
    def __init__(self, values, valid=None, dtype=None):
        self.values = np.ascontiguousarray(values)
        self.valid = None if valid is None or np.all(valid) else np.asarray(valid, bool)
        self.dtype = dtype if dtype is not None else (FLOAT64 if self.values.dtype == np.float64 else INT64)


class OGroup:
    def __init__(self, keys: OCol):
        self.ids, self.uniq, self.isnull, self.first = orc.group_ids(keys.values.astype(np.int64), keys.valid)
        self.G = len(self.uniq)


class OracleEngine:
    device = torch.device("cpu")

    def col(self, t, dtype, ok=None):
        return OCol(t.numpy().copy(), None if ok is None else ok.numpy().astype(bool), dtype)

    def values(self, col):
        return torch.from_numpy(np.ascontiguousarray(col.values))

    def group(self, key_col):
        return OGroup(key_col)

    def unique_keys(self, gb):
        return torch.from_numpy(gb.uniq.copy()), torch.from_numpy(~gb.isnull)

    def first_rows(self, gb):
        return torch.from_numpy(gb.first.copy())

    def group_ids(self, gb):
        return torch.from_numpy(gb.ids.astype(np.int64))

    def map_ids(self, gb, mapping):
        return OCol(mapping.numpy()[gb.ids.astype(np.int64)])

    def agg(self, gb, values_col, kinds):
        outs = []
        for k in kinds:
            v, ok = orc.groupby_agg(k, gb.ids, gb.G, values_col.values, values_col.valid)
            outs.append(OCol(v, ok))
        return outs

    def select_eq(self, cols, by_col, value):
        m = by_col.values == value
        return [OCol(c.values[m], None if c.valid is None else c.valid[m], c.dtype) for c in cols]

    def place(self, cols, positions, n):
        outs = []
        p = positions.numpy()
        for c in cols:
            v = np.zeros(n, c.values.dtype)
            v[p] = c.values
            ok = None
            if c.valid is not None:
                ok = np.ones(n, bool)
                ok[p] = c.valid
            outs.append(OCol(v, ok, c.dtype))
        return outs

    def dtype_of(self, col):
        return col.dtype

    def valid_bools(self, col):
        return None if col.valid is None else torch.from_numpy(col.valid)


    # ---- partial-tree exchange primitives: independent Python restatement of include/pdx/abi.h "exact multi-GPU fp64 sum"
    def group_values(self, gb, values_col):
        offsets, rows = orc.groupings(gb.ids, gb.G)
        return {"gb": gb, "vals": [values_col.values[rows[offsets[g]:offsets[g + 1]]] for g in range(gb.G)]}

    def grouped_counts(self, gv):
        return torch.tensor([len(v) for v in gv["vals"]], dtype=torch.int64)

    @staticmethod
    def _blocks(kf, kl):
        s = kf
        while s < kl:
            j = 62 if s == 0 else (s & -s).bit_length() - 1
            j = min(j, (kl - s).bit_length() - 1)
            yield s, j
            s += 1 << j

    def partial_records(self, gv, prefix, gid_map, order=None):
        keys, vals = [], []
        P, gm = prefix.numpy(), gid_map.numpy()
        emission = range(len(gv["vals"])) if order is None else [int(x) for x in order.numpy()]
        for lg in emission:
            v = gv["vals"][lg]
            a, c = int(P[lg]), len(v)
            if c == 0:
                continue
            b, gkey = a + c, int(gm[lg]) * 64
            kf, kl = (a + 15) // 16, b // 16
            if kf > kl:
                keys += [gkey] * c
                vals += list(v)
                continue
            h = 16 * kf - a
            keys += [gkey] * h
            vals += list(v[:h])
            for s0, j in self._blocks(kf, kl):
                cn = _Counter()
                for q in range(1 << j):
                    base = 16 * (s0 + q) - a
                    acc = 0.0
                    for e in range(16):
                        acc = acc + float(v[base + e])
                    cn.push(acc, 0)
                keys.append(gkey + j + 1)
                vals.append(cn.sum[j])
            t0 = 16 * kl - a
            if c > t0:  # the rows that begin the last leaf: ONE record, their sequential sum (code 32 + rows)
                acc = 0.0
                for x in v[t0:]:
                    acc = acc + float(x)
                keys.append(gkey + 32 + (c - t0))
                vals.append(acc)
        return torch.tensor(keys, dtype=torch.int64), torch.tensor(vals, dtype=torch.float64)

    def replay(self, rec_key, rec_val, gid_lo, n_own):
        k, v = rec_key.numpy(), rec_val.numpy()
        per = [[] for _ in range(n_own)]
        for key, val in zip(k, v):  # arrival order == (source rank, emission) order
            per[(int(key) >> 6) - gid_lo].append((int(key) & 63, float(val)))
        out = np.zeros(n_own)
        for g, recs in enumerate(per):
            assert recs, "an owned group received no record"
            cn, acc, fill = _Counter(), 0.0, 0
            for lvl, val in recs:
                if lvl == 0:
                    acc = acc + val
                    fill += 1
                    if fill == 16:
                        cn.push(acc, 0)
                        acc, fill = 0.0, 0
                elif lvl > 32:  # the first lvl - 32 rows of a leaf, already summed in order
                    assert fill == 0
                    acc, fill = val, lvl - 32
                else:
                    assert fill == 0
                    cn.push(val, lvl - 1)
            if fill:
                cn.push(acc, 0)
            out[g] = cn.finish()
        return torch.from_numpy(out)

    def select_tensor_eq(self, tensors, by, value):
        m = by == value
        return [t[m] for t in tensors]

    # ---- whole-column aggregates / resample over shards
    def length(self, col):
        return len(col.values)

    def aggregate(self, kind, col):
        v, cnt = orc.agg(kind, col.values, col.valid)
        return (v, v) if kind == orc.AGG_COUNT else (v, cnt)

    def to_f64(self, col):
        return OCol(col.values.astype(np.float64), col.valid, FLOAT64)

    def count_below(self, ts_col, edge, inclusive):
        return int(np.count_nonzero(ts_col.values <= edge if inclusive else ts_col.values < edge))

    def resample(self, ts_col, freq_ns, closed_right, label_right, origin, origin_custom_ns, offset_ns):
        shard = bool(origin & 0x100)
        g = OGroup.__new__(OGroup)
        if len(ts_col.values) == 0:
            g.ids, g.uniq, g.isnull, g.first = np.zeros(0, np.uint32), np.zeros(0, np.int64), np.zeros(0, bool), np.zeros(0, np.int64)
        else:
            labels = orc.resample_row_labels(ts_col.values.astype(np.int64), freq_ns, shard=shard, closed_right=closed_right, label_right=label_right,
                                             origin=origin & 0xFF, origin_custom_ns=origin_custom_ns, offset_ns=offset_ns)
            g.ids, g.uniq, g.isnull, g.first = orc.group_ids(labels)
        g.G = len(g.uniq)
        return g


class _Counter:
    """Arrow's binary counter with pushes at arbitrary levels."""

    def __init__(self):
        self.sum = [0.0] * 64
        self.mask = 0
        self.root = 0

    def push(self, x, level):
        cur, m = level, 1 << level
        self.sum[cur] += x
        self.mask ^= m
        while (self.mask & m) == 0:
            x = self.sum[cur]
            self.sum[cur] = 0.0
            cur += 1
            m <<= 1
            self.sum[cur] += x
            self.mask ^= m
        self.root = max(self.root, cur)

    def finish(self):
        for i in range(1, self.root + 1):
            self.sum[i] += self.sum[i - 1]
        return self.sum[self.root]
