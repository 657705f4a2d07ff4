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
