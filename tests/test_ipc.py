"""Arrow IPC streams (DataFrame::readBinary / toBinary, reference src/dataframe.cpp:726-791).

CPU part (no GPU): the parser of libpdx_hip.so against streams written by pyarrow / Arrow C++ 25.0.0 (tests/golden/ipc_fixtures.npz,
frozen by oracle/gen_golden_ipc.py), and the writer with host-resident columns, whose bytes pyarrow must read back identically.
GPU part: the same fixtures through readBinary (one host->device copy, columns aliasing the uploaded body, widening kernels) and
toBinary from device columns."""
import ctypes as C
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Z = np.load(os.path.join(ROOT, "tests", "golden", "ipc_fixtures.npz"))
MAN = json.loads(str(Z["manifest"]))
KIND_DTYPE = {"i64": 0, "f64": 1, "bool": 2, "u64": 3, "ts": 4}


@pytest.fixture(scope="module")
def lib():
    from pandasarrow_amd import _lib as L

    return L


def _open(L, blob):
    h = C.c_void_p()
    buf = bytes(blob)
    rc = L.load().pdx_ipc_open(buf, len(buf), C.byref(h))
    return rc, h, buf


@pytest.mark.parametrize("case", list(MAN["cases"]))
def test_parse_fixture(lib, case):
    info = MAN["cases"][case]
    rc, h, _keep = _open(lib, Z[f"{case}/blob"])
    assert rc == 0, lib.load().pdx_last_error()
    so = lib.load()
    try:
        assert so.pdx_ipc_num_rows(h) == info["rows"] and so.pdx_ipc_num_columns(h) == len(info["columns"])
        for i, col in enumerate(info["columns"]):
            assert so.pdx_ipc_column_name(h, i).decode() == col["name"]
            c = lib.PdxColumn()
            assert so.pdx_ipc_column(h, i, C.byref(c)) == 0
            assert c.dtype == KIND_DTYPE[col["kind"]] and c.length == info["rows"] and c.null_count == col["nulls"] and c.offset == 0
            assert not c.values  # nothing is on the device before pdx_ipc_load
        meta = {so.pdx_ipc_metadata_key(h, i).decode(): so.pdx_ipc_metadata_value(h, i).decode() for i in range(so.pdx_ipc_num_metadata(h))}
        assert meta == info["metadata"]
    finally:
        so.pdx_ipc_destroy(h)


@pytest.mark.parametrize("case", list(MAN["rejects"]))
def test_rejects(lib, case):
    if f"{case}/blob" not in Z.files:
        pytest.skip("codec not available when the fixtures were generated")
    rc, h, _keep = _open(lib, Z[f"{case}/blob"])
    assert rc in (lib.INVALID, lib.NOT_IMPLEMENTED)
    assert MAN["rejects"][case] in lib.load().pdx_last_error().decode()


def test_rejects_garbage_and_truncation(lib):
    good = bytes(Z["mixed_9/blob"])
    for blob in (b"", b"\x00" * 7, b"not an arrow stream at all....", good[:40], good[: len(good) // 2], b"\xff\xff\xff\xff\x10\x00\x00\x00" + b"\xff" * 16):
        rc, h, _keep = _open(lib, np.frombuffer(blob, np.uint8))
        assert rc != 0 and lib.load().pdx_last_error()


def _record_batch_structs(blob):
    """-> (position of FieldNode[0], position of Buffer[0], body length, position of RecordBatch.length) of the first RecordBatch message.
    A 20-line flatbuffer walk (continuation marker, metadata length, Message table -> header union -> RecordBatch.nodes / .buffers)."""
    import struct

    def field(table, slot):
        vt = table - struct.unpack_from("<i", blob, table)[0]
        vsize = struct.unpack_from("<H", blob, vt)[0]
        if 4 + 2 * slot + 2 > vsize:
            return None
        o = struct.unpack_from("<H", blob, vt + 4 + 2 * slot)[0]
        return table + o if o else None

    def indirect(pos):
        return pos + struct.unpack_from("<I", blob, pos)[0]

    pos = 0
    while pos + 8 <= len(blob):
        assert struct.unpack_from("<I", blob, pos)[0] == 0xFFFFFFFF
        mlen = struct.unpack_from("<i", blob, pos + 4)[0]
        meta = pos + 8
        msg = indirect(meta)
        header_type = blob[field(msg, 1)]
        body_len = struct.unpack_from("<q", blob, field(msg, 3))[0] if field(msg, 3) else 0
        if header_type == 3:  # MessageHeader.RecordBatch
            rb = indirect(field(msg, 2))
            return indirect(field(rb, 1)) + 4, indirect(field(rb, 2)) + 4, body_len, field(rb, 0)
        pos = meta + mlen + body_len
    raise AssertionError("no record batch in the stream")


def test_rejects_crafted_offsets_and_lengths(lib):
    """FieldNode / Buffer entries of a VALID stream patched to values whose sums or products wrap int64.  The open must
    refuse each of them (the reference's reader, Arrow's, does): accepted, they would send a kernel reading far outside the upload."""
    import struct

    good = bytearray(bytes(Z["mixed_9/blob"]))
    nodes, bufs, body, rb_len = _record_batch_structs(good)
    ncols = len(MAN["cases"]["mixed_9"]["columns"])
    rc, h, _keep = _open(lib, np.frombuffer(bytes(good), np.uint8))
    assert rc == 0
    lib.load().pdx_ipc_destroy(h)
    rows = struct.unpack_from("<q", good, nodes)[0]
    assert rows == MAN["cases"]["mixed_9"]["rows"]
    big = 0x7FFFFFFFFFFFFFF8
    patches = [
        ("values offset + length wraps", bufs + 16 * 1, (big, 16)),                 # Buffer[1] = column 0's values
        ("negative values length", bufs + 16 * 1 + 8, (-8,)),
        ("validity offset + length wraps", bufs + 16 * 0, (big, 64)),
        ("negative validity length", bufs + 16 * 0 + 8, (-1,)),
        ("values offset past the body", bufs + 16 * 1, (body + 8, 0)),
        ("row count whose byte size wraps", "rows", (1 << 61,)),                  # (batch length and every FieldNode agree on it)
        ("row count beyond the body", "rows", (body * 8 + 64,)),
        ("null count above the row count", nodes + 8, (rows + 1,)),
        ("negative null count", nodes + 8, (-1,)),
    ]
    for what, at, vals in patches:
        bad = bytearray(good)
        if at == "rows":
            for pos in [rb_len] + [nodes + 16 * i for i in range(ncols)]:
                struct.pack_into("<q", bad, pos, vals[0])
        else:
            struct.pack_into("<" + "q" * len(vals), bad, at, *vals)
        rc, h, _keep = _open(lib, np.frombuffer(bytes(bad), np.uint8))
        assert rc == lib.INVALID, what
        assert "inconsistent" in lib.load().pdx_last_error().decode() or "does not match" in lib.load().pdx_last_error().decode(), what


def _bits(x):
    return np.concatenate([np.packbits(np.asarray(x, bool), bitorder="little"), np.zeros(16, np.uint8)])


@pytest.mark.parametrize("n", [0, 1, 7, 64, 1000])
def test_write_host_columns_read_by_pyarrow(lib, n):
    pa = pytest.importorskip("pyarrow")
    L = lib
    rng = np.random.default_rng(n)
    a, f = rng.integers(-2**62, 2**62, n + 5), rng.standard_normal(n + 5)
    bv, vf, vb = rng.random(n + 5) < 0.5, rng.random(n + 5) > 0.2, rng.random(n + 5) > 0.3
    ts, u = rng.integers(0, 10**18, n + 5), rng.integers(0, 2**63, n + 5).astype(np.uint64) * 2 + 1
    bb, vfb, vbb = _bits(bv), _bits(vf), _bits(vb)
    off = 5 if n else 0  # sliced columns: element / bit offset 5
    cols = (L.PdxColumn * 5)(L.PdxColumn(L.INT64, 0, n, off, 0, None, a.ctypes.data), L.PdxColumn(L.FLOAT64, 0, n, off, -1, vfb.ctypes.data, f.ctypes.data),
                             L.PdxColumn(L.BOOL, 0, n, off, -1, vbb.ctypes.data, bb.ctypes.data), L.PdxColumn(L.TIMESTAMP_NS, 0, n, off, 0, None, ts.ctypes.data),
                             L.PdxColumn(L.UINT64, 0, n, off, 0, None, u.ctypes.data))
    names = (C.c_char_p * 5)(b"a", b"f", b"flag", b"when", b"u")
    kv = (C.c_char_p * 4)(b"who", b"pdx", b"rows", str(n).encode())
    out, sz = C.c_void_p(), C.c_size_t()
    assert L.load().pdx_ipc_write(cols, names, 5, kv, 2, 1, None, C.byref(out), C.byref(sz)) == 0, L.load().pdx_last_error()
    data = C.string_at(out, sz.value)
    L.load().pdx_ipc_free_blob(out)
    rd = pa.ipc.open_stream(data)
    assert [str(t) for t in rd.schema.types] == ["int64", "double", "bool", "timestamp[ns]", "uint64"] and rd.schema.names == ["a", "f", "flag", "when", "u"]
    b, md = rd.read_next_batch_with_custom_metadata()
    b.validate(full=True)
    assert {k.decode(): v.decode() for k, v in md.items()} == {"who": "pdx", "rows": str(n)} and b.num_rows == n
    sl = slice(off, off + n)
    assert np.array_equal(b.column(0).to_numpy(zero_copy_only=False), a[sl]) and b.column(0).null_count == 0
    assert b.column(1).null_count == int((~vf[sl]).sum()) and np.array_equal(np.asarray(b.column(1).is_valid()), vf[sl])
    assert np.array_equal(b.column(1).to_numpy(zero_copy_only=False)[vf[sl]], f[sl][vf[sl]])
    assert np.array_equal(np.asarray(b.column(2).is_valid()), vb[sl]) and np.array_equal(np.asarray(b.column(2).fill_null(False))[vb[sl]], bv[sl][vb[sl]])
    assert np.array_equal(b.column(3).cast(pa.int64()).to_numpy(zero_copy_only=False), ts[sl]) and np.array_equal(b.column(4).to_numpy(zero_copy_only=False), u[sl])
    with pytest.raises(StopIteration):
        rd.read_next_batch()
    # and our own reader takes our own stream
    rc, h, _keep = _open(L, np.frombuffer(data, np.uint8))
    assert rc == 0 and L.load().pdx_ipc_num_rows(h) == n and L.load().pdx_ipc_num_columns(h) == 5
    L.load().pdx_ipc_destroy(h)


# ------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def px():
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from pandasarrow_amd import _lib as L
    from pandasarrow_amd import api, column

    L.check(L.load().pdx_init(0))

    class NS:
        pass

    ns = NS()
    ns.L, ns.K, ns.api, ns.Column, ns.torch = L, column, api, column.Column, torch
    return ns


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(MAN["cases"]))
def test_read_binary_fixture(px, case):
    info = MAN["cases"][case]
    df = px.api.DataFrame.readBinary(bytes(Z[f"{case}/blob"]), index=info.get("index"))
    cols = list(zip(df.names, df.cols))
    if info.get("index"):
        assert df.index is not None and df.index.dtype == px.L.TIMESTAMP_NS  # int64 index column -> timestamp[ns]
        cols.append((info["index"], df.index))
    assert [nm for nm, _ in cols] == [c["name"] for c in info["columns"]] and df.metadata == info["metadata"]
    for i, ((nm, col), meta) in enumerate(zip(cols, info["columns"])):
        vals, valid = col.to_numpy()
        ev, eok = Z[f"{case}/col{i}"], Z[f"{case}/valid{i}"]
        assert col.length == info["rows"] and col.null_count == meta["nulls"]
        if nm != info.get("index"):
            assert col.dtype == KIND_DTYPE[meta["kind"]], nm
        assert (valid is None and eok.all()) or np.array_equal(valid, eok), nm
        if ev.dtype == np.float64:
            assert np.array_equal(vals.view(np.uint64)[eok], ev.view(np.uint64)[eok]), nm
        else:
            assert np.array_equal(vals[eok], ev[eok]), nm
    # the loaded columns are ordinary columns of the path: run a kernel over them
    if info["rows"] and "f64" in df.names:
        s = df["f64"].sum()
        import oracle as orc

        j = df.names.index("f64")
        exp, _ = orc.agg(orc.AGG_SUM, Z[f"{case}/col{j}"], Z[f"{case}/valid{j}"])
        assert (s.value is None and exp is None) or np.float64(s.value).view(np.uint64) == np.float64(exp).view(np.uint64)


@pytest.mark.gpu
def test_to_binary_from_device_and_round_trip(px):
    pa = pytest.importorskip("pyarrow")
    rng = np.random.default_rng(5)
    n = 100_003
    f, ok = rng.standard_normal(n), rng.random(n) > 0.1
    i = rng.integers(-2**62, 2**62, n)
    flag = rng.random(n) < 0.3
    ts = 946684800 * 10**9 + np.arange(n, dtype=np.int64) * 10**9
    df = px.api.DataFrame({"f": px.api.Series(f, valid=ok), "i": i, "flag": flag}, index=px.Column.from_numpy(ts, dtype=px.L.TIMESTAMP_NS))
    blob = df.toBinary(index="__index__", metadata={"k": "v"})
    b, md = pa.ipc.open_stream(blob).read_next_batch_with_custom_metadata()
    b.validate(full=True)
    assert b.schema.names == ["f", "i", "flag", "__index__"] and {k.decode(): v.decode() for k, v in md.items()} == {"k": "v"}
    assert b.column(0).null_count == int((~ok).sum()) and np.array_equal(b.column(0).to_numpy(zero_copy_only=False)[ok], f[ok])
    assert np.array_equal(b.column(1).to_numpy(), i) and np.array_equal(b.column(2).to_numpy(zero_copy_only=False), flag)
    assert str(b.schema.types[3]) == "int64" and np.array_equal(b.column(3).to_numpy(), ts)
    back = px.api.DataFrame.readBinary(blob, index="__index__")
    assert back.names == ["f", "i", "flag"] and back.index.dtype == px.L.TIMESTAMP_NS and np.array_equal(back.index.to_numpy()[0], ts)
    fv, fok = back["f"].to_numpy()
    assert np.array_equal(fok, ok) and np.array_equal(fv.view(np.uint64)[ok], f.view(np.uint64)[ok])
    assert np.array_equal(back["i"].values(), i) and np.array_equal(back["flag"].values(), flag) and back.metadata == {"k": "v"}
    # a missing index name: the frame comes back without an index (the reference logs and carries on)
    assert px.api.DataFrame.readBinary(blob, index="nope").index is None
    # sliced device columns (bit offset 3) serialise from their offset
    sl = px.api.DataFrame({"f": df["f"].col.slice(3, 1000), "flag": df["flag"].col.slice(3, 1000)})
    b2 = pa.ipc.open_stream(sl.toBinary()).read_next_batch()
    b2.validate(full=True)
    assert b2.num_rows == 1000 and np.array_equal(np.asarray(b2.column(0).is_valid()), ok[3:1003]) and np.array_equal(b2.column(1).to_numpy(zero_copy_only=False), flag[3:1003])
    # the ingested columns are ordinary device columns: group by one of them
    import oracle as orc

    gb = px.api.DataFrame({"k": px.Column.from_numpy(i % 7), "f": back["f"].col}).group_by("k")
    ids, uniq, _, _ = orc.group_ids(i % 7)
    exp, eok = orc.groupby_agg(orc.AGG_SUM, ids, len(uniq), f, ok)
    got = gb.sum("f")
    assert np.array_equal(got.values().view(np.uint64)[eok], exp.view(np.uint64)[eok])
