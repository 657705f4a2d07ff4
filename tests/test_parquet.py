"""Parquet -> device columns (DataFrame::readParquet, reference src/dataframe.cpp:646-683).

CPU part (no GPU): the footer parser of libpdx_hip.so against files written by pyarrow / Arrow C++ 25.0.0
(tests/golden/parquet_fixtures.npz, frozen by oracle/gen_golden_parquet.py), the named refusals, garbage and truncation.
GPU part: the same files decoded on the device (Snappy, definition levels, PLAIN / dictionary / RLE-boolean values, page v1 / v2),
compared bit for bit with what pyarrow reads from the same bytes."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Z = np.load(os.path.join(ROOT, "tests", "golden", "parquet_fixtures.npz"))
MAN = json.loads(str(Z["manifest"]))
KIND_DTYPE = {"i64": 0, "f64": 1, "bool": 2, "u64": 3, "ts": 4}


@pytest.fixture(scope="module")
def lib():
    from pandasarrow_amd import _lib as L

    return L


def _open(L, blob):
    h = C.c_void_p()
    buf = bytes(blob)
    rc = L.load().pdx_parquet_open(buf, len(buf), C.byref(h))
    return rc, h, buf


@pytest.mark.parametrize("case", list(MAN["cases"]))
def test_parse_footer(lib, case):
    info = MAN["cases"][case]
    rc, h, _keep = _open(lib, Z[f"{case}/blob"])
    assert rc == 0, lib.load().pdx_last_error()
    so = lib.load()
    try:
        assert so.pdx_parquet_num_rows(h) == info["rows"] and so.pdx_parquet_num_columns(h) == len(info["columns"])
        for i, col in enumerate(info["columns"]):
            assert so.pdx_parquet_column_name(h, i).decode() == col["name"]
            c = lib.PdxColumn()
            assert so.pdx_parquet_column(h, i, C.byref(c)) == 0
            assert c.dtype == KIND_DTYPE[col["kind"]] and c.length == info["rows"] and c.offset == 0
            assert c.null_count in (col["nulls"], -1)      # the chunk statistics, when the writer kept them
            assert not c.values                             # nothing is on the device before pdx_parquet_load
        keys = [so.pdx_parquet_metadata_key(h, i).decode() for i in range(so.pdx_parquet_num_metadata(h))]
        assert "ARROW:schema" in keys                       # pyarrow stores its schema in the footer's key_value_metadata
    finally:
        so.pdx_parquet_destroy(h)


@pytest.mark.parametrize("case", list(MAN["rejects"]))
def test_rejects_by_name(lib, case):
    rc, h, _keep = _open(lib, Z[f"{case}/blob"])
    msg = lib.load().pdx_last_error().decode()
    assert rc in (lib.INVALID, lib.NOT_IMPLEMENTED), (case, rc)
    assert MAN["rejects"][case] in msg, (case, msg)


def test_rejects_garbage_truncation_and_crafted_footers(lib):
    good = bytes(Z["mix_1000_none_plain_v1_1/blob"])
    for blob in (b"", b"PAR1", b"PAR1PAR1", b"not a parquet file at all ........", good[:100], good[: len(good) // 2], good[:-1], good[4:],
                 b"PAR1" + b"\xff" * 64 + struct.pack("<I", 64) + b"PAR1", good[:-8] + struct.pack("<I", 0x7FFFFFF0) + b"PAR1"):
        rc, h, _keep = _open(lib, blob)
        assert rc != 0 and lib.load().pdx_last_error(), blob[:16]
    # every single-byte corruption of the footer either parses to SOMETHING consistent or is refused -- never a crash or a wild read
    flen = struct.unpack("<I", good[-8:-4])[0]
    foot0 = len(good) - 8 - flen
    rng = np.random.default_rng(5)
    for _ in range(300):
        bad = bytearray(good)
        at = foot0 + int(rng.integers(0, flen))
        bad[at] = int(rng.integers(0, 256))
        rc, h, _keep = _open(lib, bytes(bad))
        if rc == 0:
            lib.load().pdx_parquet_destroy(h)


def _bits(x):
    return np.concatenate([np.packbits(np.asarray(x, bool), bitorder="little"), np.zeros(16, np.uint8)])


@pytest.mark.parametrize("n", [0, 1, 7, 64, 1000, 1_048_576 + 13])
def test_write_host_columns_read_by_pyarrow(lib, n):
    """pdx_parquet_write (DataFrame::toParquet, src/dataframe.cpp:685-724) with host-resident columns: Arrow's own reader must give
    back every value and null; our reader must open the file too.  The last size spans two data pages."""
    pa = pytest.importorskip("pyarrow")
    pq = pytest.importorskip("pyarrow.parquet")
    import io

    L = lib
    rng = np.random.default_rng(n)
    a, f = rng.integers(-2**62, 2**62, n + 5), rng.standard_normal(n + 5)
    bv, vf, vb = rng.random(n + 5) < 0.5, rng.random(n + 5) > 0.2, rng.random(n + 5) > 0.3
    ts, u = rng.integers(0, 10**18, n + 5), rng.integers(0, 2**63, n + 5).astype(np.uint64) * 2 + 1
    if n > 3:
        f[7], f[8] = np.nan, -0.0
    bb, vfb, vbb = _bits(bv), _bits(vf), _bits(vb)
    off = 5 if n else 0  # sliced columns: element / bit offset 5
    cols = (L.PdxColumn * 5)(L.PdxColumn(L.INT64, 0, n, off, 0, None, a.ctypes.data), L.PdxColumn(L.FLOAT64, 0, n, off, -1, vfb.ctypes.data, f.ctypes.data),
                             L.PdxColumn(L.BOOL, 0, n, off, -1, vbb.ctypes.data, bb.ctypes.data), L.PdxColumn(L.TIMESTAMP_NS, 0, n, off, 0, None, ts.ctypes.data),
                             L.PdxColumn(L.UINT64, 0, n, off, 0, None, u.ctypes.data))
    names = (C.c_char_p * 5)(b"a", b"f", b"flag", b"when", b"u")
    out, sz = C.c_void_p(), C.c_size_t()
    assert L.load().pdx_parquet_write(cols, names, 5, 1, None, C.byref(out), C.byref(sz)) == 0, L.load().pdx_last_error()
    data = C.string_at(out, sz.value)
    L.load().pdx_parquet_free_blob(out)
    t = pq.read_table(io.BytesIO(data))
    t.validate(full=True)
    assert [str(x) for x in t.schema.types] == ["int64", "double", "bool", "timestamp[ns]", "uint64"] and t.schema.names == ["a", "f", "flag", "when", "u"]
    assert t.num_rows == n and pq.ParquetFile(io.BytesIO(data)).metadata.num_row_groups == (1 if n else 1)
    sl = slice(off, off + n)
    col = [t.column(i).combine_chunks() for i in range(5)]
    assert np.array_equal(col[0].to_numpy(zero_copy_only=False), a[sl]) and col[0].null_count == 0
    assert col[1].null_count == int((~vf[sl]).sum()) and np.array_equal(np.asarray(col[1].is_valid()), vf[sl])
    assert np.array_equal(col[1].to_numpy(zero_copy_only=False)[vf[sl]].view(np.uint64), f[sl][vf[sl]].view(np.uint64))
    assert np.array_equal(np.asarray(col[2].is_valid()), vb[sl]) and np.array_equal(np.asarray(col[2].fill_null(False))[vb[sl]], bv[sl][vb[sl]])
    assert np.array_equal(col[3].cast(pa.int64()).to_numpy(zero_copy_only=False), ts[sl]) and np.array_equal(col[4].to_numpy(zero_copy_only=False), u[sl])
    if n:  # our own reader takes our own file (an empty table is refused, as by the reference)
        rc, h, _keep = _open(L, data)
        assert rc == 0 and L.load().pdx_parquet_num_rows(h) == n and L.load().pdx_parquet_num_columns(h) == 5
        L.load().pdx_parquet_destroy(h)


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
    ns.L, ns.K, ns.api = L, column, api
    return ns


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(MAN["cases"]))
def test_read_parquet_fixture(px, case):
    info = MAN["cases"][case]
    df = px.api.DataFrame.readParquet(bytes(Z[f"{case}/blob"]))
    assert df.names == [c["name"] for c in info["columns"]] and df.num_rows() == info["rows"] and df.index is None
    for col in info["columns"]:
        c = df[col["name"]].col
        assert c.dtype == KIND_DTYPE[col["kind"]], col
        got, ok = c.to_numpy()
        ev, eok = Z[f"{case}/{col['name']}"], Z[f"{case}/{col['name']}_valid"]
        ok = np.ones(len(got), bool) if ok is None else ok
        assert np.array_equal(ok, eok), (case, col["name"])
        assert c.null_count == col["nulls"], (case, col["name"])
        if col["kind"] == "bool":
            assert np.array_equal(got[eok], ev[eok]), (case, col["name"])
        else:
            assert np.array_equal(np.ascontiguousarray(got).view(np.uint64)[eok], np.ascontiguousarray(ev).view(np.uint64)[eok]), (case, col["name"])


@pytest.mark.gpu
def test_read_parquet_feeds_the_hot_path(px, tmp_path):
    """a file on disk -> readParquet -> group_by on the device, against the oracle"""
    import oracle as orc

    case = "pages_30000_snappy_v2"
    path = tmp_path / "frame.parquet"
    path.write_bytes(bytes(Z[f"{case}/blob"]))
    df = px.api.DataFrame.readParquet(str(path))
    keys, kvalid = Z[f"{case}/i64_lowcard"], Z[f"{case}/i64_lowcard_valid"]
    vals, vvalid = Z[f"{case}/f64_lowcard"], Z[f"{case}/f64_lowcard_valid"]
    gb = df.group_by("i64_lowcard")
    res = gb.sum("f64_lowcard")
    ids, uniq, isnull, _ = orc.group_ids(keys, kvalid)
    exp, eok = orc.groupby_agg(0, ids, len(uniq), vals, vvalid)
    got, ok = res.col.to_numpy()
    assert np.array_equal(ok if ok is not None else np.ones(len(got), bool), eok)
    assert np.array_equal(got[eok].view(np.uint64), exp[eok].view(np.uint64))
    with pytest.raises(RuntimeError, match="Failed to open"):
        px.api.DataFrame.readParquet(str(tmp_path / "absent.parquet"))


@pytest.mark.gpu
def test_load_detects_corrupt_pages_on_the_device(px):
    """payload bytes of a Snappy page / a dictionary index stream overwritten: the load fails with a message, nothing is read out of
    bounds, and the file object can be destroyed"""
    good = bytes(Z["pages_30000_snappy_v1_required/blob"])
    rng = np.random.default_rng(9)
    failures = 0
    for trial in range(12):
        bad = bytearray(good)
        at = int(rng.integers(200, len(good) // 2))
        bad[at:at + 64] = bytes(rng.integers(0, 256, 64, dtype=np.uint8))
        try:
            df = px.api.DataFrame.readParquet(bytes(bad))
            assert df.num_rows() == 30000   # (a change inside a literal run still decodes: other values, same shape)
        except RuntimeError as e:
            failures += 1
            assert "pdx_parquet" in str(e)
    assert failures >= 1


@pytest.mark.gpu
def test_to_parquet_from_device_and_round_trip(px, tmp_path):
    """DataFrame.toParquet from device columns (incl. a nullable int64, booleans, an index written as the named last column), read back by
    pyarrow AND by readParquet on the device: every bit returns"""
    pa = pytest.importorskip("pyarrow")
    pq = pytest.importorskip("pyarrow.parquet")
    api, K = px.api, px.K
    rng = np.random.default_rng(21)
    n = 300_007
    v = rng.standard_normal(n)
    vok = rng.random(n) > 0.1
    i = rng.integers(-10**15, 10**15, n)
    iok = rng.random(n) > 0.5
    b = rng.random(n) > 0.3
    ts = np.sort(rng.integers(0, 10**18, n))
    df = api.DataFrame({"v": K.Column.from_numpy(v, vok), "i": K.Column.from_numpy(i, iok), "b": K.Column.from_numpy(b)},
                       index=K.Column.from_numpy(ts, dtype=px.L.TIMESTAMP_NS))
    path = tmp_path / "out.parquet"
    df.toParquet(str(path), "when")
    t = pq.read_table(str(path))
    assert t.schema.names == ["v", "i", "b", "when"] and t.num_rows == n
    assert np.array_equal(np.asarray(t["v"].combine_chunks().is_valid()), vok) and np.array_equal(np.asarray(t["i"].combine_chunks().is_valid()), iok)
    assert np.array_equal(t["v"].combine_chunks().to_numpy(zero_copy_only=False)[vok].view(np.uint64), v[vok].view(np.uint64))
    assert np.array_equal(t["when"].combine_chunks().cast(pa.int64()).to_numpy(), ts)
    back = api.DataFrame.readParquet(str(path))
    assert back.names == ["v", "i", "b", "when"]
    for name, exp, ok in (("v", v, vok), ("i", i, iok), ("b", b, None), ("when", ts, None)):
        got, gok = back[name].col.to_numpy()
        assert (gok is None and ok is None) or np.array_equal(gok, ok), name
        sel = slice(None) if ok is None else ok
        assert np.array_equal(np.asarray(got)[sel], np.asarray(exp)[sel]), name


def _snappy_patterns(rng, n):
    """int64 / float64 columns whose PLAIN pages make Snappy emit every element shape the decoder has: long literals (random bits), a
    literal + a copy per value (small integers: the ingest bench's keys), overlapping copies with offsets below their length (runs of one
    byte / one value), copies with two-byte offsets that reach back tens of kilobytes -- across the decoder's 16 KB output tiles and
    4 KB stream windows -- and chains of copies of copies (a period repeated through the page)."""
    period = np.arange(5000, dtype=np.int64) * 7919 + 12345
    blocks = np.concatenate([rng.integers(0, 2 ** 62, n // 8), np.zeros(n // 8, np.int64), np.tile(period, n // 8 // 5000 + 1)[:n // 8],
                             rng.integers(0, 1 << 24, n // 8), np.full(n // 8, 0x0101010101010101, np.int64), np.arange(n // 8, dtype=np.int64),
                             np.repeat(rng.integers(0, 2 ** 40, n // 8 // 37 + 1), 37)[:n // 8], rng.integers(-3, 3, n - 7 * (n // 8))]).astype(np.int64)
    return {"random_bits": rng.integers(-(2 ** 62), 2 ** 62, n), "zeros": np.zeros(n, np.int64), "small_ints": rng.integers(0, 1 << 24, n),
            "arange": np.arange(n, dtype=np.int64) * 3, "period_40k_bytes": np.tile(period, n // 5000 + 1)[:n], "one_byte": np.full(n, 0x0101010101010101, np.int64),
            "blocks": blocks, "normal_f64": rng.standard_normal(n), "few_values_f64": rng.choice(rng.standard_normal(300), n)}


@pytest.mark.gpu
@pytest.mark.parametrize("page_bytes,use_dict,version", [(1 << 20, False, "1.0"), (8 << 20, False, "2.0"), (1 << 14, False, "1.0"), (1 << 20, True, "2.0")])
def test_snappy_decoder_element_shapes(px, monkeypatch, page_bytes, use_dict, version):
    """pyarrow-written Snappy pages of the patterns above -> device columns, bit for bit; the workgroup-parallel decoder (pointer-jumping
    parse, pointer-doubling copies) and the wave-per-page one it replaced (PDX_PQ_SNAPPY_WAVE=1) must agree with the source"""
    pa = pytest.importorskip("pyarrow")
    pq = pytest.importorskip("pyarrow.parquet")
    import io
    n = 600_011
    cols = _snappy_patterns(np.random.default_rng(page_bytes + use_dict), n)
    sink = io.BytesIO()
    pq.write_table(pa.table(cols), sink, compression="snappy", use_dictionary=use_dict, data_page_size=page_bytes, row_group_size=n,
                   data_page_version=version)
    blob = sink.getvalue()
    # ("pieces": the file goes up in 1 MB pieces and the pages of a piece are decoded on a side stream while the next piece is copied -- the
    #  form files of >= 64 MB take by default)
    for wave in ("0", "1", "pieces"):
        monkeypatch.setenv("PDX_PQ_SNAPPY_WAVE", "1" if wave == "1" else "0")
        if wave == "pieces":
            monkeypatch.setenv("PDX_PQ_UPLOAD_PIECE_MB", "1")
        df = px.api.DataFrame.readParquet(blob)
        assert df.num_rows() == n
        for name, src in cols.items():
            got, ok = df[name].col.to_numpy()
            assert ok is None or ok.all(), name
            assert np.array_equal(np.ascontiguousarray(got).view(np.uint64), np.ascontiguousarray(src).view(np.uint64)), (name, wave)


@pytest.mark.gpu
def test_snappy_decoders_agree_on_damaged_streams(px, monkeypatch):
    """A few bytes of a Snappy file flipped at random, 160 times: the workgroup-parallel decoder and the wave-per-page one must come to the
    same verdict -- both refuse the file, or both decode it to the same bits (a flip inside a literal run changes values, not the shape).
    Nothing may fault or hang: every length / offset of the damaged stream is checked against both buffers before it is used."""
    pa = pytest.importorskip("pyarrow")
    pq = pytest.importorskip("pyarrow.parquet")
    import io
    n = 120_000
    rng = np.random.default_rng(77)
    cols = {k: v for k, v in _snappy_patterns(rng, n).items() if k in ("small_ints", "blocks", "period_40k_bytes", "few_values_f64")}
    sink = io.BytesIO()
    pq.write_table(pa.table(cols), sink, compression="snappy", use_dictionary=False, data_page_size=1 << 16, row_group_size=n)
    good = sink.getvalue()
    verdicts = {"both_refuse": 0, "both_decode": 0}
    for trial in range(160):
        bad = bytearray(good)
        for _ in range(int(rng.integers(1, 5))):
            at = int(rng.integers(64, len(good) - 4096))  # (the footer stays: the damage is in the pages and their headers)
            bad[at] = int(rng.integers(0, 256))
        outs = []
        for wave in ("0", "1"):
            monkeypatch.setenv("PDX_PQ_SNAPPY_WAVE", wave)
            try:
                df = px.api.DataFrame.readParquet(bytes(bad))
                outs.append([np.ascontiguousarray(df[name].col.to_numpy()[0]).view(np.uint64).copy() for name in cols])
            except RuntimeError as e:
                assert "pdx_parquet" in str(e)
                outs.append(None)
        assert (outs[0] is None) == (outs[1] is None), (trial, "one decoder refused what the other accepted")
        if outs[0] is None:
            verdicts["both_refuse"] += 1
        else:
            verdicts["both_decode"] += 1
            for a, b, name in zip(outs[0], outs[1], cols):
                assert np.array_equal(a, b), (trial, name)
    assert verdicts["both_refuse"] >= 10 and verdicts["both_decode"] >= 10, verdicts
