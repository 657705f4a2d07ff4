#!/usr/bin/env python3
"""Generate tests/golden/parquet_fixtures.npz -- Parquet files written by Arrow C++ 25.0.0 (pyarrow.parquet) + the columns pyarrow reads
back from them, for the device-side reader behind DataFrame::readParquet (reference src/dataframe.cpp:646-683).

TEST INFRASTRUCTURE.  Cases cross the codec (UNCOMPRESSED / SNAPPY), dictionary encoding on / off, data page versions 1.0 / 2.0,
columns with / without nulls, every column type on the path (int8..64, uint8..64, float32/64, bool, timestamp in ms / us / ns), small
pages (many pages per chunk, a page boundary inside a run of nulls), a dictionary that outgrows its page limit (PLAIN fallback pages
behind dictionary pages) -- and files the reader must REFUSE by name: other codecs and encodings, strings, nested columns, decimals,
dates, INT96 timestamps, several row groups, an empty table.

Run:  python oracle/gen_golden_parquet.py
"""
import io
import json
import os
import sys

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "parquet_fixtures.npz")
store, manifest = {}, {"arrow_version": pa.__version__, "cases": {}, "rejects": {}}

KIND = {"int64": "i64", "uint64": "u64", "double": "f64", "bool": "bool"}


def table(rng, n, nulls, kinds):
    cols = {}
    for k in kinds:
        ok = (rng.random(n) > 0.2) if nulls else np.ones(n, bool)
        if n > 20 and nulls:
            ok[5:17] = False  # a run of nulls (RLE run of zeros in the definition levels)
        m = ~ok
        if k == "i64":
            a = pa.array(rng.integers(-2**62, 2**62, n), type=pa.int64(), mask=m)
        elif k == "i64_lowcard":
            a = pa.array(rng.integers(0, 7, n) * 1_000_003 - 5, type=pa.int64(), mask=m)
        elif k == "u64":
            a = pa.array(rng.integers(0, 2**63, n).astype(np.uint64) * np.uint64(2) + np.uint64(1), type=pa.uint64(), mask=m)
        elif k in ("i32", "i16", "i8"):
            bits = int(k[1:])
            a = pa.array(rng.integers(-2**(bits - 1), 2**(bits - 1), n), type={"i32": pa.int32(), "i16": pa.int16(), "i8": pa.int8()}[k], mask=m)
        elif k in ("u32", "u16", "u8"):
            bits = int(k[1:])
            a = pa.array(rng.integers(0, 2**bits, n), type={"u32": pa.uint32(), "u16": pa.uint16(), "u8": pa.uint8()}[k], mask=m)
        elif k == "f64":
            v = rng.standard_normal(n)
            if n > 4:
                v[1], v[2], v[3] = np.nan, -0.0, np.inf
            a = pa.array(v, type=pa.float64(), mask=m)
        elif k == "f64_lowcard":
            a = pa.array(rng.integers(0, 5, n) * 0.25 - 0.5, type=pa.float64(), mask=m)
        elif k == "f32":
            a = pa.array(rng.standard_normal(n).astype(np.float32), type=pa.float32(), mask=m)
        elif k == "bool":
            a = pa.array(rng.random(n) > 0.4, type=pa.bool_(), mask=m)
        elif k.startswith("ts_"):
            unit = k[3:]
            a = pa.array(rng.integers(0, 2_000_000_000, n) * {"ns": 1_000_000_007, "us": 1_000_003, "ms": 1_009}[unit], type=pa.timestamp(unit), mask=m)
        else:
            raise ValueError(k)
        cols[k] = a
    return pa.table(cols)


def expected(case, tbl):
    """what the reader must produce: 8-byte values (timestamps in ns) + valid flags, from pyarrow's own read of the bytes"""
    cols = []
    for name in tbl.column_names:
        a = tbl[name].combine_chunks()
        t = a.type
        valid = np.array(a.is_valid())
        if pa.types.is_timestamp(t):
            v = a.cast(pa.timestamp("ns")).cast(pa.int64()).fill_null(0).to_numpy(zero_copy_only=False).astype(np.int64)
            kind = "ts"
        elif pa.types.is_boolean(t):
            v = a.fill_null(False).to_numpy(zero_copy_only=False).astype(bool)
            kind = "bool"
        elif pa.types.is_floating(t):
            v = a.cast(pa.float64()).fill_null(0.0).to_numpy(zero_copy_only=False).astype(np.float64)
            kind = "f64"
        elif t == pa.uint64():
            v = a.fill_null(0).to_numpy(zero_copy_only=False).astype(np.uint64)
            kind = "u64"
        else:
            v = a.cast(pa.int64()).fill_null(0).to_numpy(zero_copy_only=False).astype(np.int64)
            kind = "i64"
        store[f"{case}/{name}"] = v
        store[f"{case}/{name}_valid"] = valid
        cols.append({"name": name, "kind": kind, "nulls": int((~valid).sum())})
    return cols


def put(case, tbl, **opts):
    buf = io.BytesIO()
    pq.write_table(tbl, buf, **opts)
    blob = buf.getvalue()
    back = pq.read_table(io.BytesIO(blob))
    md = pq.ParquetFile(io.BytesIO(blob)).metadata
    assert md.num_row_groups == 1
    store[f"{case}/blob"] = np.frombuffer(blob, np.uint8)
    pages = {}
    manifest["cases"][case] = {"rows": tbl.num_rows, "columns": expected(case, back), "bytes": len(blob), "options": {k: str(v) for k, v in opts.items()},
                               "encodings": {md.row_group(0).column(i).path_in_schema: list(md.row_group(0).column(i).encodings) for i in range(md.num_columns)}}


def reject(case, tbl, message, **opts):
    buf = io.BytesIO()
    pq.write_table(tbl, buf, **opts)
    store[f"{case}/blob"] = np.frombuffer(buf.getvalue(), np.uint8)
    manifest["rejects"][case] = message


ALL = ["i64", "u64", "i32", "i16", "i8", "u32", "u16", "u8", "f64", "f32", "bool", "ts_ns", "ts_us", "ts_ms", "i64_lowcard", "f64_lowcard"]

if __name__ == "__main__":
    rng = np.random.default_rng(20260401)
    for n in (1, 7):
        for nulls in (False, True):
            put(f"tiny_{n}_{int(nulls)}", table(rng, n, nulls, ALL), compression="NONE")
            put(f"tiny_{n}_{int(nulls)}_snappy_v2", table(rng, n, nulls, ALL), compression="SNAPPY", data_page_version="2.0")
    for comp in ("NONE", "SNAPPY"):
        for use_dict in (True, False):
            for ver in ("1.0", "2.0"):
                for nulls in (False, True):
                    put(f"mix_1000_{comp.lower()}_{'dict' if use_dict else 'plain'}_v{ver[0]}_{int(nulls)}", table(rng, 1000, nulls, ALL), compression=comp,
                        use_dictionary=use_dict, data_page_version=ver, data_page_size=1024)
    # many pages per chunk; the dictionary of the random int64 column outgrows its limit: PLAIN pages follow the dictionary-encoded ones
    big = ["i64", "f64_lowcard", "bool", "i64_lowcard"]
    put("pages_30000_none_v1", table(rng, 30000, True, big), compression="NONE", data_page_size=16384, dictionary_pagesize_limit=32768)
    put("pages_30000_snappy_v2", table(rng, 30000, True, big), compression="SNAPPY", data_page_size=16384, data_page_version="2.0",
        dictionary_pagesize_limit=32768)
    put("pages_30000_snappy_v1_required", pa.table({"a": pa.array(rng.integers(0, 1000, 30000) * 3, type=pa.int64()),
                                                     "b": pa.array(np.round(rng.standard_normal(30000), 1), type=pa.float64())},
                                                    schema=pa.schema([pa.field("a", pa.int64(), nullable=False), pa.field("b", pa.float64(), nullable=False)])),
        compression="SNAPPY", data_page_size=8192)
    # all-null and constant columns (one-entry dictionaries: bit width 0)
    put("constant_and_all_null", pa.table({"c": pa.array([42] * 500, type=pa.int64()), "z": pa.array([None] * 500, type=pa.float64()),
                                           "t": pa.array([True] * 500, type=pa.bool_())}), compression="SNAPPY")
    # ---- files the reader must refuse, each with the words its message has to carry
    t = table(rng, 100, True, ["i64", "f64"])
    reject("codec_gzip", t, "GZIP", compression="GZIP")
    reject("codec_zstd", t, "ZSTD", compression="ZSTD")
    reject("codec_lz4", t, "LZ4", compression="LZ4")
    reject("enc_delta", t, "DELTA_BINARY_PACKED", use_dictionary=False, column_encoding={"i64": "DELTA_BINARY_PACKED", "f64": "PLAIN"})
    reject("enc_byte_stream_split", t, "BYTE_STREAM_SPLIT", use_dictionary=False, column_encoding={"i64": "PLAIN", "f64": "BYTE_STREAM_SPLIT"})
    reject("string_column", pa.table({"s": pa.array(["a", "b", None]), "x": pa.array([1, 2, 3])}), "'s'")
    reject("list_column", pa.table({"l": pa.array([[1, 2], [], None], type=pa.list_(pa.int64()))}), "nested")
    reject("decimal_column", pa.table({"d": pa.array([1, 2, 3], type=pa.int32()).cast(pa.decimal128(12, 2))}), "DECIMAL")
    reject("date32_column", pa.table({"d": pa.array([1, 2, 3], type=pa.int32()).cast(pa.date32())}), "DATE")
    reject("int96_timestamps", pa.table({"t": pa.array([1, 2, 3], type=pa.timestamp("ns"))}), "INT96", use_deprecated_int96_timestamps=True)
    reject("two_row_groups", table(rng, 100, False, ["i64"]), "single record batch", row_group_size=50)
    reject("empty_table", table(rng, 0, False, ["i64", "f64"]), "empty parquet table")
    store["manifest"] = np.array(json.dumps(manifest))
    np.savez_compressed(OUT, **store)
    print(f"wrote {OUT}: {len(manifest['cases'])} cases + {len(manifest['rejects'])} rejects, {os.path.getsize(OUT) / 1e6:.2f} MB")
    for k, v in manifest["cases"].items():
        if k.startswith("pages") or k.endswith("_1") and "mix" in k and "v1" in k:
            print(k, v["bytes"], v["encodings"])
