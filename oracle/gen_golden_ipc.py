#!/usr/bin/env python3
"""Generate tests/golden/ipc_fixtures.npz -- Arrow IPC streams written by pyarrow (Arrow C++ 25.0.0), the byte layout
DataFrame::toBinary produces and DataFrame::readBinary consumes (reference src/dataframe.cpp:726-791), with the decoded
column contents beside them.  TEST INFRASTRUCTURE.  Run: python oracle/gen_golden_ipc.py"""
import json
import os

import numpy as np
import pyarrow as pa

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "ipc_fixtures.npz")
store, manifest = {}, {"arrow_version": pa.__version__, "cases": {}}


def stream_bytes(batches, schema, metadata=None, **opts):
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, schema, options=pa.ipc.IpcWriteOptions(**opts)) as w:
        for b in batches:
            w.write_batch(b, custom_metadata=metadata)
    return np.frombuffer(sink.getvalue().to_pybytes(), np.uint8)


def expect(case, batch):
    cols = []
    for i, f in enumerate(batch.schema):
        a = batch.column(i)
        valid = np.array([x is not None for x in a.to_pylist()], bool)
        t = f.type
        if pa.types.is_boolean(t):
            kind, vals = "bool", np.asarray(a.fill_null(False).to_numpy(zero_copy_only=False)).astype(bool)
        elif pa.types.is_floating(t):
            kind, vals = "f64", np.asarray(a.fill_null(0).to_numpy(zero_copy_only=False)).astype(np.float64)
        elif pa.types.is_timestamp(t) or pa.types.is_date64(t):
            kind, vals = "ts", np.asarray(a.cast(pa.timestamp("ns")).cast(pa.int64()).fill_null(0).to_numpy(zero_copy_only=False)).astype(np.int64)
        elif pa.types.is_unsigned_integer(t) and t.bit_width == 64:
            kind, vals = "u64", np.asarray(a.fill_null(0).to_numpy(zero_copy_only=False)).astype(np.uint64)
        else:
            kind, vals = "i64", np.asarray(a.fill_null(0).to_numpy(zero_copy_only=False)).astype(np.int64)
        store[f"{case}/col{i}"], store[f"{case}/valid{i}"] = vals, valid
        cols.append({"name": f.name, "kind": kind, "nulls": int((~valid).sum())})
    return cols


def main():
    rng = np.random.default_rng(20260301)
    for n in (0, 1, 9, 1000):
        m = lambda p: (rng.random(n) < p) if n else None  # noqa: E731
        batch = pa.record_batch({
            "i64": pa.array(rng.integers(-2**62, 2**62, n, dtype=np.int64)),
            "f64": pa.array(rng.standard_normal(n), mask=m(0.15)),
            "flag": pa.array(rng.random(n) < 0.5, mask=m(0.1)),
            "u64": pa.array(rng.integers(0, 2**63, n, dtype=np.uint64) * 2),
            "ts_ns": pa.array(rng.integers(0, 2 * 10**18, n, dtype=np.int64)).cast(pa.timestamp("ns")),
            "i32": pa.array(rng.integers(-2**31, 2**31, n).astype(np.int32), mask=m(0.2)),
            "i8": pa.array(rng.integers(-128, 128, n).astype(np.int8)),
            "u16": pa.array(rng.integers(0, 65536, n).astype(np.uint16)),
            "u32": pa.array(rng.integers(0, 2**32, n).astype(np.uint32)),
            "f32": pa.array(rng.standard_normal(n).astype(np.float32), mask=m(0.1)),
            "ts_us": pa.array(rng.integers(-10**15, 10**15, n, dtype=np.int64)).cast(pa.timestamp("us", tz="UTC")),
            "ts_s": pa.array(rng.integers(0, 2 * 10**9, n, dtype=np.int64)).cast(pa.timestamp("s")),
            "d64": pa.array(rng.integers(0, 20000, n, dtype=np.int64) * 86400000).cast(pa.date64()),
        })
        case = f"mixed_{n}"
        store[f"{case}/blob"] = stream_bytes([batch], batch.schema, {"source": "pyarrow", "rows": str(n)})
        manifest["cases"][case] = {"columns": expect(case, batch), "rows": n, "metadata": {"source": "pyarrow", "rows": str(n)}}
    # the reference's shape: numeric columns + the index written as the last int64 column ("toBinary(index = ...)")
    n = 500
    batch = pa.record_batch({"price": pa.array(rng.random(n) * 100), "volume": pa.array(rng.integers(0, 10**6, n, dtype=np.int64)),
                             "__index__": pa.array(946684800 * 10**9 + np.arange(n, dtype=np.int64) * 60 * 10**9)})
    store["with_index/blob"] = stream_bytes([batch], batch.schema)
    manifest["cases"]["with_index"] = {"columns": expect("with_index", batch), "rows": n, "metadata": {}, "index": "__index__"}
    # a sliced batch: buffers are truncated / re-based by the writer, validity offsets are not byte aligned in the source
    big = pa.record_batch({"a": pa.array(rng.standard_normal(300), mask=rng.random(300) < 0.3), "b": pa.array(rng.random(300) < 0.5, mask=rng.random(300) < 0.2)})
    sl = big.slice(13, 200)
    store["sliced/blob"] = stream_bytes([sl], sl.schema)
    manifest["cases"]["sliced"] = {"columns": expect("sliced", sl), "rows": 200, "metadata": {}}
    # streams the reader must refuse
    two = pa.record_batch({"x": pa.array([1, 2, 3])})
    store["reject_two_batches/blob"] = stream_bytes([two, two], two.schema)
    s = pa.record_batch({"x": pa.array([1, 2, 3]), "name": pa.array(["a", "b", "c"])})
    store["reject_string/blob"] = stream_bytes([s], s.schema)
    d = pa.record_batch({"x": pa.array(["a", "b", "a"]).dictionary_encode()})
    store["reject_dictionary/blob"] = stream_bytes([d], d.schema)
    c = pa.record_batch({"x": pa.array(np.arange(5000, dtype=np.int64))})
    if pa.Codec.is_available("lz4"):
        store["reject_compressed/blob"] = stream_bytes([c], c.schema, compression="lz4")
    empty = pa.record_batch({"x": pa.array([1, 2, 3])})
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, empty.schema):
        pass
    store["reject_no_batch/blob"] = np.frombuffer(sink.getvalue().to_pybytes(), np.uint8)
    manifest["rejects"] = {"reject_two_batches": "Always Assume Single RecordBatch", "reject_string": "field 'name'", "reject_dictionary": "dictionary",
                           "reject_compressed": "compressed", "reject_no_batch": "Always Assume Single RecordBatch"}
    store["manifest"] = np.array(json.dumps(manifest))
    np.savez_compressed(OUT, **store)
    print(f"wrote {OUT}: {os.path.getsize(OUT)} bytes")


if __name__ == "__main__":
    main()
