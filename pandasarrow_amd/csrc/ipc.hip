// ipc.hip -- Arrow IPC streams <-> device-resident columns (the step either side of the hot path).
//
// Replaces, for the column types of this path, DataFrame::readBinary / DataFrame::toBinary (reference src/dataframe.cpp:726-791:
// arrow::ipc::RecordBatchStreamReader::Open(...)->ToRecordBatches() and arrow::ipc::MakeStreamWriter + WriteRecordBatch).
// An IPC record batch body IS the Arrow layout the kernels read (validity bitmap + contiguous values per column, 8-byte aligned
// buffers), so reading is: parse the two flatbuffer messages on the host (microseconds), ONE host->device copy of the whole
// body into one device allocation, and the columns are pointers into it -- no per-value work, no host-side Arrow arrays.
// Narrow numeric columns (int8..int32, uint8..uint32, float32) are widened to the path's 8-byte types by one small kernel.
// Writing is the mirror image: device->host copies of the buffers into a stream laid out by a ~100-line flatbuffer emitter.
//
// Format (Arrow columnar format 1.x, little endian, metadata version V5): a stream is a sequence of encapsulated messages
//   <0xFFFFFFFF> <int32 metadata size> <flatbuffer Message, padded to 8> <body, padded to 8>
// Schema first, then RecordBatch messages, then the end-of-stream marker <0xFFFFFFFF> <0>.  Legacy (pre-0.15) framing without the
// continuation word is accepted.  Not supported (PDX_NOT_IMPLEMENTED, naming the field): dictionary batches, body compression,
// big-endian streams, and column types outside {bool, (u)int8..64, float32/64, timestamp, date64} (strings, nested, decimals).
#include <string.h>
#include <memory>
#include <string>
#include <vector>
#include "pdx_common.hpp"

namespace pdx {
namespace {

// ---------------------------------------------------------------- flatbuffer reading (bounds-checked, tables / vectors / strings)
struct Fb {
  const uint8_t* p;
  size_t n;
  bool ok = true;
  template <typename T>
  T rd(size_t at) {
    if (at > n || n - at < sizeof(T)) {  // (written so that a wild offset cannot wrap around)
      ok = false;
      return T(0);
    }
    T v;
    memcpy(&v, p + at, sizeof(T));
    return v;
  }
  size_t root() { return rd<uint32_t>(0); }
  // position of field `id` of the table at `t`, or 0 when absent
  size_t field(size_t t, int id) {
    const int32_t so = rd<int32_t>(t);
    const size_t vt = (size_t)((int64_t)t - so);
    const uint16_t vsize = rd<uint16_t>(vt);
    const size_t slot = 4 + 2 * (size_t)id;
    if (slot + 2 > vsize) return 0;
    const uint16_t off = rd<uint16_t>(vt + slot);
    return off ? t + off : 0;
  }
  template <typename T>
  T scalar(size_t t, int id, T dflt) {
    size_t f = field(t, id);
    return f ? rd<T>(f) : dflt;
  }
  size_t indirect(size_t t, int id) {  // table / vector / string referenced by an offset field; 0 when absent
    size_t f = field(t, id);
    return f ? f + rd<uint32_t>(f) : 0;
  }
  uint32_t vec_len(size_t v) { return v ? rd<uint32_t>(v) : 0; }
  size_t vec_table(size_t v, uint32_t i) {  // element i of a vector of tables
    size_t e = v + 4 + 4 * (size_t)i;
    return e + rd<uint32_t>(e);
  }
  std::string str(size_t s) {
    if (!s) return std::string();
    uint32_t len = rd<uint32_t>(s);
    if (!ok || s + 4 + len > n) {
      ok = false;
      return std::string();
    }
    return std::string(reinterpret_cast<const char*>(p + s + 4), len);
  }
};

enum { kMsgSchema = 1, kMsgDictionary = 2, kMsgRecordBatch = 3 };
enum { kTyInt = 2, kTyFloat = 3, kTyBool = 6, kTyDate = 8, kTyTimestamp = 10 };

struct FieldInfo {
  std::string name;
  int type_id = 0;       // Arrow flatbuffer Type union tag
  int bit_width = 64;    // Int / FloatingPoint width
  bool is_signed = true;
  int ts_unit = 3;       // Timestamp unit: 0 s, 1 ms, 2 us, 3 ns
  std::string timezone;
  int pdx_dtype = -1;    // what the column becomes on the device
  bool nullable = true;
  int nchildren = 0;
  // filled from the RecordBatch message
  int64_t length = 0, null_count = 0;
  int64_t validity_off = 0, validity_len = 0, values_off = 0, values_len = 0;
};

struct Message {
  int header_type = 0;
  size_t header = 0;      // table position inside the flatbuffer
  int64_t body_length = 0;
  const uint8_t* meta = nullptr;
  size_t meta_size = 0;
  const uint8_t* body = nullptr;
  size_t custom_metadata = 0;
};

// next encapsulated message at *pos; returns 0 = ok, 1 = end of stream, <0 = malformed
int next_message(const uint8_t* blob, size_t size, size_t* pos, Message* m) {
  size_t at = *pos;
  if (at + 4 > size) return 1;  // a stream may simply end
  uint32_t w;
  memcpy(&w, blob + at, 4);
  int32_t msize;
  if (w == 0xFFFFFFFFu) {
    if (at + 8 > size) return -1;
    memcpy(&msize, blob + at + 4, 4);
    at += 8;
  } else {  // legacy framing: the first word is the size
    msize = (int32_t)w;
    at += 4;
  }
  if (msize == 0) return 1;
  if (msize < 0 || at + (size_t)msize > size) return -1;
  Fb fb{blob + at, (size_t)msize};
  const size_t t = fb.root();
  m->meta = blob + at;
  m->meta_size = (size_t)msize;
  m->header_type = fb.scalar<uint8_t>(t, 1, 0);
  m->header = fb.indirect(t, 2);
  m->body_length = fb.scalar<int64_t>(t, 3, 0);
  m->custom_metadata = fb.indirect(t, 4);
  if (!fb.ok || m->body_length < 0) return -1;
  at += (size_t)msize;
  if (at + (size_t)m->body_length > size) return -1;
  m->body = blob + at;
  at += (size_t)m->body_length;
  *pos = at;
  return 0;
}

__global__ void k_widen(const void* __restrict__ in, int bit_width, int is_signed, int is_float, int64_t mul, int64_t n, void* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (is_float) {
      static_cast<double*>(out)[i] = (double)static_cast<const float*>(in)[i];
      continue;
    }
    long long v;
    switch (bit_width) {
      case 8: v = is_signed ? (long long)static_cast<const int8_t*>(in)[i] : (long long)static_cast<const uint8_t*>(in)[i]; break;
      case 16: v = is_signed ? (long long)static_cast<const int16_t*>(in)[i] : (long long)static_cast<const uint16_t*>(in)[i]; break;
      case 32: v = is_signed ? (long long)static_cast<const int32_t*>(in)[i] : (long long)static_cast<const uint32_t*>(in)[i]; break;
      default: v = static_cast<const long long*>(in)[i]; break;
    }
    static_cast<long long*>(out)[i] = v * mul;
  }
}

}  // namespace
}  // namespace pdx

using namespace pdx;

struct pdx_ipc_frame {
  std::vector<FieldInfo> fields;
  int64_t num_rows = 0;
  std::vector<std::pair<std::string, std::string>> metadata;  // the record batch message's custom metadata (toBinary's `metadata`)
  const uint8_t* body = nullptr;  // borrowed: inside the caller's blob (until pdx_ipc_load)
  int64_t body_length = 0;
  // device side (after pdx_ipc_load)
  void* dev_body = nullptr;
  std::vector<void*> widened;     // per field: own 8-byte values buffer, or nullptr when the body's buffer is used in place
  bool loaded = false;
  hipStream_t stream = nullptr;
  ~pdx_ipc_frame() {
    if (!dev_body && widened.empty()) return;  // parsed only: nothing on the device
    StreamNote note(stream);
    std::vector<void*> all(widened);
    all.push_back(dev_body);
    pool_free_many(all.data(), (int)all.size());
  }
};

namespace pdx {
namespace {

int parse_schema(Fb& fb, size_t schema, std::vector<FieldInfo>* out) {
  if (fb.scalar<int16_t>(schema, 0, 0) != 0) return fail(PDX_NOT_IMPLEMENTED, "pdx_ipc_open: big-endian streams are not supported");
  const size_t fields = fb.indirect(schema, 1);
  const uint32_t nf = fb.vec_len(fields);
  for (uint32_t i = 0; i < nf; ++i) {
    const size_t f = fb.vec_table(fields, i);
    FieldInfo fi;
    fi.name = fb.str(fb.indirect(f, 0));
    fi.nullable = fb.scalar<uint8_t>(f, 1, 0) != 0;
    fi.type_id = fb.scalar<uint8_t>(f, 2, 0);
    const size_t ty = fb.indirect(f, 3);
    if (fb.indirect(f, 4)) return fail(PDX_NOT_IMPLEMENTED, "pdx_ipc_open: field '" + fi.name + "' is dictionary encoded (not supported)");
    fi.nchildren = (int)fb.vec_len(fb.indirect(f, 5));
    switch (fi.type_id) {
      case kTyInt:
        fi.bit_width = ty ? fb.scalar<int32_t>(ty, 0, 0) : 0;
        fi.is_signed = ty ? fb.scalar<uint8_t>(ty, 1, 0) != 0 : false;
        if (fi.bit_width != 8 && fi.bit_width != 16 && fi.bit_width != 32 && fi.bit_width != 64)
          return fail(PDX_NOT_IMPLEMENTED, "pdx_ipc_open: field '" + fi.name + "': unsupported integer width");
        fi.pdx_dtype = (fi.is_signed || fi.bit_width < 64) ? PDX_INT64 : PDX_UINT64;
        break;
      case kTyFloat: {
        const int prec = ty ? fb.scalar<int16_t>(ty, 0, 0) : 0;
        if (prec != 1 && prec != 2) return fail(PDX_NOT_IMPLEMENTED, "pdx_ipc_open: field '" + fi.name + "': half floats are not supported");
        fi.bit_width = prec == 2 ? 64 : 32;
        fi.pdx_dtype = PDX_FLOAT64;
        break;
      }
      case kTyBool:
        fi.bit_width = 1;
        fi.pdx_dtype = PDX_BOOL;
        break;
      case kTyTimestamp:
        fi.ts_unit = ty ? fb.scalar<int16_t>(ty, 0, 0) : 0;
        fi.timezone = ty ? fb.str(fb.indirect(ty, 1)) : std::string();
        fi.bit_width = 64;
        fi.pdx_dtype = PDX_TIMESTAMP_NS;
        break;
      case kTyDate:  // DateUnit: DAY = 0 (32 bit), MILLISECOND = 1 (64 bit, the default): only date64 maps onto the path
        if ((ty ? fb.scalar<int16_t>(ty, 0, 1) : 1) == 0) return fail(PDX_NOT_IMPLEMENTED, "pdx_ipc_open: field '" + fi.name + "': date32 is not supported");
        fi.ts_unit = 1;
        fi.bit_width = 64;
        fi.pdx_dtype = PDX_TIMESTAMP_NS;
        break;
      default:
        return fail(PDX_NOT_IMPLEMENTED, "pdx_ipc_open: field '" + fi.name + "' has Arrow type id " + std::to_string(fi.type_id) +
                                             " (only bool / integers / float32 / float64 / timestamp / date64 columns are on this path)");
    }
    if (fi.nchildren) return fail(PDX_NOT_IMPLEMENTED, "pdx_ipc_open: field '" + fi.name + "' is nested (not supported)");
    out->push_back(fi);
  }
  if (!fb.ok) return fail(PDX_INVALID, "pdx_ipc_open: malformed Schema message");
  return PDX_OK;
}

int parse_record_batch(Fb& fb, const Message& m, pdx_ipc_frame* fr) {
  const size_t rb = m.header;
  fr->num_rows = fb.scalar<int64_t>(rb, 0, 0);
  const size_t nodes = fb.indirect(rb, 1), bufs = fb.indirect(rb, 2);
  if (fb.indirect(rb, 3)) return fail(PDX_NOT_IMPLEMENTED, "pdx_ipc_open: compressed record batch bodies are not supported");
  const uint32_t nn = fb.vec_len(nodes), nb = fb.vec_len(bufs);
  if (nn != fr->fields.size() || nb != 2 * fr->fields.size()) return fail(PDX_INVALID, "pdx_ipc_open: record batch does not match the schema");
  for (uint32_t i = 0; i < nn; ++i) {
    FieldInfo& f = fr->fields[i];
    f.length = fb.rd<int64_t>(nodes + 4 + 16 * (size_t)i);
    f.null_count = fb.rd<int64_t>(nodes + 4 + 16 * (size_t)i + 8);
    f.validity_off = fb.rd<int64_t>(bufs + 4 + 16 * (size_t)(2 * i));
    f.validity_len = fb.rd<int64_t>(bufs + 4 + 16 * (size_t)(2 * i) + 8);
    f.values_off = fb.rd<int64_t>(bufs + 4 + 16 * (size_t)(2 * i + 1));
    f.values_len = fb.rd<int64_t>(bufs + 4 + 16 * (size_t)(2 * i + 1) + 8);
    // every number below comes from an untrusted stream: no sum or product may wrap.  A buffer lies inside the body iff
    // 0 <= off <= body and 0 <= len <= body - off; the row count is bounded by the body's bits before `need` is formed.
    const int64_t body = m.body_length;
    auto inside = [body](int64_t off, int64_t len) { return off >= 0 && len >= 0 && off <= body && len <= body - off; };
    const bool counts_ok = f.length == fr->num_rows && f.length >= 0 && f.null_count >= 0 && f.null_count <= f.length && body >= 0 &&
                           (f.length == 0 || f.length / 8 <= body);  // (>= 1 bit per row in the body: length * 8 cannot wrap below)
    const int64_t need = !counts_ok ? 0 : f.bit_width == 1 ? (f.length + 7) / 8 : f.length * (int64_t)(f.bit_width / 8);
    if (!counts_ok || !inside(f.values_off, f.values_len) || !inside(f.validity_off, f.validity_len) || f.values_len < need ||
        (f.null_count > 0 && f.validity_len < (f.length + 7) / 8) || (f.values_off & 7) || (f.validity_off & 7))
      return fail(PDX_INVALID, "pdx_ipc_open: field '" + f.name + "': buffer layout is inconsistent with the batch");
  }
  if (m.custom_metadata) {
    Fb mf{m.meta, m.meta_size};
    const uint32_t nk = mf.vec_len(m.custom_metadata);
    for (uint32_t i = 0; i < nk; ++i) {
      const size_t kv = mf.vec_table(m.custom_metadata, i);
      fr->metadata.emplace_back(mf.str(mf.indirect(kv, 0)), mf.str(mf.indirect(kv, 1)));
    }
    if (!mf.ok) return fail(PDX_INVALID, "pdx_ipc_open: malformed custom metadata");
  }
  if (!fb.ok) return fail(PDX_INVALID, "pdx_ipc_open: malformed RecordBatch message");
  fr->body = m.body;
  fr->body_length = m.body_length;
  return PDX_OK;
}

// ---------------------------------------------------------------- flatbuffer writing: front to back, children after their parents
// (every uoffset points forward, every table's vtable sits right in front of it: a valid, if unusual, flatbuffer layout)
struct FbOut {
  std::vector<uint8_t> b;
  size_t pos() const { return b.size(); }
  void align(size_t a) {
    while (b.size() % a) b.push_back(0);
  }
  template <typename T>
  void put(T v) {
    const uint8_t* q = reinterpret_cast<const uint8_t*>(&v);
    b.insert(b.end(), q, q + sizeof(T));
  }
  template <typename T>
  void set(size_t at, T v) { memcpy(b.data() + at, &v, sizeof(T)); }
  void patch(size_t field_pos) { set<uint32_t>(field_pos, (uint32_t)(pos() - field_pos)); }  // uoffset field -> the object about to be written
  // table with `nslots` vtable slots; sizes[i] = inline size of slot i (0 = absent).  Returns the position of every present slot.
  size_t table(const std::vector<int>& sizes, std::vector<size_t>* slot_pos) {
    // inline layout: soffset (4), then the slots in DESCENDING size order (keeps every scalar naturally aligned)
    std::vector<int> order;
    for (int sz : {8, 4, 2, 1})
      for (size_t i = 0; i < sizes.size(); ++i)
        if (sizes[i] == sz) order.push_back((int)i);
    std::vector<uint16_t> off(sizes.size(), 0);
    uint16_t cur = 4;
    bool has8 = false;
    for (int i : order) has8 = has8 || sizes[(size_t)i] == 8;
    if (has8) cur = 8;  // 8-byte scalars start at table + 8 (the table itself is 8-aligned below)
    for (int i : order) {
      off[(size_t)i] = cur;
      cur = (uint16_t)(cur + sizes[(size_t)i]);
    }
    const uint16_t tsize = cur, vsize = (uint16_t)(4 + 2 * sizes.size());
    // place the vtable so that the table lands on an 8-byte boundary
    align(2);
    while ((pos() + vsize) % 8) put<uint16_t>(0);
    const size_t vt = pos();
    put<uint16_t>(vsize);
    put<uint16_t>(tsize);
    for (uint16_t o : off) put<uint16_t>(o);
    const size_t t = pos();
    put<int32_t>((int32_t)(t - vt));
    b.resize(t + tsize, 0);
    slot_pos->assign(sizes.size(), 0);
    for (size_t i = 0; i < sizes.size(); ++i)
      if (sizes[i]) (*slot_pos)[i] = t + off[i];
    return t;
  }
  void string(const std::string& s) {
    align(4);
    put<uint32_t>((uint32_t)s.size());
    b.insert(b.end(), s.begin(), s.end());
    b.push_back(0);
  }
};

struct OutField {
  std::string name;
  int dtype;
};

void emit_kv_vector(FbOut& o, size_t field_pos, const std::vector<std::pair<std::string, std::string>>& kv) {
  o.align(4);
  o.patch(field_pos);
  o.put<uint32_t>((uint32_t)kv.size());
  const size_t elems = o.pos();
  for (size_t i = 0; i < kv.size(); ++i) o.put<uint32_t>(0);
  for (size_t i = 0; i < kv.size(); ++i) {
    std::vector<size_t> sp;
    o.align(4);
    // the table position is only known after FbOut::table aligned it: patch afterwards
    const size_t t = o.table({4, 4}, &sp);
    o.set<uint32_t>(elems + 4 * i, (uint32_t)(t - (elems + 4 * i)));
    o.align(4);
    o.patch(sp[0]);
    o.string(kv[i].first);
    o.align(4);
    o.patch(sp[1]);
    o.string(kv[i].second);
  }
}

// Message { version = V5 (4), header_type, header, bodyLength, custom_metadata }
std::vector<uint8_t> schema_message(const std::vector<OutField>& fields) {
  FbOut o;
  o.put<uint32_t>(0);  // root uoffset, patched below
  std::vector<size_t> msg;
  const size_t mt = o.table({2, 1, 4, 8, 0}, &msg);
  o.set<uint32_t>(0, (uint32_t)mt);
  o.set<int16_t>(msg[0], 4);
  o.set<uint8_t>(msg[1], kMsgSchema);
  o.set<int64_t>(msg[3], 0);
  // Schema { endianness (default little: omitted), fields }
  std::vector<size_t> sc;
  o.align(4);
  const size_t st = o.table({0, 4, 0, 0}, &sc);
  o.set<uint32_t>(msg[2], (uint32_t)(st - msg[2]));
  o.align(4);
  o.patch(sc[1]);
  o.put<uint32_t>((uint32_t)fields.size());
  const size_t elems = o.pos();
  for (size_t i = 0; i < fields.size(); ++i) o.put<uint32_t>(0);
  for (size_t i = 0; i < fields.size(); ++i) {
    // Field { name, nullable, type_type, type, dictionary, children, custom_metadata }
    std::vector<size_t> f;
    const size_t ft = o.table({4, 1, 1, 4, 0, 4, 0}, &f);
    o.set<uint32_t>(elems + 4 * i, (uint32_t)(ft - (elems + 4 * i)));
    o.set<uint8_t>(f[1], 1);
    const int dt = fields[i].dtype;
    const uint8_t tag = dt == PDX_FLOAT64 ? kTyFloat : dt == PDX_BOOL ? kTyBool : dt == PDX_TIMESTAMP_NS ? kTyTimestamp : kTyInt;
    o.set<uint8_t>(f[2], tag);
    o.align(4);
    o.patch(f[0]);
    o.string(fields[i].name);
    std::vector<size_t> ty;
    size_t tt;
    if (tag == kTyInt) {
      tt = o.table({4, 1}, &ty);
      o.set<int32_t>(ty[0], 64);
      o.set<uint8_t>(ty[1], dt == PDX_UINT64 ? 0 : 1);
    } else if (tag == kTyFloat) {
      tt = o.table({2}, &ty);
      o.set<int16_t>(ty[0], 2);  // DOUBLE
    } else if (tag == kTyTimestamp) {
      tt = o.table({2, 0}, &ty);
      o.set<int16_t>(ty[0], 3);  // NANOSECOND, no time zone
    } else {
      tt = o.table({}, &ty);
    }
    o.set<uint32_t>(f[3], (uint32_t)(tt - f[3]));
    o.align(4);
    o.patch(f[5]);
    o.put<uint32_t>(0);  // children: empty vector
  }
  o.align(8);
  return o.b;
}

struct OutBuffer {
  int64_t offset, length;
};
std::vector<uint8_t> record_batch_message(int64_t nrows, const std::vector<std::pair<int64_t, int64_t>>& nodes, const std::vector<OutBuffer>& bufs,
                                          int64_t body_length, const std::vector<std::pair<std::string, std::string>>& meta) {
  FbOut o;
  o.put<uint32_t>(0);
  std::vector<size_t> msg;
  const size_t mt = o.table({2, 1, 4, 8, meta.empty() ? 0 : 4}, &msg);
  o.set<uint32_t>(0, (uint32_t)mt);
  o.set<int16_t>(msg[0], 4);
  o.set<uint8_t>(msg[1], kMsgRecordBatch);
  o.set<int64_t>(msg[3], body_length);
  // RecordBatch { length, nodes, buffers }
  std::vector<size_t> rb;
  const size_t rt = o.table({8, 4, 4, 0, 0}, &rb);
  o.set<uint32_t>(msg[2], (uint32_t)(rt - msg[2]));
  o.set<int64_t>(rb[0], nrows);
  // vectors of structs: the length word sits right in front of 8-aligned elements
  o.align(8);
  o.put<uint32_t>(0);
  o.patch(rb[1]);
  o.put<uint32_t>((uint32_t)nodes.size());
  for (auto& nd : nodes) {
    o.put<int64_t>(nd.first);
    o.put<int64_t>(nd.second);
  }
  o.align(8);
  o.put<uint32_t>(0);
  o.patch(rb[2]);
  o.put<uint32_t>((uint32_t)bufs.size());
  for (auto& bf : bufs) {
    o.put<int64_t>(bf.offset);
    o.put<int64_t>(bf.length);
  }
  if (!meta.empty()) emit_kv_vector(o, msg[4], meta);
  o.align(8);
  return o.b;
}

void append_framed(std::vector<uint8_t>& out, const std::vector<uint8_t>& meta) {
  const uint32_t cont = 0xFFFFFFFFu;
  const int32_t len = (int32_t)meta.size();
  out.insert(out.end(), reinterpret_cast<const uint8_t*>(&cont), reinterpret_cast<const uint8_t*>(&cont) + 4);
  out.insert(out.end(), reinterpret_cast<const uint8_t*>(&len), reinterpret_cast<const uint8_t*>(&len) + 4);
  out.insert(out.end(), meta.begin(), meta.end());
}

}  // namespace
}  // namespace pdx

extern "C" {

int pdx_ipc_open(const void* blob, size_t size, pdx_ipc_frame** out) {
  if (!blob || !out) return fail(PDX_INVALID, "pdx_ipc_open: null argument");
  const uint8_t* p = static_cast<const uint8_t*>(blob);
  std::unique_ptr<pdx_ipc_frame> fr(new pdx_ipc_frame());
  size_t pos = 0;
  Message m;
  int rc = next_message(p, size, &pos, &m);
  if (rc != 0 || m.header_type != kMsgSchema || !m.header) return fail(PDX_INVALID, "pdx_ipc_open: not an Arrow IPC stream (no Schema message)");
  {
    Fb fb{m.meta, m.meta_size};
    PDX_TRY(parse_schema(fb, m.header, &fr->fields));
  }
  int batches = 0;
  for (;;) {
    rc = next_message(p, size, &pos, &m);
    if (rc == 1) break;
    if (rc < 0) return fail(PDX_INVALID, "pdx_ipc_open: truncated or malformed message");
    if (m.header_type == kMsgDictionary) return fail(PDX_NOT_IMPLEMENTED, "pdx_ipc_open: dictionary batches are not supported");
    if (m.header_type != kMsgRecordBatch) continue;
    if (++batches > 1)  // DataFrame::readBinary: "Always Assume Single RecordBatch." (src/dataframe.cpp:764-767)
      return fail(PDX_INVALID, "PandasArrow Cannot ReadBinary from a Table or Array of RecordBatches yet. Always Assume Single RecordBatch.");
    Fb fb{m.meta, m.meta_size};
    PDX_TRY(parse_record_batch(fb, m, fr.get()));
  }
  if (batches != 1) return fail(PDX_INVALID, "PandasArrow Cannot ReadBinary from a Table or Array of RecordBatches yet. Always Assume Single RecordBatch.");
  *out = fr.release();
  return PDX_OK;
}

int pdx_ipc_destroy(pdx_ipc_frame* fr) {
  delete fr;
  return PDX_OK;
}
int pdx_ipc_num_columns(const pdx_ipc_frame* fr) { return fr ? (int)fr->fields.size() : -1; }
int64_t pdx_ipc_num_rows(const pdx_ipc_frame* fr) { return fr ? fr->num_rows : -1; }
const char* pdx_ipc_column_name(const pdx_ipc_frame* fr, int i) {
  return (fr && i >= 0 && i < (int)fr->fields.size()) ? fr->fields[(size_t)i].name.c_str() : nullptr;
}
int pdx_ipc_num_metadata(const pdx_ipc_frame* fr) { return fr ? (int)fr->metadata.size() : -1; }
const char* pdx_ipc_metadata_key(const pdx_ipc_frame* fr, int i) {
  return (fr && i >= 0 && i < (int)fr->metadata.size()) ? fr->metadata[(size_t)i].first.c_str() : nullptr;
}
const char* pdx_ipc_metadata_value(const pdx_ipc_frame* fr, int i) {
  return (fr && i >= 0 && i < (int)fr->metadata.size()) ? fr->metadata[(size_t)i].second.c_str() : nullptr;
}

int pdx_ipc_load(pdx_ipc_frame* fr, void* stream) {
  if (!fr) return fail(PDX_INVALID, "pdx_ipc_load: null frame");
  if (fr->loaded) return PDX_OK;
  hipStream_t st = as_stream(stream);
  fr->stream = st;
  // a failed load leaves NOTHING behind (a second call starts clean instead of orphaning the first call's blocks in the pool)
  auto undo = [fr](int rc) {
    for (void*& w : fr->widened) {
      pool_free(w);
      w = nullptr;
    }
    pool_free(fr->dev_body);
    fr->dev_body = nullptr;
    return rc;
  };
  fr->widened.assign(fr->fields.size(), nullptr);
  fr->dev_body = pool_alloc((size_t)fr->body_length + 64);  // slack: kernels may read whole 64-bit bitmap words
  if (!fr->dev_body) return PDX_OOM;
  // ONE host->device copy: the body is already the Arrow layout the kernels read
  if (fr->body_length) {
    const hipError_t ec = hipMemcpyAsync(fr->dev_body, fr->body, (size_t)fr->body_length, hipMemcpyHostToDevice, st);
    if (ec != hipSuccess) return undo(hip_fail(ec, "pdx_ipc_load: copy of the record batch body"));
  }
  static const int64_t unit_to_ns[4] = {1000000000LL, 1000000LL, 1000LL, 1LL};
  for (size_t i = 0; i < fr->fields.size(); ++i) {
    const FieldInfo& f = fr->fields[i];
    const bool narrow = f.bit_width == 8 || f.bit_width == 16 || f.bit_width == 32;
    const int64_t mul = f.pdx_dtype == PDX_TIMESTAMP_NS ? unit_to_ns[f.ts_unit & 3] : 1;
    if (!narrow && mul == 1) continue;
    void* w = pool_alloc((size_t)(f.length > 0 ? f.length : 1) * 8);
    if (!w) return undo(PDX_OOM);
    fr->widened[i] = w;
    if (f.length)
      hipLaunchKernelGGL(k_widen, dim3(grid_for(f.length, 256, 4)), dim3(256), 0, st, static_cast<const uint8_t*>(fr->dev_body) + f.values_off,
                         f.bit_width, f.is_signed ? 1 : 0, f.type_id == kTyFloat ? 1 : 0, mul, f.length, w);
    const hipError_t el = hipGetLastError();
    if (el != hipSuccess) return undo(hip_fail(el, "pdx_ipc_load: k_widen"));
  }
  {
    const hipError_t es = hipStreamSynchronize(st);
    if (es != hipSuccess) return undo(hip_fail(es, "pdx_ipc_load"));
  }
  fr->body = nullptr;  // the caller may free the blob once this returns
  fr->loaded = true;
  return PDX_OK;
}

int pdx_ipc_column(const pdx_ipc_frame* fr, int i, pdx_column* out) {
  if (!fr || !out || i < 0 || i >= (int)fr->fields.size()) return fail(PDX_INVALID, "pdx_ipc_column: bad argument");
  const FieldInfo& f = fr->fields[(size_t)i];
  memset(out, 0, sizeof(*out));
  out->dtype = f.pdx_dtype;
  out->length = f.length;
  out->offset = 0;
  out->null_count = f.null_count;
  if (fr->loaded) {
    const uint8_t* base = static_cast<const uint8_t*>(fr->dev_body);
    out->validity = (f.null_count > 0 && f.validity_len > 0) ? base + f.validity_off : nullptr;
    out->values = fr->widened[(size_t)i] ? fr->widened[(size_t)i] : static_cast<const void*>(base + f.values_off);
  }
  return PDX_OK;
}

int pdx_ipc_write(const pdx_column* cols, const char* const* names, int ncols, const char* const* metadata_kv, int nmeta, int columns_on_host,
                  void* stream, void** out_blob, size_t* out_size) {
  if ((ncols > 0 && (!cols || !names)) || ncols < 0 || !out_blob || !out_size || nmeta < 0 || (nmeta > 0 && !metadata_kv))
    return fail(PDX_INVALID, "pdx_ipc_write: bad argument");
  hipStream_t st = as_stream(stream);
  std::vector<OutField> fields;
  std::vector<std::pair<int64_t, int64_t>> nodes;
  std::vector<OutBuffer> bufs;
  const int64_t nrows = ncols ? cols[0].length : 0;
  int64_t body = 0;
  auto pad8 = [](int64_t x) { return (x + 7) & ~int64_t(7); };
  struct Src {
    const uint8_t* validity;
    const uint8_t* values;
    int64_t vbytes, dbytes, voff, doff, bit_off;
  };
  std::vector<Src> srcs;
  for (int c = 0; c < ncols; ++c) {
    PDX_TRY(check_column(&cols[c], "pdx_ipc_write"));
    if (cols[c].length != nrows) return fail(PDX_INVALID, "pdx_ipc_write: all columns must have the same length");
    const int dt = cols[c].dtype;
    if (dt < PDX_INT64 || dt > PDX_TIMESTAMP_NS) return fail(PDX_INVALID, "pdx_ipc_write: unknown dtype");
    if (!names[c]) return fail(PDX_INVALID, "pdx_ipc_write: null column name");
    const bool has_v = validity_or_null(&cols[c]) != nullptr;
    fields.push_back({names[c], dt});
    Src s{};
    s.validity = has_v ? static_cast<const uint8_t*>(cols[c].validity) : nullptr;
    s.values = static_cast<const uint8_t*>(cols[c].values);
    s.vbytes = has_v ? (nrows + 7) / 8 : 0;
    s.dbytes = dt == PDX_BOOL ? (nrows + 7) / 8 : nrows * 8;
    s.bit_off = cols[c].offset;
    s.voff = body;
    body += pad8(s.vbytes);
    s.doff = body;
    body += pad8(s.dbytes);
    srcs.push_back(s);
    bufs.push_back({s.voff, s.vbytes});
    bufs.push_back({s.doff, s.dbytes});
    nodes.emplace_back(nrows, 0);
  }
  // body: fetch every buffer (device -> host unless the columns already live on the host), re-basing bit-packed buffers to bit 0
  std::vector<uint8_t> bodybuf((size_t)body, 0);
  std::vector<uint8_t> tmp;
  auto fetch = [&](void* dst, const void* src, size_t bytes) -> int {
    if (!bytes) return PDX_OK;
    if (columns_on_host) memcpy(dst, src, bytes);
    else PDX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
    return PDX_OK;
  };
  auto fetch_bits = [&](uint8_t* dst, const uint8_t* src, int64_t bit_off, int64_t nbits) -> int {
    if (!nbits) return PDX_OK;
    const int64_t b0 = bit_off >> 3, nb = ((bit_off + nbits + 7) >> 3) - b0;
    const int sh = (int)(bit_off & 7);
    if (sh == 0) return fetch(dst, src + b0, (size_t)((nbits + 7) / 8));
    tmp.assign((size_t)nb + 1, 0);
    PDX_TRY(fetch(tmp.data(), src + b0, (size_t)nb));
    if (!columns_on_host) PDX_HIP(hipStreamSynchronize(st));
    for (int64_t k = 0; k < (nbits + 7) / 8; ++k) dst[k] = (uint8_t)((tmp[(size_t)k] >> sh) | (tmp[(size_t)k + 1] << (8 - sh)));
    return PDX_OK;
  };
  for (int c = 0; c < ncols; ++c) {
    const Src& s = srcs[(size_t)c];
    if (s.validity) PDX_TRY(fetch_bits(bodybuf.data() + s.voff, s.validity, s.bit_off, nrows));
    if (cols[c].dtype == PDX_BOOL) PDX_TRY(fetch_bits(bodybuf.data() + s.doff, s.values, s.bit_off, nrows));
    else PDX_TRY(fetch(bodybuf.data() + s.doff, s.values + 8 * s.bit_off, (size_t)s.dbytes));
  }
  if (!columns_on_host) PDX_HIP(hipStreamSynchronize(st));
  for (int c = 0; c < ncols; ++c) {
    const Src& s = srcs[(size_t)c];
    if (!s.validity) continue;
    uint8_t* v = bodybuf.data() + s.voff;
    if (nrows & 7) v[(nrows - 1) / 8] &= (uint8_t)((1u << (nrows & 7)) - 1u);  // bits past the end are zero (deterministic output)
    int64_t set = 0;
    for (int64_t k = 0; k < s.vbytes; ++k) set += __builtin_popcount(v[k]);
    nodes[(size_t)c].second = nrows - set;
  }
  for (int c = 0; c < ncols; ++c)
    if (cols[c].dtype == PDX_BOOL && (nrows & 7)) bodybuf[(size_t)(srcs[(size_t)c].doff + (nrows - 1) / 8)] &= (uint8_t)((1u << (nrows & 7)) - 1u);
  std::vector<std::pair<std::string, std::string>> meta;
  for (int k = 0; k < nmeta; ++k) meta.emplace_back(metadata_kv[2 * k] ? metadata_kv[2 * k] : "", metadata_kv[2 * k + 1] ? metadata_kv[2 * k + 1] : "");
  std::vector<uint8_t> outv;
  append_framed(outv, schema_message(fields));
  append_framed(outv, record_batch_message(nrows, nodes, bufs, body, meta));
  outv.insert(outv.end(), bodybuf.begin(), bodybuf.end());
  const uint32_t eos[2] = {0xFFFFFFFFu, 0u};
  outv.insert(outv.end(), reinterpret_cast<const uint8_t*>(eos), reinterpret_cast<const uint8_t*>(eos) + 8);
  void* blob = malloc(outv.size() ? outv.size() : 1);
  if (!blob) return fail(PDX_OOM, "pdx_ipc_write: host allocation failed");
  memcpy(blob, outv.data(), outv.size());
  *out_blob = blob;
  *out_size = outv.size();
  return PDX_OK;
}

int pdx_ipc_free_blob(void* blob) {
  free(blob);
  return PDX_OK;
}

}  // extern "C"
