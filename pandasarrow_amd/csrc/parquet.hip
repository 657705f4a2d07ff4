// parquet.hip -- Parquet files -> device-resident columns (the ingest step in front of the hot path).
//
// Replaces, for the column types of this path, DataFrame::readParquet (reference src/dataframe.cpp:646-683: parquet::arrow::OpenFile
// -> FileReader::ReadTable -> TableBatchReader::ToRecordBatches, exactly ONE record batch or a std::runtime_error).  The reference
// decodes every page on the host into Arrow arrays; here the host only walks the metadata (the Thrift-compact footer and the page
// headers: a few hundred bytes per megabyte of data), the column chunks travel to the device in ONE copy, and every byte of page
// payload -- Snappy blocks, RLE / bit-packed definition levels, PLAIN and dictionary-encoded values -- is decoded by kernels
// straight into the Arrow layout the other entry points read (8-byte values + validity bitmap).  No Parquet / Arrow / Snappy library.
//
// Format (parquet-format 2.x): "PAR1" <row groups: column chunks: pages> <FileMetaData, Thrift compact> <u32 footer length> "PAR1".
// A page = PageHeader (Thrift compact) + payload.  DATA_PAGE (v1): payload = [u32 length + RLE/bit-packed hybrid definition levels,
// only for OPTIONAL columns][values], compressed as a whole.  DATA_PAGE_V2: [definition levels, not compressed, length in the
// header][values, compressed when is_compressed].  DICTIONARY_PAGE: PLAIN values; data pages then carry <u8 bit width> + hybrid runs
// of indices (RLE_DICTIONARY / PLAIN_DICTIONARY).
//
// Supported: flat schemas of BOOLEAN / INT32 / INT64 / FLOAT / DOUBLE leaves (signed / unsigned INTEGER annotations, TIMESTAMP in
// ms / us / ns -> timestamp[ns]), REQUIRED or OPTIONAL, one row group (the reference refuses more than one record batch), codecs
// UNCOMPRESSED and SNAPPY, encodings PLAIN / PLAIN_DICTIONARY / RLE_DICTIONARY (+ RLE levels), data pages v1 and v2.
// Everything else is refused by pdx_parquet_open with a message that names the column and the feature (strings, nested columns,
// INT96, DATE / TIME / DECIMAL annotations, GZIP / ZSTD / LZ4 / BROTLI, DELTA_* and BYTE_STREAM_SPLIT encodings, encryption).
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>
#include "pdx_common.hpp"

namespace pdx {
namespace {

// ---------------------------------------------------------------- Thrift compact protocol (bounds-checked reader)
struct Thrift {
  const uint8_t* p;
  size_t n;
  size_t at = 0;
  bool ok = true;
  uint8_t byte() {
    if (at >= n) {
      ok = false;
      return 0;
    }
    return p[at++];
  }
  uint64_t varint() {
    uint64_t v = 0;
    for (int sh = 0; sh < 70; sh += 7) {
      const uint8_t b = byte();
      v |= (uint64_t)(b & 0x7F) << (sh < 64 ? sh : 63);
      if (!(b & 0x80) || !ok) return v;
    }
    ok = false;
    return v;
  }
  int64_t zigzag() {
    const uint64_t v = varint();
    return (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
  }
  std::string binary() {
    const uint64_t len = varint();
    if (!ok || len > n - at) {
      ok = false;
      return std::string();
    }
    std::string s(reinterpret_cast<const char*>(p + at), (size_t)len);
    at += (size_t)len;
    return s;
  }
  // field header of a struct: returns the wire type (0 = STOP) and updates *id
  int field(int16_t* id) {
    const uint8_t b = byte();
    if (!ok || b == 0) return 0;
    const int delta = b >> 4, type = b & 15;
    if (delta) *id = (int16_t)(*id + delta);
    else *id = (int16_t)zigzag();
    return type;
  }
  void list(int* elem_type, uint64_t* size) {
    const uint8_t b = byte();
    *elem_type = b & 15;
    *size = b >> 4;
    if (*size == 15) *size = varint();
    if (*size > n) ok = false;  // (every element takes at least a byte... or none for bools packed in the header: still bounded)
  }
  void skip(int type, int depth = 0) {
    if (!ok || depth > 32) {
      ok = false;
      return;
    }
    switch (type) {
      case 1: case 2: break;  // bool carried by the field header
      case 3: byte(); break;
      case 4: case 5: case 6: varint(); break;
      case 7: at += 8; if (at > n) ok = false; break;
      case 8: binary(); break;
      case 9: case 10: {
        int et;
        uint64_t sz;
        list(&et, &sz);
        for (uint64_t i = 0; i < sz && ok; ++i) {
          if (et == 1 || et == 2) byte();  // bools inside a list take a byte each
          else skip(et, depth + 1);
        }
        break;
      }
      case 11: {
        const uint64_t sz = varint();
        if (sz) {
          const uint8_t kv = byte();
          for (uint64_t i = 0; i < sz && ok; ++i) {
            skip(kv >> 4, depth + 1);
            skip(kv & 15, depth + 1);
          }
        }
        break;
      }
      case 12: {
        int16_t id = 0;
        for (int t = field(&id); t && ok; t = field(&id)) skip(t, depth + 1);
        break;
      }
      default: ok = false;
    }
  }
};

enum { kPqBool = 0, kPqInt32 = 1, kPqInt64 = 2, kPqInt96 = 3, kPqFloat = 4, kPqDouble = 5, kPqByteArray = 6, kPqFixed = 7 };
enum { kEncPlain = 0, kEncPlainDict = 2, kEncRle = 3, kEncBitPacked = 4, kEncDeltaBinary = 5, kEncDeltaLen = 6, kEncDeltaBytes = 7, kEncRleDict = 8,
       kEncByteStreamSplit = 9 };
enum { kCodecNone = 0, kCodecSnappy = 1 };
const char* codec_name(int c) {
  static const char* names[] = {"UNCOMPRESSED", "SNAPPY", "GZIP", "LZO", "BROTLI", "LZ4", "ZSTD", "LZ4_RAW"};
  return c >= 0 && c < 8 ? names[c] : "unknown";
}
const char* encoding_name(int e) {
  static const char* names[] = {"PLAIN", "GROUP_VAR_INT", "PLAIN_DICTIONARY", "RLE", "BIT_PACKED", "DELTA_BINARY_PACKED", "DELTA_LENGTH_BYTE_ARRAY",
                                "DELTA_BYTE_ARRAY", "RLE_DICTIONARY", "BYTE_STREAM_SPLIT"};
  return e >= 0 && e < 10 ? names[e] : "unknown";
}

struct PqColumn {
  std::string name;
  int physical = -1;       // kPq*
  int repetition = 0;      // 0 REQUIRED, 1 OPTIONAL, 2 REPEATED
  int converted = -1;      // ConvertedType (legacy annotation)
  int ts_unit = -1;        // LogicalType TIMESTAMP unit: 0 ms, 1 us, 2 ns
  int int_bits = 0;        // LogicalType INTEGER
  bool int_signed = true;
  std::string logical_other;  // a logical type this path has no column for (STRING, DECIMAL, DATE, ...)
  int pdx_dtype = -1;
  int64_t mul = 1;         // timestamp -> nanoseconds
  // column chunk (row group 0)
  int codec = 0;
  int64_t num_values = 0, data_page_offset = 0, dict_page_offset = 0, total_compressed = 0;
  int64_t stat_null_count = -1;
  std::vector<int> encodings;
  // device side
  void* values = nullptr;    // 8-byte values (bit-packed for bool)
  void* validity = nullptr;  // bitmap or nullptr
  int64_t null_count = 0;
};

// one page of a column chunk, as the kernels see it
struct PqPage {
  int64_t src_off;     // payload offset in the uploaded file bytes
  int64_t raw_off;     // offset of the page's UNCOMPRESSED payload: in the raw buffer (in_raw = 1) or in the file bytes
  int64_t row0;        // first row of the page
  int32_t comp_size;   // payload bytes in the file
  int32_t raw_size;    // payload bytes once uncompressed
  int32_t num_values;  // rows of the page (flat columns: one value or null per row)
  int32_t def_len;     // v2: bytes of definition levels in front of the values; v1: -1 (u32 length prefix inside the payload)
  int32_t dict;        // 1: values are dictionary indices; 2: BOOLEAN values in the RLE encoding
  int32_t in_raw;
  int32_t nonnull;     // written by k_pq_levels (rows whose definition level is 1); = num_values for REQUIRED columns
  int32_t pad;
};
// a byte range to bring into the raw buffer: Snappy block or plain copy
struct PqSegment {
  int64_t src_off, dst_off;
  int32_t src_size, dst_size;
  int32_t snappy, pad;
};

constexpr int kPqErrSnappy = 1, kPqErrLevels = 2, kPqErrValues = 3, kPqErrDictIndex = 4;

// ---------------------------------------------------------------- device helpers: unaligned little-endian loads from byte streams
// (buffers carry 64 bytes of slack, so the second word of a straddling load is always addressable)
__device__ __forceinline__ uint64_t ld_u64(const uint8_t* p) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  const uint64_t* q = reinterpret_cast<const uint64_t*>(a & ~uintptr_t(7));
  const int sh = (int)(a & 7) * 8;
  const uint64_t lo = q[0];
  if (sh == 0) return lo;
  return (lo >> sh) | (q[1] << (64 - sh));
}
__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) { return (uint32_t)ld_u64(p); }
__device__ __forceinline__ uint32_t rd_varint(const uint8_t* p, int64_t* pos, int64_t end) {
  uint32_t v = 0;
  for (int sh = 0; sh < 35 && *pos < end; sh += 7) {
    const uint8_t b = p[(*pos)++];
    v |= (uint32_t)(b & 0x7F) << sh;
    if (!(b & 0x80)) break;
  }
  return v;
}

// ---------------------------------------------------------------- Snappy (raw block format), one wave per segment
// The element stream is sequential by nature; every lane follows the same tags (uniform loads) and the 64 lanes move the bytes of
// each literal / copy together.  A copy whose offset is smaller than its length repeats its source with period `offset`.
// Malformed input (lengths past either buffer, offsets before the start) sets the error word and ends the segment.
__global__ void __launch_bounds__(64) k_pq_unpack(const uint8_t* __restrict__ file, uint8_t* __restrict__ raw, const PqSegment* __restrict__ segs, int nsegs,
                                                   unsigned int* __restrict__ err) {
  const int lane = threadIdx.x;
  for (int si = blockIdx.x; si < nsegs; si += gridDim.x) {
    const PqSegment sg = segs[si];
    const uint8_t* src = file + sg.src_off;
    uint8_t* dst = raw + sg.dst_off;
    if (!sg.snappy) {
      for (int64_t i = lane; i < sg.src_size; i += 64) dst[i] = src[i];
      continue;
    }
    int64_t ip = 0, op = 0;
    const int64_t iend = sg.src_size, oend = sg.dst_size;
    const uint32_t ulen = rd_varint(src, &ip, iend);
    bool bad = (int64_t)ulen != oend;
    // (round 4: the stream position and everything decoded from the tags are wave-uniform; handing them over through readfirstlane keeps
    //  them in scalar registers, so the tag / length / offset bytes are fetched with scalar loads through the constant cache instead of a
    //  64-lane vector load of one byte each)
    ip = ((int64_t)__builtin_amdgcn_readfirstlane((int)(ip >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)ip);
    while (!bad && ip < iend) {
      const uint64_t w8 = ld_u64(src + ip);  // the tag and the <= 4 bytes behind it in one (scalar) read; 64 bytes of slack follow the file
      const uint64_t w = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(w8 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)w8);
      const uint32_t tag = (uint32_t)(w & 0xFF);
      ++ip;
      uint32_t len, off = 0;
      if ((tag & 3) == 0) {
        len = tag >> 2;
        if (len >= 60) {
          const int nb = (int)len - 59;
          if (ip + nb > iend) { bad = true; break; }
          len = (uint32_t)((w >> 8) & (nb == 4 ? 0xFFFFFFFFull : ((1ull << (8 * nb)) - 1ull)));
          ip += nb;
        }
        ++len;
        if (ip + (int64_t)len > iend || op + (int64_t)len > oend) { bad = true; break; }
        for (uint32_t i = lane; i < len; i += 64) dst[op + i] = src[ip + i];
        ip += len;
      } else {
        if ((tag & 3) == 1) {
          if (ip + 1 > iend) { bad = true; break; }
          len = 4 + ((tag >> 2) & 7);
          off = ((tag >> 5) << 8) | (uint32_t)((w >> 8) & 0xFF);
          ip += 1;
        } else if ((tag & 3) == 2) {
          if (ip + 2 > iend) { bad = true; break; }
          len = (tag >> 2) + 1;
          off = (uint32_t)((w >> 8) & 0xFFFF);
          ip += 2;
        } else {
          if (ip + 4 > iend) { bad = true; break; }
          len = (tag >> 2) + 1;
          off = (uint32_t)((w >> 8) & 0xFFFFFFFFull);
          ip += 4;
        }
        if (off == 0 || (int64_t)off > op || op + (int64_t)len > oend) { bad = true; break; }
        // the source bytes were written by this wave's earlier stores: make them visible before they are read back
        __threadfence_block();
        for (uint32_t i = lane; i < len; i += 64) dst[op + i] = dst[op - off + (i % off)];
      }
      op += len;
    }
    if (bad || op != oend) {
      if (lane == 0) atomicMax(err, (unsigned int)kPqErrSnappy);
    }
    __threadfence_block();
  }
}

// ---------------------------------------------------------------- Snappy, one WORKGROUP per segment (the product path)
// The wave-per-page kernel above walks the element stream one element at a time (a dependent tag read and, for copies, a read of bytes it
// has just stored: ~420 cycles per element, 46 ms for a 1 MB page of int64 values below 2^24 -- 262 K elements).  Here the sequential
// parts become logarithmic:
//   parse   a window of 4 KB of the stream lies in LDS; EVERY byte is decoded as if an element began there (next[i] = i + its size);
//           the true element starts are the nodes on the path from the window's first byte, marked by pointer jumping (<= 12 rounds
//           of mark[next^(2^k)(i)] |= mark[i], next^(2^(k+1)) = next^(2^k) o next^(2^k));
//   place   output position of every element = exclusive scan of the marked elements' lengths;
//   copy    the window's output is produced in tiles of 8 K bytes, one LDS word per byte: a literal byte is known at once, a copy byte
//           whose source lies in front of the tile is read back from the output (written and fenced by earlier tiles), a copy byte whose
//           source lies inside the tile starts as a pointer to it and is resolved by pointer doubling (tile[j] = tile[tile[j]]: the
//           pointers only point backwards, chains of overlapping copies halve every round).  Long literals (incompressible pages: one
//           64 KB literal per Snappy block) are moved global -> global in 16-byte pieces without the tile.
// Malformed input (lengths past either buffer, offsets before the start, a length prefix that does not match) sets the error word.
constexpr int kUsThreads = 1024, kUsWin = 4096, kUsTile = 8192, kUsMaxEl = kUsWin / 2 + 2, kUsPerThread = kUsWin / kUsThreads;
constexpr uint32_t kUsResolved = 0x80000000u;
constexpr int kUsTail = 256, kUsAhead = 64;
static_assert(kUsPerThread == 4 && kUsTile % kUsThreads == 0, "window / tile shares per thread");
// element that would begin at byte i of the window: header bytes, output length, copy offset (0: literal)
__device__ __forceinline__ void us_decode(const uint8_t* c, int i, uint32_t* hdr, uint64_t* len, uint32_t* off) {
  const uint32_t tag = c[i];
  const uint32_t ext = (uint32_t)c[i + 1] | ((uint32_t)c[i + 2] << 8) | ((uint32_t)c[i + 3] << 16) | ((uint32_t)c[i + 4] << 24);
  switch (tag & 3) {
    case 0: {
      uint32_t l = tag >> 2;
      *hdr = 1;
      if (l >= 60) {
        const int nb = (int)l - 59;
        l = nb == 4 ? ext : (ext & ((1u << (8 * nb)) - 1u));
        *hdr = 1 + nb;
      }
      *len = (uint64_t)l + 1;
      *off = 0;
      return;
    }
    case 1:
      *hdr = 2;
      *len = 4 + ((tag >> 2) & 7);
      *off = ((tag >> 5) << 8) | (ext & 0xFF);
      break;
    case 2:
      *hdr = 3;
      *len = (tag >> 2) + 1;
      *off = ext & 0xFFFF;
      break;
    default:
      *hdr = 5;
      *len = (tag >> 2) + 1;
      *off = ext;
      break;
  }
  if (*off == 0) *off = 0xFFFFFFFFu;  // a copy with offset 0 is malformed: an offset no output can satisfy (off == 0 means "literal" to the callers)
}
// scans over the workgroup's threads (16 waves)
// two exclusive sums in the same three barriers (element counts and output bytes)
__device__ __forceinline__ void us_excl_sum2(uint32_t a, uint32_t b, uint32_t* part /* >= 34 words */, uint32_t* ea, uint32_t* eb, uint32_t* ta, uint32_t* tb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t ia = a, ib = b;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t ua = __shfl_up(ia, d, 64), ub = __shfl_up(ib, d, 64);
    if (lane >= d) {
      ia += ua;
      ib += ub;
    }
  }
  if (lane == 63) {
    part[wave] = ia;
    part[17 + wave] = ib;
  }
  __syncthreads();
  if (wave == 0) {
    const uint32_t wa = lane < 16 ? part[lane] : 0, wb = lane < 16 ? part[17 + lane] : 0;
    uint32_t xa = wa, xb = wb;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const uint32_t ua = __shfl_up(xa, d, 64), ub = __shfl_up(xb, d, 64);
      if (lane >= d) {
        xa += ua;
        xb += ub;
      }
    }
    if (lane < 16) {
      part[lane] = xa - wa;
      part[17 + lane] = xb - wb;
    }
    if (lane == 15) {
      part[16] = xa;
      part[33] = xb;
    }
  }
  __syncthreads();
  *ea = part[wave] + ia - a;
  *eb = part[17 + wave] + ib - b;
  *ta = part[16];
  *tb = part[33];
  __syncthreads();
}
__device__ __forceinline__ uint32_t us_incl_max(uint32_t v, uint32_t* part) {  // returns the maximum over all threads IN FRONT of this one
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t u = __shfl_up(incl, d, 64);
    if (lane >= d) incl = incl > u ? incl : u;
  }
  if (lane == 63) part[wave] = incl;
  __syncthreads();
  uint32_t before = 0;
  for (int w = 0; w < wave; ++w) before = before > part[w] ? before : part[w];
  uint32_t prev = __shfl_up(incl, 1, 64);
  if (lane == 0) prev = 0;
  __syncthreads();
  return before > prev ? before : prev;
}
#ifdef PDX_US_TIMING  // per-phase cycle sums of every workgroup (diagnostic build: tools/build_variant.py), err[4 + phase] in units of 1024 cycles
#define US_T(k)                                     \
  do {                                              \
    if (threadIdx.x == 0) {                         \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
      acc_[k] += now_ - last_;                      \
      last_ = now_;                                 \
    }                                               \
  } while (0)
#else
#define US_T(k) do { } while (0)
#endif
__global__ void __launch_bounds__(kUsThreads) k_pq_unsnap(const uint8_t* __restrict__ file, uint8_t* __restrict__ raw, const PqSegment* __restrict__ segs,
                                                          int nsegs, unsigned int* __restrict__ err) {
  __shared__ __attribute__((aligned(16))) uint8_t cbuf[kUsWin + 16];
  __shared__ __attribute__((aligned(16))) uint16_t ja[kUsWin + 8], jb[kUsWin + 8];
  __shared__ __attribute__((aligned(16))) uint8_t mark[kUsWin + 8];
  __shared__ uint8_t tailb[kUsTail];  // the last bytes produced before the current tile (copies with short offsets read them here)
  __shared__ uint16_t el_cpos[kUsMaxEl];
  __shared__ uint32_t el_opos[kUsMaxEl];
  __shared__ __attribute__((aligned(16))) uint32_t tile[kUsTile];
  __shared__ uint32_t part[36];
  __shared__ int s_bad;
  __shared__ long long s_next;
  const int tid = threadIdx.x;
#ifdef PDX_US_TIMING
  unsigned long long acc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#endif
  for (int si = blockIdx.x; si < nsegs; si += gridDim.x) {
    const PqSegment sg = segs[si];
    const uint8_t* src = file + sg.src_off;
    uint8_t* dst = raw + sg.dst_off;
    if (!sg.snappy) {
      for (int64_t i = tid; i < sg.src_size; i += kUsThreads) dst[i] = src[i];
      continue;
    }
    __syncthreads();  // (the previous segment's last reads of the shared words)
    if (tid == 0) s_bad = 0;
    int64_t ip = 0, op = 0;
    const int64_t iend = sg.src_size, oend = sg.dst_size;
    const uint32_t ulen = rd_varint(src, &ip, iend);
    bool bad = (int64_t)ulen != oend;
    constexpr int kPfLen = kUsWin + 16 + 2 * kUsAhead, kPf = (kPfLen + kUsThreads - 1) / kUsThreads;
    uint8_t pf[kPf];
    int64_t pf_at = -1;
#pragma unroll
    for (int q = 0; q < kPf; ++q) pf[q] = 0;
    __syncthreads();
    while (!bad && ip < iend) {
      // ---- the window [ip, ip + 4 KB) of the stream (+ the <= 4 header bytes of an element that begins on its last byte)
      const int64_t s = ip;
      const int wlen = (int)(iend - s < kUsWin ? iend - s : kUsWin);
      // the bytes requested while the previous window was at work (pf_at: where they begin) cover this window unless a long literal
      // jumped ahead; whatever they do not cover comes straight from the stream
      {
        const int64_t shift = pf_at - s;  // cbuf index of the first prefetched byte
        const bool usable = pf_at >= 0 && shift <= 0 && shift > -2 * kUsAhead;
#pragma unroll
        for (int q = 0; q < kPf; ++q) {
          const int64_t k = (int64_t)tid + (int64_t)q * kUsThreads;  // byte k of the prefetched range
          const int64_t i = k + shift;
          if (usable && k < kPfLen && i >= 0 && i < kUsWin + 16) cbuf[i] = pf[q];
        }
        const int covered = usable ? (int)(kPfLen + shift) : 0;  // cbuf[0, covered) is filled
        for (int i = (covered > 0 ? covered : 0) + tid; i < kUsWin + 16; i += kUsThreads) cbuf[i] = s + i < iend ? src[s + i] : (uint8_t)0;
      }
      __syncthreads();
      pf_at = s + kUsWin - kUsAhead;  // the next window begins within a few bytes of s + 4 KB (an element straddles the end) unless a literal runs on
#pragma unroll
      for (int q = 0; q < kPf; ++q) {
        const int64_t a = pf_at + tid + (int64_t)q * kUsThreads;
        pf[q] = (tid + q * kUsThreads < kPfLen && a < iend) ? src[a] : (uint8_t)0;
      }
      US_T(0);
      // ---- parse: next[] of every byte, then the path from byte 0
#pragma unroll
      for (int q = 0; q < kUsPerThread; ++q) {
        const int i = tid * kUsPerThread + q;
        uint32_t nx = kUsWin;
        if (i < wlen) {
          uint32_t hdr, off;
          uint64_t len;
          us_decode(cbuf, i, &hdr, &len, &off);
          const uint64_t e = (uint64_t)i + hdr + (off ? 0 : len);
          nx = e < (uint64_t)kUsWin ? (uint32_t)e : (uint32_t)kUsWin;
        }
        ja[i] = (uint16_t)nx;
        mark[i] = i == 0 ? 1 : 0;
      }
      if (tid == 0) ja[kUsWin] = jb[kUsWin] = kUsWin;
      __syncthreads();
      US_T(1);
      {
        // round k: j0 = next^(2^k).  The path from byte 0 has left the window within 2^k elements when j0[0] is the exit: the rounds so
        // far marked its first 2^k nodes, i.e. all of them (a window of one long literal needs no round at all).
        uint16_t *j0 = ja, *j1 = jb;
        for (int round = 0; round < 12; ++round) {
          if (j0[0] == kUsWin) break;
          const ushort4 jv = *reinterpret_cast<const ushort4*>(j0 + kUsPerThread * tid);
          const uint32_t mv = *reinterpret_cast<const uint32_t*>(mark + kUsPerThread * tid);
          ushort4 nv;
          nv.x = j0[jv.x];  // (j0[exit] = exit)
          nv.y = j0[jv.y];
          nv.z = j0[jv.z];
          nv.w = j0[jv.w];
          if ((mv & 0xFFu) && jv.x < kUsWin) mark[jv.x] = 1;
          if ((mv & 0xFF00u) && jv.y < kUsWin) mark[jv.y] = 1;
          if ((mv & 0xFF0000u) && jv.z < kUsWin) mark[jv.z] = 1;
          if ((mv & 0xFF000000u) && jv.w < kUsWin) mark[jv.w] = 1;
          *reinterpret_cast<ushort4*>(j1 + kUsPerThread * tid) = nv;
          __syncthreads();
          uint16_t* t = j0;
          j0 = j1;
          j1 = t;
        }
      }
      US_T(2);
      // ---- place: the marked bytes are the elements, in order; their output positions by a scan of the lengths
      uint32_t cnt[kUsPerThread], ol[kUsPerThread], my_cnt = 0, my_len = 0;
      int my_bad = 0;
      int64_t my_next = -1;
#pragma unroll
      for (int q = 0; q < kUsPerThread; ++q) {
        const int i = tid * kUsPerThread + q;
        cnt[q] = 0;
        ol[q] = 0;
        if (i < wlen && mark[i]) {
          uint32_t hdr, off;
          uint64_t len;
          us_decode(cbuf, i, &hdr, &len, &off);
          cnt[q] = 1;
          if (s + i + hdr + (int64_t)(off ? 0 : len) > iend || (int64_t)len > oend) {
            my_bad = 1;
            len = 0;
          }
          ol[q] = (uint32_t)len;
          my_next = s + i + hdr + (int64_t)(off ? 0 : len);  // (the thread's LAST element wins below)
        }
        my_cnt += cnt[q];
        my_len += ol[q];
      }
      uint32_t nel, total, e0, o0;
      us_excl_sum2(my_cnt, my_len, part, &e0, &o0, &nel, &total);
#pragma unroll
      for (int q = 0; q < kUsPerThread; ++q) {
        const int i = tid * kUsPerThread + q;
        if (cnt[q]) {
          uint32_t hdr, off;
          uint64_t len;
          us_decode(cbuf, i, &hdr, &len, &off);
          // a copy must begin inside what has been produced so far, and nothing may be produced behind the announced length
          if (off && (off > (uint64_t)op + o0)) my_bad = 1;
          if (op + (int64_t)o0 + (int64_t)ol[q] > oend) my_bad = 1;
          el_cpos[e0] = (uint16_t)i;
          el_opos[e0] = o0;
          if (e0 + 1 == nel) s_next = my_next;
          ++e0;
          o0 += ol[q];
        }
      }
      if (my_bad) s_bad = 1;
      if (tid == 0) el_opos[nel] = total;
      __syncthreads();
      if (s_bad || op + (int64_t)total > oend) {
        bad = true;
        break;
      }
      US_T(3);
      // ---- copy: the window's output [op, op + total) in tiles
      uint32_t t0 = 0;
      while (t0 < total) {
        // the element that covers output byte t0 (the same search in every thread)
        uint32_t lo = 0, hi = nel - 1;
        while (lo < hi) {
          const uint32_t mid = (lo + hi + 1) >> 1;
          if (el_opos[mid] <= t0) lo = mid;
          else hi = mid - 1;
        }
        const uint32_t efirst = lo;
        {
          uint32_t hdr, off;
          uint64_t len;
          us_decode(cbuf, el_cpos[efirst], &hdr, &len, &off);
          const uint32_t rem = el_opos[efirst] + (uint32_t)len - t0;
          if (!off && rem >= 2048) {  // a long literal: straight from the stream to the output, 16 bytes per thread and step
            const uint8_t* from = src + s + el_cpos[efirst] + hdr + (t0 - el_opos[efirst]);
            uint8_t* to = dst + op + t0;
            const uint32_t head = (uint32_t)((16 - (reinterpret_cast<uintptr_t>(to) & 15)) & 15);
            if ((uint32_t)tid < head) to[tid] = from[tid];
            const uint32_t body = (rem - head) >> 4;
            for (uint32_t k = tid; k < body; k += kUsThreads) {
              const uint8_t* f = from + head + ((size_t)k << 4);
              ulonglong2 v;
              v.x = ld_u64(f);
              v.y = ld_u64(f + 8);
              *reinterpret_cast<ulonglong2*>(to + head + ((size_t)k << 4)) = v;
            }
            for (uint32_t k = head + (body << 4) + tid; k < rem; k += kUsThreads) to[k] = from[k];
            if (tid < kUsTail) tailb[tid] = from[rem - kUsTail + tid];  // (rem >= 2048)
            t0 += rem;
            __threadfence_block();
            __syncthreads();
            US_T(8);
            continue;
          }
        }
        const uint32_t t1 = total - t0 < (uint32_t)kUsTile ? total : t0 + kUsTile, tl = t1 - t0;
        constexpr int kOwn = kUsTile / kUsThreads;  // 16 consecutive bytes per thread for the element lookup
        {
          uint4* tz = reinterpret_cast<uint4*>(tile + tid * kOwn);
#pragma unroll
          for (int q = 0; q < kOwn / 4; ++q) tz[q] = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        // heads: element e begins at tile byte el_opos[e] - t0 (value e + 1); the element that reaches in from the left sits on byte 0
        for (uint32_t e = efirst + tid; e < nel && el_opos[e] < t1; e += kUsThreads) tile[el_opos[e] > t0 ? el_opos[e] - t0 : 0] = e + 1;
        __syncthreads();
        uint32_t own[kOwn];
        {
          const uint4* tz = reinterpret_cast<const uint4*>(tile + tid * kOwn);
#pragma unroll
          for (int q = 0; q < kOwn / 4; ++q) {
            const uint4 v = tz[q];
            own[4 * q] = v.x;
            own[4 * q + 1] = v.y;
            own[4 * q + 2] = v.z;
            own[4 * q + 3] = v.w;
          }
        }
        uint32_t run = 0;
#pragma unroll
        for (int q = 0; q < kOwn; ++q) {
          run = own[q] > run ? own[q] : run;
          own[q] = run;
        }
        const uint32_t before = us_incl_max(run, part);
        US_T(4);
        // one word per byte: its value (resolved) or the tile byte it repeats
#pragma unroll
        for (int q = 0; q < kOwn; ++q) {
          const uint32_t jj = tid * kOwn + q;
          uint32_t word = kUsResolved;
          if (jj < tl) {
            const uint32_t e = (own[q] > before ? own[q] : before) - 1;
            const uint32_t cp = el_cpos[e];
            uint32_t hdr, off;
            uint64_t len;
            us_decode(cbuf, cp, &hdr, &len, &off);
            const uint32_t j = t0 + jj;  // output byte, relative to the window's first
            if (!off) {
              const uint32_t lp = cp + hdr + (j - el_opos[e]);  // literal byte: in the window's LDS copy or further along the stream
              word = kUsResolved | (lp < (uint32_t)kUsWin + 16 ? (uint32_t)cbuf[lp] : (uint32_t)src[s + lp]);
            } else if (off > jj) {  // in front of the tile: among the last bytes kept in LDS, or read back (produced and fenced earlier)
              word = kUsResolved | (off - jj <= (uint32_t)kUsTail ? (uint32_t)tailb[kUsTail + jj - off] : (uint32_t)dst[op + j - off]);
            } else {
              word = jj - off;
            }
          }
          own[q] = word;
        }
        {
          uint4* tz = reinterpret_cast<uint4*>(tile + tid * kOwn);
#pragma unroll
          for (int q = 0; q < kOwn / 4; ++q) tz[q] = make_uint4(own[4 * q], own[4 * q + 1], own[4 * q + 2], own[4 * q + 3]);
        }
        __syncthreads();
        US_T(5);
        for (int round = 0; round < 16; ++round) {  // pointer doubling: chains of <= 8 K bytes need <= 13 rounds
          int open = 0;
          for (uint32_t jj = tid; jj < tl; jj += kUsThreads) {
            const uint32_t w = tile[jj];
            if (!(w & kUsResolved)) {
              const uint32_t w2 = tile[w];
              tile[jj] = w2;
              open |= !(w2 & kUsResolved);
            }
          }
          if (!__syncthreads_or(open)) break;
        }
        US_T(6);
        for (uint32_t jj = tid; jj < tl; jj += kUsThreads) dst[op + t0 + jj] = (uint8_t)tile[jj];
        if (tl >= (uint32_t)kUsTail) {  // the tile's last bytes for the next tile's short-offset copies (tailb is not read between the
          if (tid < kUsTail) tailb[tid] = (uint8_t)tile[tl - kUsTail + tid];  // words above and the barrier below)
        } else {
          uint8_t keep = 0;
          if (tid < kUsTail && tid + tl < (uint32_t)kUsTail) keep = tailb[tid + tl];
          __syncthreads();
          if (tid < kUsTail) tailb[tid] = tid + tl < (uint32_t)kUsTail ? keep : (uint8_t)tile[tid + tl - kUsTail];
        }
        t0 = t1;
        __threadfence_block();
        __syncthreads();
        US_T(7);
      }
      op += total;
      ip = s_next;
      __syncthreads();
    }
    if (bad || op != oend || ip != iend) {
      if (tid == 0) atomicMax(err, (unsigned int)kPqErrSnappy);
    }
  }
#ifdef PDX_US_TIMING
  if (tid == 0)
    for (int k = 0; k < 10; ++k) atomicAdd(&err[4 + k], (unsigned int)(acc_[k] >> 10));
#endif
}

// ---------------------------------------------------------------- RLE / bit-packed hybrid runs (definition levels, dictionary indices)
// Calls emit(k, v) for the first `need` values of the stream in [pos, end); every lane walks the run headers, the values of a run are
// spread over the lanes.  Returns false when the stream ends early or a run header is malformed.
template <typename Emit>
__device__ __forceinline__ bool hybrid_decode(const uint8_t* p, int64_t pos, int64_t end, int bw, int64_t need, int lane, Emit emit) {
  int64_t done = 0;
  const int vbytes = (bw + 7) >> 3;
  while (done < need) {
    if (pos >= end) return false;
    const uint32_t h = rd_varint(p, &pos, end);
    if (h & 1) {  // bit-packed run: groups of 8 values, bw bits each, LSB first
      const int64_t cnt = (int64_t)(h >> 1) * 8, bytes = (int64_t)(h >> 1) * bw;
      if (cnt == 0 || pos + bytes > end + 8) return false;  // (writers may cut the padding of the last group short)
      const int64_t take = cnt < need - done ? cnt : need - done;
      for (int64_t j = lane; j < take; j += 64) {
        const int64_t bit = j * bw;
        const uint64_t w = ld_u64(p + pos + (bit >> 3));
        emit(done + j, (uint32_t)((w >> (bit & 7)) & ((bw == 32) ? 0xFFFFFFFFull : ((1ull << bw) - 1ull))));
      }
      pos += bytes;
      done += take;
    } else {  // RLE run: one value, repeated
      const int64_t cnt = h >> 1;
      if (cnt == 0 || pos + vbytes > end) return false;
      uint32_t v = 0;
      for (int k = 0; k < vbytes; ++k) v |= (uint32_t)p[pos + k] << (8 * k);
      pos += vbytes;
      const int64_t take = cnt < need - done ? cnt : need - done;
      for (int64_t j = lane; j < take; j += 64) emit(done + j, v);
      done += take;
    }
  }
  return true;
}

__device__ __forceinline__ const uint8_t* page_payload(const PqPage& pg, const uint8_t* file, const uint8_t* raw) {
  return (pg.in_raw ? raw : file) + pg.raw_off;
}
// where the definition levels and the values of a page lie inside its uncompressed payload
__device__ __forceinline__ void page_split(const PqPage& pg, const uint8_t* pay, int optional, int64_t* lev0, int64_t* lev1, int64_t* val0) {
  if (!optional) {
    *lev0 = *lev1 = 0;
    *val0 = pg.def_len > 0 ? pg.def_len : 0;
  } else if (pg.def_len >= 0) {  // v2: length from the header
    *lev0 = 0;
    *lev1 = pg.def_len;
    *val0 = pg.def_len;
  } else {  // v1: u32 length prefix
    const int64_t L = pg.raw_size >= 4 ? (int64_t)ld_u32(pay) : (int64_t)0x7FFFFFFF;
    *lev0 = 4;
    *lev1 = 4 + L;
    *val0 = 4 + L;
  }
}

// definition levels of an OPTIONAL flat column (bit width 1): one byte per row (1 = valid) + the page's count of valid rows
__global__ void __launch_bounds__(64) k_pq_levels(const uint8_t* __restrict__ file, const uint8_t* __restrict__ raw, PqPage* __restrict__ pages, int npages,
                                                   uint8_t* __restrict__ valid_bytes, unsigned int* __restrict__ err) {
  const int lane = threadIdx.x;
  for (int pi = blockIdx.x; pi < npages; pi += gridDim.x) {
    const PqPage pg = pages[pi];
    const uint8_t* pay = page_payload(pg, file, raw);
    int64_t lev0, lev1, val0;
    page_split(pg, pay, 1, &lev0, &lev1, &val0);
    int cnt = 0;
    bool good = lev1 <= pg.raw_size;
    if (good) {
      uint8_t* out = valid_bytes + pg.row0;
      good = hybrid_decode(pay, lev0, lev1, 1, pg.num_values, lane, [&](int64_t k, uint32_t v) {
        out[k] = (uint8_t)(v & 1);
        cnt += (int)(v & 1);
      });
    }
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
    if (lane == 0) {
      pages[pi].nonnull = good ? cnt : 0;
      if (!good) atomicMax(err, (unsigned int)kPqErrLevels);
    }
  }
}

// physical value j of a PLAIN stream -> the path's 8-byte representation
template <int PHYS>
__device__ __forceinline__ uint64_t plain_value(const uint8_t* vals, int64_t j, int is_signed, long long mul) {
  if (PHYS == kPqInt64) return (uint64_t)((long long)ld_u64(vals + j * 8) * mul);
  if (PHYS == kPqDouble) return ld_u64(vals + j * 8);
  if (PHYS == kPqInt32) {
    const uint32_t w = ld_u32(vals + j * 4);
    return (uint64_t)((is_signed ? (long long)(int32_t)w : (long long)w) * mul);
  }
  if (PHYS == kPqFloat) return (uint64_t)__double_as_longlong((double)__uint_as_float(ld_u32(vals + j * 4)));
  return (uint64_t)((vals[j >> 3] >> (j & 7)) & 1);  // BOOLEAN: bit-packed, LSB first
}
template <int PHYS>
__device__ __forceinline__ int64_t plain_bytes(int64_t count) {
  return PHYS == kPqBool ? (count + 7) / 8 : count * ((PHYS == kPqInt64 || PHYS == kPqDouble) ? 8 : 4);
}

// dictionary page -> 8-byte values (PLAIN)
template <int PHYS>
__global__ void k_pq_dictionary(const uint8_t* __restrict__ file, const uint8_t* __restrict__ raw, const PqPage* __restrict__ page, int is_signed, long long mul,
                                uint64_t* __restrict__ dict, unsigned int* __restrict__ err) {
  const PqPage pg = *page;
  const uint8_t* pay = page_payload(pg, file, raw);
  if (plain_bytes<PHYS>(pg.num_values) > pg.raw_size) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(err, (unsigned int)kPqErrValues);
    return;
  }
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < pg.num_values; j += (int64_t)gridDim.x * blockDim.x)
    dict[j] = plain_value<PHYS>(pay, j, is_signed, mul);
}

// values of the data pages, densely: value j of a page (its j-th NON-NULL row) goes to slot row0 + j of `dense`.  A page without nulls
// therefore lands on its final rows already; pages with nulls are spread out by k_pq_expand.
template <int PHYS>
__global__ void __launch_bounds__(64) k_pq_values(const uint8_t* __restrict__ file, const uint8_t* __restrict__ raw, const PqPage* __restrict__ pages, int npages,
                                                   int optional, int is_signed, long long mul, const uint64_t* __restrict__ dict, int64_t dict_size,
                                                   uint64_t* __restrict__ dense, unsigned int* __restrict__ err) {
  const int lane = threadIdx.x;
  for (int pi = blockIdx.x; pi < npages; pi += gridDim.x) {
    const PqPage pg = pages[pi];
    const uint8_t* pay = page_payload(pg, file, raw);
    int64_t lev0, lev1, val0;
    page_split(pg, pay, optional, &lev0, &lev1, &val0);
    const int64_t cnt = optional ? pg.nonnull : pg.num_values;
    uint64_t* out = dense + pg.row0;
    bool good = val0 <= pg.raw_size;
    if (good && cnt > 0) {
      if (pg.dict == 2) {
        good = val0 + 4 <= pg.raw_size;
        if (good) {
          const int64_t L = (int64_t)ld_u32(pay + val0);
          good = val0 + 4 + L <= pg.raw_size && hybrid_decode(pay, val0 + 4, val0 + 4 + L, 1, cnt, lane, [&](int64_t k, uint32_t v) { out[k] = v & 1; });
        }
      } else if (pg.dict) {
        const int bw = pay[val0];
        good = bw <= 32 && val0 + 1 <= pg.raw_size;
        if (good && bw == 0) {  // a one-entry dictionary: every index is 0
          good = dict_size > 0;
          if (good)
            for (int64_t j = lane; j < cnt; j += 64) out[j] = dict[0];
        } else if (good) {
          bool idx_ok = true;
          good = hybrid_decode(pay, val0 + 1, pg.raw_size, bw, cnt, lane, [&](int64_t k, uint32_t v) {
            if ((int64_t)v < dict_size) out[k] = dict[v];
            else idx_ok = false;
          });
          if (!idx_ok) atomicMax(err, (unsigned int)kPqErrDictIndex);
        }
      } else {
        good = val0 + plain_bytes<PHYS>(cnt) <= pg.raw_size;
        if (good)
          for (int64_t j = lane; j < cnt; j += 64) out[j] = plain_value<PHYS>(pay + val0, j, is_signed, mul);
      }
    }
    if (!good && lane == 0) atomicMax(err, (unsigned int)kPqErrValues);
  }
}

// pages with nulls: row i of the page takes dense value (number of valid rows in front of it); walked back to front in 64-row steps
// so that the expansion can run IN PLACE (a row's source slot never lies behind its own position)
__global__ void __launch_bounds__(64) k_pq_expand(const PqPage* __restrict__ pages, int npages, const uint8_t* __restrict__ valid_bytes, uint64_t* __restrict__ vals) {
  const int lane = threadIdx.x;
  for (int pi = blockIdx.x; pi < npages; pi += gridDim.x) {
    const PqPage pg = pages[pi];
    if (pg.nonnull == pg.num_values) continue;  // no nulls: the dense values are in place
    const uint8_t* vb = valid_bytes + pg.row0;
    uint64_t* v = vals + pg.row0;
    int64_t remaining = pg.nonnull;  // valid rows in front of the step being written, counted from the page's end backwards
    const int64_t nsteps = ((int64_t)pg.num_values + 63) >> 6;
    for (int64_t s = nsteps - 1; s >= 0; --s) {
      const int64_t i = s * 64 + lane;
      const bool ok = i < pg.num_values && vb[i] != 0;
      const unsigned long long m = __ballot(ok);
      const int here = __popcll(m);
      const int64_t base = remaining - here;  // valid rows in front of this step
      const int64_t src = base + __popcll(m & ((1ull << lane) - 1ull));
      const uint64_t x = ok ? v[src] : 0ull;  // every source of this step lies at or in front of the step's first row ...
      __threadfence_block();                   // ... or inside it: all reads of the step come before its writes
      if (i < pg.num_values) v[i] = x;
      __threadfence_block();
      remaining = base;
    }
  }
}

__global__ void k_pq_pack_bits(const uint8_t* __restrict__ bytes, int64_t n, uint8_t* __restrict__ bits) {
  const int64_t nb = (n + 7) / 8, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += stride) {
    uint8_t w = 0;
    for (int k = 0; k < 8; ++k) {
      const int64_t i = b * 8 + k;
      if (i < n && bytes[i]) w |= (uint8_t)(1u << k);
    }
    bits[b] = w;
  }
}
__global__ void k_pq_pack_bool_values(const uint64_t* __restrict__ vals, int64_t n, uint8_t* __restrict__ bits) {
  const int64_t nb = (n + 7) / 8, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += stride) {
    uint8_t w = 0;
    for (int k = 0; k < 8; ++k) {
      const int64_t i = b * 8 + k;
      if (i < n && (vals[i] & 1)) w |= (uint8_t)(1u << k);
    }
    bits[b] = w;
  }
}

}  // namespace
}  // namespace pdx

using namespace pdx;

struct pdx_parquet_file {
  std::vector<PqColumn> cols;
  int64_t num_rows = 0;
  std::vector<std::pair<std::string, std::string>> metadata;  // FileMetaData.key_value_metadata
  std::string created_by;
  const uint8_t* blob = nullptr;  // borrowed until pdx_parquet_load returns
  size_t size = 0;
  bool loaded = false;
  hipStream_t stream = nullptr;
  std::vector<void*> owned;
  ~pdx_parquet_file() {
    if (owned.empty()) return;
    StreamNote note(stream);
    pool_free_many(owned.data(), (int)owned.size());
  }
};

namespace pdx {
namespace {

int parse_logical_type(Thrift& t, PqColumn* c) {
  static const char* names[] = {"", "STRING", "MAP", "LIST", "ENUM", "DECIMAL", "DATE", "TIME", "TIMESTAMP", "", "INTEGER", "UNKNOWN", "JSON", "BSON", "UUID", "FLOAT16"};
  int16_t id = 0;
  for (int ty = t.field(&id); ty && t.ok; ty = t.field(&id)) {
    if (id == 8 && ty == 12) {  // TimestampType {1: isAdjustedToUTC, 2: unit {1 MILLIS, 2 MICROS, 3 NANOS}}
      int16_t f = 0;
      for (int t2 = t.field(&f); t2 && t.ok; t2 = t.field(&f)) {
        if (f == 2 && t2 == 12) {
          int16_t u = 0;
          for (int t3 = t.field(&u); t3 && t.ok; t3 = t.field(&u)) {
            if (u >= 1 && u <= 3) c->ts_unit = u - 1;
            t.skip(t3);
          }
        } else {
          t.skip(t2);
        }
      }
    } else if (id == 10 && ty == 12) {  // IntType {1: bitWidth i8, 2: isSigned bool}
      int16_t f = 0;
      for (int t2 = t.field(&f); t2 && t.ok; t2 = t.field(&f)) {
        if (f == 1 && t2 == 3) c->int_bits = (int8_t)t.byte();
        else if (f == 2 && (t2 == 1 || t2 == 2)) c->int_signed = t2 == 1;
        else t.skip(t2);
      }
    } else {
      if (id != 11) c->logical_other = id > 0 && id < 16 ? names[id] : "unknown";  // (UNKNOWN = an all-null column: no constraint)
      t.skip(ty);
    }
  }
  return PDX_OK;
}

struct SchemaElem {
  PqColumn col;
  int num_children = -1;
};

int parse_schema_element(Thrift& t, SchemaElem* e) {
  int16_t id = 0;
  for (int ty = t.field(&id); ty && t.ok; ty = t.field(&id)) {
    switch (id) {
      case 1: e->col.physical = (int)t.zigzag(); break;
      case 3: e->col.repetition = (int)t.zigzag(); break;
      case 4: e->col.name = t.binary(); break;
      case 5: e->num_children = (int)t.zigzag(); break;
      case 6: e->col.converted = (int)t.zigzag(); break;
      case 10:
        if (ty == 12) PDX_TRY(parse_logical_type(t, &e->col));
        else t.skip(ty);
        break;
      default: t.skip(ty);
    }
  }
  return PDX_OK;
}

int parse_column_meta(Thrift& t, PqColumn* c, std::vector<std::string>* path) {
  int16_t id = 0;
  for (int ty = t.field(&id); ty && t.ok; ty = t.field(&id)) {
    switch (id) {
      case 2: {
        int et;
        uint64_t sz;
        t.list(&et, &sz);
        for (uint64_t i = 0; i < sz && t.ok; ++i) c->encodings.push_back((int)t.zigzag());
        break;
      }
      case 3: {
        int et;
        uint64_t sz;
        t.list(&et, &sz);
        for (uint64_t i = 0; i < sz && t.ok; ++i) path->push_back(t.binary());
        break;
      }
      case 4: c->codec = (int)t.zigzag(); break;
      case 5: c->num_values = t.zigzag(); break;
      case 7: c->total_compressed = t.zigzag(); break;
      case 9: c->data_page_offset = t.zigzag(); break;
      case 11: c->dict_page_offset = t.zigzag(); break;
      case 12: {  // Statistics: only null_count (3)
        int16_t f = 0;
        for (int t2 = t.field(&f); t2 && t.ok; t2 = t.field(&f)) {
          if (f == 3 && t2 == 6) c->stat_null_count = t.zigzag();
          else t.skip(t2);
        }
        break;
      }
      default: t.skip(ty);
    }
  }
  return PDX_OK;
}

// what a leaf becomes on the device, or a refusal that names the column
int resolve_type(PqColumn* c) {
  const std::string who = "pdx_parquet_open: column '" + c->name + "'";
  if (c->repetition == 2) return fail(PDX_NOT_IMPLEMENTED, who + " is REPEATED (nested / list columns are not on this path)");
  if (!c->logical_other.empty())
    return fail(PDX_NOT_IMPLEMENTED, who + " has logical type " + c->logical_other + " (only plain numbers, booleans and timestamps are on this path)");
  // ConvertedType (files written without LogicalType): 9 TIMESTAMP_MILLIS, 10 TIMESTAMP_MICROS, 11..14 UINT_8..64, 15..18 INT_8..64
  if (c->converted >= 0 && c->ts_unit < 0 && c->int_bits == 0) {
    if (c->converted == 9) c->ts_unit = 0;
    else if (c->converted == 10) c->ts_unit = 1;
    else if (c->converted >= 11 && c->converted <= 14) { c->int_bits = 8 << (c->converted - 11); c->int_signed = false; }
    else if (c->converted >= 15 && c->converted <= 18) { c->int_bits = 8 << (c->converted - 15); c->int_signed = true; }
    else return fail(PDX_NOT_IMPLEMENTED, who + " has converted type " + std::to_string(c->converted) + " (strings, decimals, dates and times are not on this path)");
  }
  switch (c->physical) {
    case kPqBool: c->pdx_dtype = PDX_BOOL; break;
    case kPqInt32:
      if (c->ts_unit >= 0) return fail(PDX_INVALID, who + ": TIMESTAMP on INT32");
      c->pdx_dtype = PDX_INT64;
      break;
    case kPqInt64:
      if (c->ts_unit >= 0) {
        static const int64_t to_ns[3] = {1000000, 1000, 1};
        c->pdx_dtype = PDX_TIMESTAMP_NS;
        c->mul = to_ns[c->ts_unit];
      } else {
        c->pdx_dtype = (c->int_bits == 64 && !c->int_signed) ? PDX_UINT64 : PDX_INT64;
      }
      break;
    case kPqFloat: case kPqDouble: c->pdx_dtype = PDX_FLOAT64; break;
    case kPqInt96: return fail(PDX_NOT_IMPLEMENTED, who + " is INT96 (legacy timestamps are not supported: write int64 timestamps)");
    case kPqByteArray: case kPqFixed: return fail(PDX_NOT_IMPLEMENTED, who + " is a BYTE_ARRAY / string column (only numeric, boolean and timestamp columns are on this path)");
    default: return fail(PDX_INVALID, who + ": unknown physical type");
  }
  return PDX_OK;
}

int parse_footer(pdx_parquet_file* f) {
  const uint8_t* b = f->blob;
  const size_t n = f->size;
  if (n < 12 || memcmp(b, "PAR1", 4) != 0) return fail(PDX_INVALID, "pdx_parquet_open: not a Parquet file (no PAR1 magic)");
  if (memcmp(b + n - 4, "PARE", 4) == 0) return fail(PDX_NOT_IMPLEMENTED, "pdx_parquet_open: encrypted footers are not supported");
  if (memcmp(b + n - 4, "PAR1", 4) != 0) return fail(PDX_INVALID, "pdx_parquet_open: truncated file (no PAR1 magic at the end)");
  uint32_t flen;
  memcpy(&flen, b + n - 8, 4);
  if ((size_t)flen > n - 12) return fail(PDX_INVALID, "pdx_parquet_open: footer length exceeds the file");
  Thrift t{b + n - 8 - flen, flen};
  std::vector<SchemaElem> schema;
  struct Chunk {
    PqColumn meta;
    std::vector<std::string> path;
    bool has_meta = false;
    std::string file_path;
  };
  std::vector<std::vector<Chunk>> row_groups;
  std::vector<int64_t> rg_rows;
  int16_t id = 0;
  for (int ty = t.field(&id); ty && t.ok; ty = t.field(&id)) {
    if (id == 2 && ty == 9) {
      int et;
      uint64_t sz;
      t.list(&et, &sz);
      for (uint64_t i = 0; i < sz && t.ok; ++i) {
        SchemaElem e;
        PDX_TRY(parse_schema_element(t, &e));
        schema.push_back(std::move(e));
      }
    } else if (id == 3 && ty == 6) {
      f->num_rows = t.zigzag();
    } else if (id == 4 && ty == 9) {
      int et;
      uint64_t sz;
      t.list(&et, &sz);
      for (uint64_t r = 0; r < sz && t.ok; ++r) {
        std::vector<Chunk> chunks;
        int64_t rows = 0;
        int16_t rid = 0;
        for (int rt = t.field(&rid); rt && t.ok; rt = t.field(&rid)) {
          if (rid == 1 && rt == 9) {
            int cet;
            uint64_t csz;
            t.list(&cet, &csz);
            for (uint64_t c = 0; c < csz && t.ok; ++c) {
              Chunk ch;
              int16_t cid = 0;
              for (int ct = t.field(&cid); ct && t.ok; ct = t.field(&cid)) {
                if (cid == 1 && ct == 8) ch.file_path = t.binary();
                else if (cid == 3 && ct == 12) {
                  PDX_TRY(parse_column_meta(t, &ch.meta, &ch.path));
                  ch.has_meta = true;
                } else if (cid == 8 || cid == 9) {  // crypto_metadata / encrypted_column_metadata
                  return fail(PDX_NOT_IMPLEMENTED, "pdx_parquet_open: encrypted columns are not supported");
                } else {
                  t.skip(ct);
                }
              }
              chunks.push_back(std::move(ch));
            }
          } else if (rid == 3 && rt == 6) {
            rows = t.zigzag();
          } else {
            t.skip(rt);
          }
        }
        row_groups.push_back(std::move(chunks));
        rg_rows.push_back(rows);
      }
    } else if (id == 5 && ty == 9) {
      int et;
      uint64_t sz;
      t.list(&et, &sz);
      for (uint64_t i = 0; i < sz && t.ok; ++i) {
        std::string k, v;
        int16_t kid = 0;
        for (int kt = t.field(&kid); kt && t.ok; kt = t.field(&kid)) {
          if (kid == 1 && kt == 8) k = t.binary();
          else if (kid == 2 && kt == 8) v = t.binary();
          else t.skip(kt);
        }
        f->metadata.emplace_back(std::move(k), std::move(v));
      }
    } else if (id == 6 && ty == 8) {
      f->created_by = t.binary();
    } else if (id == 8 || id == 9) {  // encryption_algorithm / footer_signing_key_metadata
      return fail(PDX_NOT_IMPLEMENTED, "pdx_parquet_open: encrypted files are not supported");
    } else {
      t.skip(ty);
    }
  }
  if (!t.ok || schema.empty() || f->num_rows < 0) return fail(PDX_INVALID, "pdx_parquet_open: malformed footer (FileMetaData)");
  // flat schema: the root, then leaves only
  const int nleaves = (int)schema.size() - 1;
  if (schema[0].num_children != nleaves)
    return fail(PDX_NOT_IMPLEMENTED, "pdx_parquet_open: nested schema (groups / lists / maps are not on this path: flat columns only)");
  for (int i = 1; i <= nleaves; ++i) {
    if (schema[(size_t)i].num_children > 0)
      return fail(PDX_NOT_IMPLEMENTED, "pdx_parquet_open: column '" + schema[(size_t)i].col.name + "' is a group (nested columns are not on this path)");
    PDX_TRY(resolve_type(&schema[(size_t)i].col));
  }
  // the reference turns the table into record batches and accepts exactly one (src/dataframe.cpp:662-676): one chunk per column
  if (row_groups.size() > 1 && f->num_rows > 0)
    return fail(PDX_INVALID, "DataFrame Only supports Parquet Table with single record batch\nFound " + std::to_string(row_groups.size()) + " record batches\n");
  // an empty table gives the reference zero record batches: it throws as well (src/dataframe.cpp:666-667)
  if (row_groups.empty() || f->num_rows == 0) return fail(PDX_INVALID, "Cannot Initialize DataFrame with empty parquet table");
  const std::vector<Chunk>& chunks = row_groups[0];
  if ((int)chunks.size() != nleaves || rg_rows[0] != f->num_rows) return fail(PDX_INVALID, "pdx_parquet_open: row group does not match the schema");
  for (int i = 0; i < nleaves; ++i) {
    PqColumn c = schema[(size_t)i + 1].col;
    const Chunk& ch = chunks[(size_t)i];
    const std::string who = "pdx_parquet_open: column '" + c.name + "'";
    if (!ch.has_meta || !ch.file_path.empty()) return fail(PDX_NOT_IMPLEMENTED, who + ": column chunk metadata is not in this file");
    if (ch.path.size() != 1 || ch.path[0] != c.name) return fail(PDX_INVALID, who + ": column chunk path does not match the schema");
    const PqColumn& m = ch.meta;
    if (m.codec != kCodecNone && m.codec != kCodecSnappy)
      return fail(PDX_NOT_IMPLEMENTED, who + " is compressed with " + codec_name(m.codec) + " (only UNCOMPRESSED and SNAPPY pages are decoded on the device)");
    for (int e : m.encodings)
      if (e != kEncPlain && e != kEncPlainDict && e != kEncRle && e != kEncRleDict && e != kEncBitPacked)
        return fail(PDX_NOT_IMPLEMENTED, who + " uses encoding " + encoding_name(e) + " (only PLAIN and dictionary encodings are decoded on the device)");
    c.codec = m.codec;
    c.num_values = m.num_values;
    c.data_page_offset = m.data_page_offset;
    c.dict_page_offset = m.dict_page_offset;
    c.total_compressed = m.total_compressed;
    c.stat_null_count = m.stat_null_count;
    c.encodings = m.encodings;
    if (c.num_values != f->num_rows) return fail(PDX_INVALID, who + ": value count differs from the row count (flat columns hold one value or null per row)");
    const int64_t start = (c.dict_page_offset > 0 && c.dict_page_offset < c.data_page_offset) ? c.dict_page_offset : c.data_page_offset;
    if (start < 4 || c.total_compressed < 0 || (uint64_t)start > n || (uint64_t)c.total_compressed > n - (uint64_t)start)
      return fail(PDX_INVALID, who + ": column chunk lies outside the file");
    f->cols.push_back(std::move(c));
  }
  return PDX_OK;
}

struct PageHeader {
  int type = -1;  // 0 data v1, 2 dictionary, 3 data v2
  int32_t uncompressed = 0, compressed = 0;
  int32_t num_values = 0, encoding = 0, def_len = 0, rep_len = 0, num_nulls = -1;
  int def_encoding = kEncRle;
  bool v2_compressed = true;
  size_t header_bytes = 0;
};
int parse_page_header(const uint8_t* p, size_t n, PageHeader* h) {
  Thrift t{p, n};
  int16_t id = 0;
  for (int ty = t.field(&id); ty && t.ok; ty = t.field(&id)) {
    if (id == 1 && ty == 5) h->type = (int)t.zigzag();
    else if (id == 2 && ty == 5) h->uncompressed = (int32_t)t.zigzag();
    else if (id == 3 && ty == 5) h->compressed = (int32_t)t.zigzag();
    else if ((id == 5 || id == 7 || id == 8) && ty == 12) {
      int16_t f = 0;
      for (int t2 = t.field(&f); t2 && t.ok; t2 = t.field(&f)) {
        if (id == 5) {  // DataPageHeader
          if (f == 1 && t2 == 5) h->num_values = (int32_t)t.zigzag();
          else if (f == 2 && t2 == 5) h->encoding = (int)t.zigzag();
          else if (f == 3 && t2 == 5) h->def_encoding = (int)t.zigzag();
          else t.skip(t2);
        } else if (id == 7) {  // DictionaryPageHeader
          if (f == 1 && t2 == 5) h->num_values = (int32_t)t.zigzag();
          else if (f == 2 && t2 == 5) h->encoding = (int)t.zigzag();
          else t.skip(t2);
        } else {  // DataPageHeaderV2
          if (f == 1 && t2 == 5) h->num_values = (int32_t)t.zigzag();
          else if (f == 2 && t2 == 5) h->num_nulls = (int32_t)t.zigzag();
          else if (f == 4 && t2 == 5) h->encoding = (int)t.zigzag();
          else if (f == 5 && t2 == 5) h->def_len = (int32_t)t.zigzag();
          else if (f == 6 && t2 == 5) h->rep_len = (int32_t)t.zigzag();
          else if (f == 7 && (t2 == 1 || t2 == 2)) h->v2_compressed = t2 == 1;
          else t.skip(t2);
        }
      }
    } else {
      t.skip(ty);
    }
  }
  if (!t.ok) return fail(PDX_INVALID, "pdx_parquet_load: malformed page header");
  h->header_bytes = t.at;
  return PDX_OK;
}

template <typename F>
int dispatch_physical(int phys, F&& f) {
  switch (phys) {
    case kPqBool: return f(std::integral_constant<int, kPqBool>());
    case kPqInt32: return f(std::integral_constant<int, kPqInt32>());
    case kPqInt64: return f(std::integral_constant<int, kPqInt64>());
    case kPqFloat: return f(std::integral_constant<int, kPqFloat>());
    default: return f(std::integral_constant<int, kPqDouble>());
  }
}

}  // namespace
}  // namespace pdx

extern "C" {

int pdx_parquet_open(const void* blob, size_t size, pdx_parquet_file** out) {
  if (!out) return fail(PDX_INVALID, "pdx_parquet_open: null output");
  *out = nullptr;
  if (!blob) return fail(PDX_INVALID, "pdx_parquet_open: null buffer");
  std::unique_ptr<pdx_parquet_file> f(new pdx_parquet_file());
  f->blob = static_cast<const uint8_t*>(blob);
  f->size = size;
  PDX_TRY(parse_footer(f.get()));
  *out = f.release();
  return PDX_OK;
}
int pdx_parquet_destroy(pdx_parquet_file* f) {
  delete f;
  return PDX_OK;
}
int pdx_parquet_num_columns(const pdx_parquet_file* f) { return f ? (int)f->cols.size() : -1; }
int64_t pdx_parquet_num_rows(const pdx_parquet_file* f) { return f ? f->num_rows : -1; }
const char* pdx_parquet_column_name(const pdx_parquet_file* f, int i) {
  return (f && i >= 0 && i < (int)f->cols.size()) ? f->cols[(size_t)i].name.c_str() : nullptr;
}
int pdx_parquet_num_metadata(const pdx_parquet_file* f) { return f ? (int)f->metadata.size() : -1; }
const char* pdx_parquet_metadata_key(const pdx_parquet_file* f, int i) {
  return (f && i >= 0 && i < (int)f->metadata.size()) ? f->metadata[(size_t)i].first.c_str() : nullptr;
}
const char* pdx_parquet_metadata_value(const pdx_parquet_file* f, int i) {
  return (f && i >= 0 && i < (int)f->metadata.size()) ? f->metadata[(size_t)i].second.c_str() : nullptr;
}

int pdx_parquet_load(pdx_parquet_file* f, void* stream) {
  if (!f) return fail(PDX_INVALID, "pdx_parquet_load: null file");
  if (f->loaded) return PDX_OK;
  if (!f->blob) return fail(PDX_INVALID, "pdx_parquet_load: the file bytes were released by an earlier, failed load");
  hipStream_t st = as_stream(stream);
  f->stream = st;
  const int64_t n = f->num_rows;
  auto undo = [f](int rc) {  // a failed load leaves nothing behind
    {
      StreamNote note(f->stream);
      pool_free_many(f->owned.data(), (int)f->owned.size());
    }
    f->owned.clear();
    for (auto& c : f->cols) c.values = c.validity = nullptr;
    return rc;
  };
  auto own = [f](size_t bytes) -> void* {
    void* p = pool_alloc(bytes ? bytes : 1);
    if (p) f->owned.push_back(p);
    return p;
  };
  // ---- host: walk the page headers of every column chunk (a header is ~20-60 bytes in front of ~1 MB of payload)
  struct ColPlan {
    std::vector<PqPage> pages;  // data pages
    PqPage dict{};
    bool has_dict = false;
    int64_t dict_values = 0;
  };
  std::vector<ColPlan> plans(f->cols.size());
  std::vector<PqSegment> segs;
  int64_t raw_total = 0;
  int64_t lo = (int64_t)f->size, hi = 0;
  for (size_t ci = 0; ci < f->cols.size(); ++ci) {
    PqColumn& c = f->cols[ci];
    ColPlan& pl = plans[ci];
    const std::string who = "pdx_parquet_load: column '" + c.name + "'";
    int64_t at = (c.dict_page_offset > 0 && c.dict_page_offset < c.data_page_offset) ? c.dict_page_offset : c.data_page_offset;
    const int64_t end = at + c.total_compressed;
    lo = std::min(lo, at);
    hi = std::max(hi, end);
    int64_t rows = 0;
    while (at < end && rows < n) {
      PageHeader h;
      PDX_TRY(parse_page_header(f->blob + at, (size_t)(end - at), &h));
      const int64_t pay = at + (int64_t)h.header_bytes;
      if (h.compressed < 0 || h.uncompressed < 0 || pay + h.compressed > end) return fail(PDX_INVALID, who + ": page lies outside its column chunk");
      at = pay + h.compressed;
      if (h.type == 1) continue;  // INDEX_PAGE: nothing to decode
      if (h.type != 0 && h.type != 2 && h.type != 3) return fail(PDX_NOT_IMPLEMENTED, who + ": unknown page type " + std::to_string(h.type));
      if (h.num_values < 0) return fail(PDX_INVALID, who + ": negative value count in a page header");
      PqPage pg{};
      pg.src_off = pay;
      pg.comp_size = h.compressed;
      pg.raw_size = h.uncompressed;
      pg.num_values = h.num_values;
      pg.def_len = -1;
      pg.row0 = rows;
      pg.nonnull = h.num_values;
      if (h.type == 2) {
        if (h.encoding != kEncPlain && h.encoding != kEncPlainDict) return fail(PDX_NOT_IMPLEMENTED, who + ": dictionary page encoding " + encoding_name(h.encoding));
        if (pl.has_dict || !pl.pages.empty()) return fail(PDX_INVALID, who + ": more than one dictionary page");
      } else {
        if (h.encoding == kEncPlainDict || h.encoding == kEncRleDict) pg.dict = 1;
        else if (h.encoding == kEncRle && c.physical == kPqBool) pg.dict = 2;  // booleans as <u32 length> + hybrid runs of bit width 1 (v2 pages)
        else if (h.encoding != kEncPlain)
          return fail(PDX_NOT_IMPLEMENTED, who + " uses encoding " + encoding_name(h.encoding) + " (only PLAIN and dictionary encodings are decoded on the device)");
        if (pg.dict == 1 && !pl.has_dict) return fail(PDX_INVALID, who + ": dictionary-encoded page without a dictionary page");
        if (h.type == 0 && c.repetition == 1 && h.def_encoding != kEncRle)
          return fail(PDX_NOT_IMPLEMENTED, who + ": definition levels in the legacy BIT_PACKED encoding");
        if (rows + h.num_values > n) return fail(PDX_INVALID, who + ": pages hold more values than the file has rows");
      }
      const bool compressed = c.codec == kCodecSnappy && (h.type != 3 || h.v2_compressed);
      if (h.type == 3) {
        if (h.rep_len != 0) return fail(PDX_NOT_IMPLEMENTED, who + ": repetition levels (nested data)");
        if (h.def_len < 0 || h.def_len > h.compressed || h.def_len > h.uncompressed) return fail(PDX_INVALID, who + ": bad level length in a v2 page header");
        pg.def_len = h.def_len;
      }
      if (compressed) {
        pg.in_raw = 1;
        pg.raw_off = raw_total;
        if (h.type == 3) {  // levels are stored as they are, only the values are compressed
          if (h.def_len) segs.push_back(PqSegment{pay, raw_total, h.def_len, h.def_len, 0, 0});
          segs.push_back(PqSegment{pay + h.def_len, raw_total + h.def_len, h.compressed - h.def_len, h.uncompressed - h.def_len, 1, 0});
        } else {
          segs.push_back(PqSegment{pay, raw_total, h.compressed, h.uncompressed, 1, 0});
        }
        raw_total += ((int64_t)h.uncompressed + 15) & ~int64_t(15);
      } else {
        if (h.compressed != h.uncompressed) return fail(PDX_INVALID, who + ": uncompressed page whose sizes differ");
        pg.in_raw = 0;
        pg.raw_off = pay;  // (rebased to the uploaded range below)
      }
      if (h.type == 2) {
        pl.dict = pg;
        pl.has_dict = true;
        pl.dict_values = h.num_values;
      } else {
        pl.pages.push_back(pg);
        rows += h.num_values;
      }
    }
    if (rows != n) return fail(PDX_INVALID, who + ": pages hold " + std::to_string(rows) + " values, the file has " + std::to_string(n) + " rows");
  }
  if (hi <= lo) return fail(PDX_INVALID, "pdx_parquet_load: no column data");
  // ---- ONE host->device copy: the byte range that holds every column chunk
  const int64_t span = hi - lo;
  uint8_t* dfile = static_cast<uint8_t*>(own((size_t)span + 64));
  uint8_t* draw = static_cast<uint8_t*>(own((size_t)raw_total + 64));
  unsigned int* derr = static_cast<unsigned int*>(own(64));
  if (!dfile || !draw || !derr) return undo(PDX_OOM);
#define PQ_HIP(expr)                                              \
  do {                                                            \
    const hipError_t _e = (expr);                                 \
    if (_e != hipSuccess) return undo(hip_fail(_e, #expr));       \
  } while (0)
  PQ_HIP(hipMemsetAsync(derr, 0, 64, st));
  PQ_HIP(hipMemsetAsync(dfile + span, 0, 64, st));
  PQ_HIP(hipMemsetAsync(draw + raw_total, 0, 64, st));
  for (auto& s : segs) s.src_off -= lo;
  std::vector<PqPage> all_pages;  // dictionary page (if any) first, then the data pages, per column
  std::vector<size_t> first_page(f->cols.size());
  for (size_t ci = 0; ci < f->cols.size(); ++ci) {
    first_page[ci] = all_pages.size();
    ColPlan& pl = plans[ci];
    if (pl.has_dict) all_pages.push_back(pl.dict);
    for (auto& pg : pl.pages) all_pages.push_back(pg);
  }
  for (auto& pg : all_pages) {
    pg.src_off -= lo;
    if (!pg.in_raw) pg.raw_off -= lo;
  }
  PqPage* dpages = static_cast<PqPage*>(own(all_pages.size() * sizeof(PqPage)));
  PqSegment* dsegs = static_cast<PqSegment*>(own(segs.size() * sizeof(PqSegment)));
  if (!dpages || !dsegs) return undo(PDX_OOM);
  if (!all_pages.empty()) PQ_HIP(hipMemcpyAsync(dpages, all_pages.data(), all_pages.size() * sizeof(PqPage), hipMemcpyHostToDevice, st));
  // ---- the file bytes go up in pieces and the pages of a piece are decompressed while the next piece is on its way: the upload of a
  // pageable host buffer (~57 GB/s) and the Snappy kernel (~65 GB/s of page bytes) take about the same time, one after the other they
  // were 4.5 + 4.0 of the 9 ms of a 259 MB file.  Pieces are copied on the caller's stream, the kernels run on a side stream behind an
  // event per piece; the caller's stream waits for the side stream before anything reads the pages.
  // (PDX_PQ_SNAPPY_WAVE=1: the wave-per-page decoder, kept as the cross-check of the workgroup-parallel one: one copy, one launch)
  const char* wenv = getenv("PDX_PQ_SNAPPY_WAVE");
  const bool wave_form = wenv && wenv[0] == '1';
  // piece size: a fifth of the file, at least 32 MB (every piece's pages are one launch: 8 MB pieces = 8 pages per launch leave the chip
  // idle, 11.7 ms; 16 / 32 / 48-64 MB: 8.3 / 7.8 / 7.2 ms for the 259 MB file; one copy + one launch: 9.1 ms)
  const int64_t piece = [span] {
    const char* e = getenv("PDX_PQ_UPLOAD_PIECE_MB");
    return e ? (int64_t)std::max(atoll(e), 1ll) << 20 : std::max<int64_t>((int64_t)32 << 20, span / 5);
  }();
  bool overlapped = false;
  if (!segs.empty()) {
    // decode order = file order (a page is ready when the piece that holds its last byte has arrived)
    std::stable_sort(segs.begin(), segs.end(), [](const PqSegment& a, const PqSegment& b) { return a.src_off + a.src_size < b.src_off + b.src_size; });
    PQ_HIP(hipMemcpyAsync(dsegs, segs.data(), segs.size() * sizeof(PqSegment), hipMemcpyHostToDevice, st));
  }
  if (!segs.empty() && !wave_form && span >= 2 * piece) {
    hipStream_t side = nullptr;
    hipEvent_t ev = nullptr;
    if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
      overlapped = true;
      hipError_t e = hipSuccess;
      size_t done = 0;  // segments launched so far
      for (int64_t at = 0; at < span && e == hipSuccess; at += piece) {
        const int64_t len = std::min(piece, span - at);
        e = hipMemcpyAsync(dfile + at, f->blob + lo + at, (size_t)len, hipMemcpyHostToDevice, st);
        size_t upto = done;
        while (upto < segs.size() && segs[upto].src_off + segs[upto].src_size <= at + len) ++upto;
        if (e == hipSuccess && upto > done) {
          e = hipEventRecord(ev, st);
          if (e == hipSuccess) e = hipStreamWaitEvent(side, ev, 0);
          if (e == hipSuccess) {
            hipLaunchKernelGGL(k_pq_unsnap, dim3((unsigned)std::min<size_t>(upto - done, (size_t)kCUs * 8)), dim3(kUsThreads), 0, side, dfile, draw,
                               dsegs + done, (int)(upto - done), derr);
            e = hipGetLastError();
          }
          done = upto;
        }
      }
      if (e == hipSuccess) e = hipEventRecord(ev, side);
      if (e == hipSuccess) e = hipStreamWaitEvent(st, ev, 0);
      if (e != hipSuccess) (void)hipStreamSynchronize(side);  // (nothing of this load may still be running when its blocks go back)
      // the side stream's work is ordered in front of everything queued on `st` from here on; destroying a stream lets its work finish
      (void)hipEventDestroy(ev);
      (void)hipStreamDestroy(side);
      if (e != hipSuccess) return undo(hip_fail(e, "pdx_parquet_load: overlapped upload"));
      if (done != segs.size()) return undo(fail(PDX_INVALID, "pdx_parquet_load: a page lies outside the uploaded range"));
    } else {
      (void)hipGetLastError();
      if (ev) (void)hipEventDestroy(ev);
      if (side) (void)hipStreamDestroy(side);
    }
  }
  if (!overlapped) {
    PQ_HIP(hipMemcpyAsync(dfile, f->blob + lo, (size_t)span, hipMemcpyHostToDevice, st));
    if (!segs.empty()) {
      if (wave_form)
        hipLaunchKernelGGL(k_pq_unpack, dim3((unsigned)std::min<size_t>(segs.size(), (size_t)kCUs * 16)), dim3(64), 0, st, dfile, draw, dsegs, (int)segs.size(), derr);
      else
        hipLaunchKernelGGL(k_pq_unsnap, dim3((unsigned)std::min<size_t>(segs.size(), (size_t)kCUs * 8)), dim3(kUsThreads), 0, st, dfile, draw, dsegs,
                           (int)segs.size(), derr);
    }
  }
  // ---- per column: levels -> valid bytes, dictionary, values (dense per page), expansion of the pages with nulls, bitmaps
  std::vector<uint8_t*> valid_bytes(f->cols.size(), nullptr);
  for (size_t ci = 0; ci < f->cols.size(); ++ci) {
    PqColumn& c = f->cols[ci];
    ColPlan& pl = plans[ci];
    const int np = (int)pl.pages.size();
    PqPage* dp = dpages + first_page[ci] + (pl.has_dict ? 1 : 0);
    const int optional = c.repetition == 1;
    const unsigned grid = (unsigned)std::min<int>(std::max(np, 1), kCUs * 16);
    uint64_t* vals = static_cast<uint64_t*>(own((size_t)(n > 0 ? n : 1) * 8 + 64));
    if (!vals) return undo(PDX_OOM);
    if (optional) {
      valid_bytes[ci] = static_cast<uint8_t*>(own((size_t)n + 64));
      if (!valid_bytes[ci]) return undo(PDX_OOM);
      if (np) hipLaunchKernelGGL(k_pq_levels, dim3(grid), dim3(64), 0, st, dfile, draw, dp, np, valid_bytes[ci], derr);
    }
    uint64_t* dict = nullptr;
    if (pl.has_dict) {
      dict = static_cast<uint64_t*>(own((size_t)(pl.dict_values > 0 ? pl.dict_values : 1) * 8));
      if (!dict) return undo(PDX_OOM);
    }
    const int rc = dispatch_physical(c.physical, [&](auto phys) -> int {
      constexpr int P = decltype(phys)::value;
      if (pl.has_dict && pl.dict_values > 0)
        hipLaunchKernelGGL((k_pq_dictionary<P>), dim3(grid_for(pl.dict_values, 256)), dim3(256), 0, st, dfile, draw, dpages + first_page[ci],
                           c.int_signed ? 1 : 0, (long long)c.mul, dict, derr);
      if (np)
        hipLaunchKernelGGL((k_pq_values<P>), dim3(grid), dim3(64), 0, st, dfile, draw, dp, np, optional, c.int_signed ? 1 : 0, (long long)c.mul, dict,
                           pl.dict_values, vals, derr);
      return PDX_OK;
    });
    if (rc != PDX_OK) return undo(rc);
    if (optional && np) hipLaunchKernelGGL(k_pq_expand, dim3(grid), dim3(64), 0, st, dp, np, valid_bytes[ci], vals);
    PQ_HIP(hipGetLastError());
    if (c.pdx_dtype == PDX_BOOL) {
      uint8_t* bits = static_cast<uint8_t*>(own((size_t)((n + 7) / 8) + 64));
      if (!bits) return undo(PDX_OOM);
      hipLaunchKernelGGL(k_pq_pack_bool_values, dim3(grid_for((n + 7) / 8, 256)), dim3(256), 0, st, vals, n, bits);
      c.values = bits;
    } else {
      c.values = vals;
    }
    if (optional) {
      uint8_t* bits = static_cast<uint8_t*>(own((size_t)((n + 7) / 8) + 64));
      if (!bits) return undo(PDX_OOM);
      hipLaunchKernelGGL(k_pq_pack_bits, dim3(grid_for((n + 7) / 8, 256)), dim3(256), 0, st, valid_bytes[ci], n, bits);
      c.validity = bits;
    }
    PQ_HIP(hipGetLastError());
  }
  // ---- results of the device-side checks + the null counts
  unsigned int herr = 0;
#ifdef PDX_US_TIMING
  {
    unsigned int t[16];
    PQ_HIP(hipMemcpyAsync(t, derr, sizeof(t), hipMemcpyDeviceToHost, st));
    PQ_HIP(hipStreamSynchronize(st));
    fprintf(stderr, "[pdx] k_pq_unsnap Kcycles: load %u decode %u mark %u place %u heads %u words %u resolve %u (sum over workgroups) writeout %u literal %u\n", t[4], t[5], t[6], t[7],
            t[8], t[9], t[10], t[11], t[12]);
  }
#endif
  PQ_HIP(hipMemcpyAsync(&herr, derr, sizeof(herr), hipMemcpyDeviceToHost, st));
  if (!all_pages.empty()) PQ_HIP(hipMemcpyAsync(all_pages.data(), dpages, all_pages.size() * sizeof(PqPage), hipMemcpyDeviceToHost, st));
  PQ_HIP(hipStreamSynchronize(st));
#undef PQ_HIP
  if (herr) {
    static const char* what[] = {"", "a Snappy block is malformed", "definition levels are malformed", "a values section is shorter than its page header says",
                                 "a dictionary index lies outside the dictionary"};
    return undo(fail(PDX_INVALID, std::string("pdx_parquet_load: ") + what[herr < 5 ? herr : 3]));
  }
  for (size_t ci = 0; ci < f->cols.size(); ++ci) {
    PqColumn& c = f->cols[ci];
    int64_t nulls = 0;
    const size_t p0 = first_page[ci] + (plans[ci].has_dict ? 1 : 0);
    for (size_t k = 0; k < plans[ci].pages.size(); ++k) nulls += all_pages[p0 + k].num_values - all_pages[p0 + k].nonnull;
    c.null_count = c.repetition == 1 ? nulls : 0;
  }
  // the staging blocks -- the uploaded file range, the decompressed pages, one byte per row of validity, dictionaries, page tables, the
  // 8-byte form of boolean values -- go back to the pool now: the handle lives as long as the frame that aliases its columns, and must
  // not pin file size + uncompressed size + a byte per nullable row of HBM beside them.  Only the columns' own buffers stay.
  {
    std::vector<void*> keep, drop;
    for (void* q : f->owned) {
      bool k = false;
      for (const PqColumn& c : f->cols) k = k || q == c.values || q == c.validity;
      (k ? keep : drop).push_back(q);
    }
    StreamNote note(f->stream);
    pool_free_many(drop.data(), (int)drop.size());
    f->owned.swap(keep);
  }
  f->blob = nullptr;  // the caller may free the file bytes once this returns
  f->loaded = true;
  return PDX_OK;
}

int pdx_parquet_column(const pdx_parquet_file* f, int i, pdx_column* out) {
  if (!f || !out || i < 0 || i >= (int)f->cols.size()) return fail(PDX_INVALID, "pdx_parquet_column: bad argument");
  const PqColumn& c = f->cols[(size_t)i];
  memset(out, 0, sizeof(*out));
  out->dtype = c.pdx_dtype;
  out->length = f->num_rows;
  out->offset = 0;
  out->null_count = f->loaded ? c.null_count : (c.repetition == 1 ? c.stat_null_count : 0);  // before the load: the chunk's statistics, -1 if absent
  if (f->loaded) {
    out->validity = c.null_count > 0 ? c.validity : nullptr;
    out->values = c.values;
  }
  return PDX_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- writing (DataFrame::toParquet, reference src/dataframe.cpp:685-724)
// The reference hands one record batch to parquet::arrow::WriteTable with default properties.  This writer emits the same logical
// content in the simplest valid physical form: ONE row group, per column a chain of uncompressed v1 data pages with PLAIN values; an
// OPTIONAL column's definition levels are ONE bit-packed run per page -- whose payload is, bit for bit, the Arrow validity bitmap of
// the page's rows (LSB first, 1 = valid = level 1).  Assembly happens on the host (buffers are fetched once, like pdx_ipc_write).
namespace pdx {
namespace {
struct ThriftOut {
  std::vector<uint8_t> b;
  std::vector<int16_t> last{0};
  void varint(uint64_t v) {
    while (v >= 0x80) {
      b.push_back((uint8_t)(v | 0x80));
      v >>= 7;
    }
    b.push_back((uint8_t)v);
  }
  void zigzag(int64_t v) { varint(((uint64_t)v << 1) ^ (uint64_t)(v >> 63)); }
  void field(int16_t id, int type) {
    const int delta = id - last.back();
    if (delta > 0 && delta <= 15) b.push_back((uint8_t)((delta << 4) | type));
    else {
      b.push_back((uint8_t)type);
      zigzag(id);
    }
    last.back() = id;
  }
  void i32(int16_t id, int32_t v) { field(id, 5); zigzag(v); }
  void i64(int16_t id, int64_t v) { field(id, 6); zigzag(v); }
  void i8(int16_t id, int8_t v) { field(id, 3); b.push_back((uint8_t)v); }
  void boolean(int16_t id, bool v) { field(id, v ? 1 : 2); }
  void str(int16_t id, const std::string& v) { field(id, 8); varint(v.size()); b.insert(b.end(), v.begin(), v.end()); }
  void list(int16_t id, int elem_type, size_t n) {
    field(id, 9);
    if (n < 15) b.push_back((uint8_t)((n << 4) | elem_type));
    else {
      b.push_back((uint8_t)(0xF0 | elem_type));
      varint(n);
    }
  }
  void begin_field_struct(int16_t id) { field(id, 12); last.push_back(0); }
  void begin_elem_struct() { last.push_back(0); }
  void end_struct() { b.push_back(0); last.pop_back(); }
};
}  // namespace
}  // namespace pdx

extern "C" {

int pdx_parquet_write(const pdx_column* cols, const char* const* names, int ncols, int columns_on_host, void* stream, void** out_blob, size_t* out_size) {
  if ((ncols > 0 && (!cols || !names)) || ncols <= 0 || !out_blob || !out_size) return fail(PDX_INVALID, "pdx_parquet_write: bad argument");
  hipStream_t st = as_stream(stream);
  const int64_t n = cols[0].length;
  constexpr int64_t kPageRows = 1 << 20;  // rows per data page (a multiple of 8: pages start on a byte of the validity bitmap)
  std::vector<uint8_t> file{'P', 'A', 'R', '1'};
  struct ChunkInfo {
    int64_t first_page = 0, bytes = 0, nulls = 0;
  };
  std::vector<ChunkInfo> chunks((size_t)ncols);
  auto fetch = [&](void* dst, const void* src, size_t bytes) -> int {
    if (!bytes) return PDX_OK;
    if (columns_on_host) memcpy(dst, src, bytes);
    else {
      PDX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
      PDX_HIP(hipStreamSynchronize(st));
    }
    return PDX_OK;
  };
  // bits [bit_off, bit_off + nbits) of a bitmap, re-based to bit 0
  auto fetch_bits = [&](std::vector<uint8_t>& dst, const uint8_t* src, int64_t bit_off, int64_t nbits) -> int {
    dst.assign((size_t)((nbits + 7) / 8) + 1, 0);
    if (!nbits) return PDX_OK;
    const int64_t b0 = bit_off >> 3, nb = ((bit_off + nbits + 7) >> 3) - b0;
    const int sh = (int)(bit_off & 7);
    std::vector<uint8_t> tmp((size_t)nb + 1, 0);
    PDX_TRY(fetch(tmp.data(), src + b0, (size_t)nb));
    for (int64_t k = 0; k < (nbits + 7) / 8; ++k) dst[(size_t)k] = sh ? (uint8_t)((tmp[(size_t)k] >> sh) | (tmp[(size_t)k + 1] << (8 - sh))) : tmp[(size_t)k];
    if (nbits & 7) dst[(size_t)((nbits - 1) / 8)] &= (uint8_t)((1u << (nbits & 7)) - 1u);
    return PDX_OK;
  };
  for (int c = 0; c < ncols; ++c) {
    PDX_TRY(check_column(&cols[c], "pdx_parquet_write"));
    if (cols[c].length != n) return fail(PDX_INVALID, "pdx_parquet_write: all columns must have the same length");
    if (!names[c]) return fail(PDX_INVALID, "pdx_parquet_write: null column name");
    const int dt = cols[c].dtype;
    if (dt < PDX_INT64 || dt > PDX_TIMESTAMP_NS) return fail(PDX_INVALID, "pdx_parquet_write: unknown dtype");
    const bool optional = validity_or_null(&cols[c]) != nullptr;
    std::vector<uint8_t> valid, bits;
    std::vector<uint64_t> vals;
    if (optional) PDX_TRY(fetch_bits(valid, static_cast<const uint8_t*>(cols[c].validity), cols[c].offset, n));
    if (dt == PDX_BOOL) PDX_TRY(fetch_bits(bits, static_cast<const uint8_t*>(cols[c].values), cols[c].offset, n));
    else {
      vals.resize((size_t)n);
      PDX_TRY(fetch(vals.data(), static_cast<const uint64_t*>(cols[c].values) + cols[c].offset, (size_t)n * 8));
    }
    chunks[(size_t)c].first_page = (int64_t)file.size();
    for (int64_t r0 = 0; r0 < n || (n == 0 && r0 == 0); r0 += kPageRows) {
      const int64_t rows = std::min<int64_t>(kPageRows, n - r0);
      std::vector<uint8_t> payload;
      int64_t nonnull = rows;
      if (optional) {
        const int64_t groups = (rows + 7) / 8;
        ThriftOut hv;
        hv.varint(((uint64_t)groups << 1) | 1u);  // ONE bit-packed run: `groups` groups of 8 one-bit levels = the validity bytes
        const uint32_t len = (uint32_t)(hv.b.size() + (size_t)groups);
        payload.resize(4);
        memcpy(payload.data(), &len, 4);
        payload.insert(payload.end(), hv.b.begin(), hv.b.end());
        const uint8_t* vb = valid.data() + r0 / 8;
        payload.insert(payload.end(), vb, vb + groups);
        if (rows & 7) payload.back() &= (uint8_t)((1u << (rows & 7)) - 1u);
        nonnull = 0;
        for (int64_t k = 0; k < groups; ++k) nonnull += __builtin_popcount(payload[payload.size() - (size_t)groups + (size_t)k]);
        chunks[(size_t)c].nulls += rows - nonnull;
      }
      auto is_valid = [&](int64_t i) { return !optional || ((valid[(size_t)(i >> 3)] >> (i & 7)) & 1); };
      if (dt == PDX_BOOL) {  // PLAIN booleans: the non-null values bit-packed, LSB first
        const size_t at = payload.size();
        payload.resize(at + (size_t)((nonnull + 7) / 8), 0);
        int64_t j = 0;
        for (int64_t i = r0; i < r0 + rows; ++i)
          if (is_valid(i)) {
            if ((bits[(size_t)(i >> 3)] >> (i & 7)) & 1) payload[at + (size_t)(j >> 3)] |= (uint8_t)(1u << (j & 7));
            ++j;
          }
      } else {  // PLAIN 8-byte values of the non-null rows
        const size_t at = payload.size();
        payload.resize(at + (size_t)nonnull * 8);
        if (nonnull == rows) {
          if (rows) memcpy(payload.data() + at, vals.data() + r0, (size_t)rows * 8);  // (an empty page: both pointers may be null)
        } else {
          int64_t j = 0;
          for (int64_t i = r0; i < r0 + rows; ++i)
            if (is_valid(i)) memcpy(payload.data() + at + (size_t)(j++) * 8, &vals[(size_t)i], 8);
        }
      }
      if (payload.size() > 0x7FFFFFF0u) return fail(PDX_INVALID, "pdx_parquet_write: page too large");
      ThriftOut ph;  // PageHeader
      ph.i32(1, 0);  // DATA_PAGE
      ph.i32(2, (int32_t)payload.size());
      ph.i32(3, (int32_t)payload.size());
      ph.begin_field_struct(5);  // DataPageHeader
      ph.i32(1, (int32_t)rows);
      ph.i32(2, kEncPlain);
      ph.i32(3, kEncRle);
      ph.i32(4, kEncRle);
      ph.end_struct();
      ph.b.push_back(0);
      file.insert(file.end(), ph.b.begin(), ph.b.end());
      file.insert(file.end(), payload.begin(), payload.end());
      if (n == 0) break;
    }
    chunks[(size_t)c].bytes = (int64_t)file.size() - chunks[(size_t)c].first_page;
  }
  // ---- footer
  ThriftOut f;
  f.i32(1, 2);  // version
  f.list(2, 12, (size_t)ncols + 1);
  f.begin_elem_struct();  // root
  f.str(4, "schema");
  f.i32(5, ncols);
  f.end_struct();
  for (int c = 0; c < ncols; ++c) {
    const int dt = cols[c].dtype;
    f.begin_elem_struct();
    f.i32(1, dt == PDX_BOOL ? kPqBool : dt == PDX_FLOAT64 ? kPqDouble : kPqInt64);
    f.i32(3, validity_or_null(&cols[c]) ? 1 : 0);  // OPTIONAL / REQUIRED
    f.str(4, names[c]);
    if (dt == PDX_UINT64) f.i32(6, 14);  // ConvertedType UINT_64
    if (dt == PDX_UINT64) {
      f.begin_field_struct(10);  // LogicalType
      f.begin_field_struct(10);  // INTEGER
      f.i8(1, 64);
      f.boolean(2, false);
      f.end_struct();
      f.end_struct();
    } else if (dt == PDX_TIMESTAMP_NS) {
      f.begin_field_struct(10);
      f.begin_field_struct(8);  // TIMESTAMP
      f.boolean(1, false);      // isAdjustedToUTC
      f.begin_field_struct(2);  // unit
      f.begin_field_struct(3);  // NANOS
      f.end_struct();
      f.end_struct();
      f.end_struct();
      f.end_struct();
    }
    f.end_struct();
  }
  f.i64(3, n);
  f.list(4, 12, 1);
  f.begin_elem_struct();  // RowGroup
  f.list(1, 12, (size_t)ncols);
  int64_t total = 0;
  for (int c = 0; c < ncols; ++c) {
    const ChunkInfo& ci = chunks[(size_t)c];
    const int dt = cols[c].dtype;
    total += ci.bytes;
    f.begin_elem_struct();  // ColumnChunk
    f.i64(2, ci.first_page);
    f.begin_field_struct(3);  // ColumnMetaData
    f.i32(1, dt == PDX_BOOL ? kPqBool : dt == PDX_FLOAT64 ? kPqDouble : kPqInt64);
    f.list(2, 5, 2);
    f.zigzag(kEncPlain);
    f.zigzag(kEncRle);
    f.list(3, 8, 1);
    f.varint(strlen(names[c]));
    f.b.insert(f.b.end(), names[c], names[c] + strlen(names[c]));
    f.i32(4, kCodecNone);
    f.i64(5, n);
    f.i64(6, ci.bytes);
    f.i64(7, ci.bytes);
    f.i64(9, ci.first_page);
    f.begin_field_struct(12);  // Statistics
    f.i64(3, ci.nulls);
    f.end_struct();
    f.end_struct();
    f.end_struct();
  }
  f.i64(2, total);
  f.i64(3, n);
  f.end_struct();
  f.str(6, "pdx-hip (PandasArrow MI355X backend)");
  f.b.push_back(0);
  file.insert(file.end(), f.b.begin(), f.b.end());
  const uint32_t flen = (uint32_t)f.b.size();
  file.insert(file.end(), reinterpret_cast<const uint8_t*>(&flen), reinterpret_cast<const uint8_t*>(&flen) + 4);
  file.insert(file.end(), {'P', 'A', 'R', '1'});
  void* blob = malloc(file.size());
  if (!blob) return fail(PDX_OOM, "pdx_parquet_write: host allocation failed");
  memcpy(blob, file.data(), file.size());
  *out_blob = blob;
  *out_size = file.size();
  return PDX_OK;
}
int pdx_parquet_free_blob(void* blob) {
  free(blob);
  return PDX_OK;
}

}  // extern "C"
