// align.hip -- index alignment for binary operations on UNEQUAL indexes (the first caller-side "next" row).
//
// Reference: Series::broadcast (src/series.cpp:212-227) = Concatenate(index_a, index_b) -> Unique -> array_sort_indices
// (ascending) -> Take, then Series::reindex (src/series.cpp:1255-1309) of both operands onto that index: a host
// std::unordered_map<int64, int64> built with insert_or_assign (so the LAST position of a duplicated label wins) and one
// GetScalar/AppendScalar per new label; labels that are absent become null.
//
// Device form: both steps are the group-by dictionary (csrc/groupby.hip) applied to a concatenation.
//   union   : uniques of concat(a, b) (first-occurrence dictionary), then a stable LSD radix sort of the 64-bit labels
//             (sort_packed64 below: eight 8-bit passes over (low key half, high key half | row) elements, constant digits skipped)
//   reindex : group ids of concat(reverse(old), new); a new label is present iff its group's first row lies in the reversed old
//             part, and that first row is the LAST position of the label in the old index.  Output = take indices with a
//             validity bitmap (absent -> null); values then go through pdx_take, whose null indices yield null rows.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "compact.hpp"
#include "pdx/abi.h"
#include "pdx_common.hpp"
#include "radix_sort.hpp"

namespace pdx {

__global__ void k_concat2_i64(const long long* __restrict__ a, int64_t na, int reverse_a, const long long* __restrict__ b, int64_t nb,
                              long long* __restrict__ out) {
  const int64_t n = na + nb, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = i < na ? a[reverse_a ? na - 1 - i : i] : b[i - na];
}
__global__ void k_gather_labels(const long long* __restrict__ labels, const uint32_t* __restrict__ perm, int64_t n, long long* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = labels[perm[i]];
}
__global__ void k_reindex_emit(const uint32_t* __restrict__ gids, const int64_t* __restrict__ first_rows, int64_t n_old, int64_t n_new,
                               long long* __restrict__ out_idx, uint8_t* __restrict__ ok) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_new; j += stride) {
    const int64_t f = first_rows[gids[n_old + j]];
    const bool present = f < n_old;
    out_idx[j] = present ? n_old - 1 - f : 0;
    ok[j] = present;
  }
}

// ---- 64-bit stable sort with row numbers.  An element is (lo: low 32 key bits, pay: high 32 key bits << 32 | row): four LSD passes on
// `lo` carry the payload along, four more take their digit from the payload's own high half (k_radix_scatter PAYLOAD_DIGIT), so
// the keys are never gathered through a permutation (the three-round form did: k_label_chunk read keys[perm[i]], ten 16 B/row
// passes + four random gathers = 13.5 ms per 1e8 rows).  Digits that are equal in every key (OR == AND over the column) are skipped.

// the order-preserving unsigned image of a value (ascending: as is, descending: complemented) and its class (0 number, 1 NaN,
// 2 null): numbers first, then NaNs, then nulls in BOTH orders, as Arrow's array_sort_indices places them
__device__ __forceinline__ unsigned long long sort_image(unsigned long long u, bool is_null, int dtype, int descending, int* cls) {
  *cls = 0;
  if (is_null) {
    *cls = 2;
    return 0;
  }
  if (dtype == PDX_FLOAT64) {
    const double x = __longlong_as_double((long long)u);
    if (x != x) {
      *cls = 1;
      return 0;
    }
    if (x == 0.0) u = 0;  // -0.0 and 0.0 compare equal: one key, so the stable sort keeps their row order
    u = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
  } else if (dtype != PDX_UINT64) {
    u ^= 0x8000000000000000ull;  // int64 / timestamp
  }
  return descending ? ~u : u;
}
struct SortStats {
  unsigned long long bits_or, bits_and;  // over the keys of class 0
  unsigned long long others;             // rows of class 1 / 2
};
__device__ __forceinline__ void sort_stats_flush(SortStats* st, unsigned long long o, unsigned long long a, unsigned long long others) {
  for (int d = 32; d >= 1; d >>= 1) {
    o |= __shfl_xor(o, d, 64);
    a &= __shfl_xor(a, d, 64);
    others += __shfl_xor(others, d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicOr(&st->bits_or, o);
    atomicAnd(&st->bits_and, a);
    if (others) atomicAdd(&st->others, others);
  }
}
// every row: class byte (cls8 != nullptr) and -- when lo / pay are given -- the packed element at the row's own position
__global__ void k_sort_pack(const unsigned long long* __restrict__ v, const uint8_t* __restrict__ valid, int64_t off, int64_t n, int dtype, int descending,
                            unsigned long long flip, uint8_t* __restrict__ cls8, uint32_t* __restrict__ lo, unsigned long long* __restrict__ pay,
                            SortStats* __restrict__ stats) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long o = 0, a = ~0ull, others = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int c;
    const unsigned long long u = sort_image(v[i] ^ flip, valid && !bit_get(valid, off + i), dtype, descending, &c);
    if (cls8) cls8[i] = (uint8_t)c;
    if (c == 0) {
      o |= u;
      a &= u;
      if (lo) {
        lo[i] = (uint32_t)u;
        pay[i] = (u & 0xFFFFFFFF00000000ull) | (unsigned long long)i;
      }
    } else {
      ++others;
    }
  }
  sort_stats_flush(stats, o, a, others);
}
struct ClassPred {
  const uint8_t* cls8;
  int c;
  __device__ bool operator()(int64_t i) const { return cls8[i] == c; }
};
struct PackEmit {  // the numbers, compacted in row order
  const unsigned long long* v;
  int dtype, descending;
  uint32_t* lo;
  unsigned long long* pay;
  __device__ void operator()(int64_t pos, int64_t i) const {
    int c;
    const unsigned long long u = sort_image(v[i], false, dtype, descending, &c);
    lo[pos] = (uint32_t)u;
    pay[pos] = (u & 0xFFFFFFFF00000000ull) | (unsigned long long)i;
  }
};
struct RowEmit {  // NaN / null rows behind the numbers, in row order
  unsigned long long* out;
  int64_t base;
  __device__ void operator()(int64_t pos, int64_t i) const { out[base + pos] = (unsigned long long)i; }
};
__global__ void k_pay_rows_u64(const unsigned long long* __restrict__ pay, int64_t n, unsigned long long* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = pay[i] & 0xFFFFFFFFull;
}
__global__ void k_pay_rows_u32(const unsigned long long* __restrict__ pay, int64_t n, uint32_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (uint32_t)pay[i];
}
// Sorts the n packed elements (lo0, pay0) by their 64-bit key; `varying` = key bits that differ somewhere.  *sorted points at the
// payloads in key order (one of the ping-pong buffers).
static int sort_packed64(uint32_t* lo0, unsigned long long* pay0, int64_t n, unsigned long long varying, Scratch& s, hipStream_t st,
                         const unsigned long long** sorted) {
  *sorted = pay0;
  if (n <= 1 || varying == 0) return PDX_OK;
  const int64_t ntiles = ceil_div(n, kSortTile), nchunks = ceil_div(ntiles, kColChunk);
  uint32_t* lo1 = s.get<uint32_t>((size_t)n);
  unsigned long long* pay1 = s.get<unsigned long long>((size_t)n);
  uint32_t* hist = s.get<uint32_t>((size_t)ntiles << 8);
  uint32_t* chunk = s.get<uint32_t>((size_t)(nchunks + 1) << 8);
  PDX_SCRATCH_CHECK(s);
  const uint32_t* kin = lo0;
  const unsigned long long* vin = pay0;
  int last_lo = -1;
  for (int d = 0; d < 4; ++d)
    if ((varying >> (8 * d)) & 0xFFull) last_lo = d;
  for (int d = 0; d < 8; ++d) {
    if (!((varying >> (8 * d)) & 0xFFull)) continue;
    unsigned long long* vout = vin == pay0 ? pay1 : pay0;
    if (d < 4) {
      uint32_t* kout = kin == lo0 ? lo1 : lo0;
      PDX_TRY((radix_pass_dispatch<uint64_t>(8, kin, reinterpret_cast<const uint64_t*>(vin), kout, reinterpret_cast<uint64_t*>(vout), n, 8 * d, d != last_lo,
                                             hist, chunk, st)));
      if (d != last_lo) kin = kout;
    } else {
      PDX_TRY((radix_pass_payload_hi<8>(reinterpret_cast<const uint64_t*>(vin), reinterpret_cast<uint64_t*>(vout), n, 8 * (d - 4), hist, chunk, st)));
    }
    vin = vout;
  }
  *sorted = vin;
  return PDX_OK;
}
// perm[i] = index of the i-th smallest label (labels ^ flip in unsigned order), stable
static int argsort_labels64(const long long* labels, int64_t n, unsigned long long flip, uint32_t* perm, Scratch& s, hipStream_t st) {
  uint32_t* lo = s.get<uint32_t>((size_t)n);
  unsigned long long* pay = s.get<unsigned long long>((size_t)n);
  SortStats* stats = s.get<SortStats>(1);
  PDX_SCRATCH_CHECK(s);
  const SortStats init{0ull, ~0ull, 0ull};
  SortStats h{};
  PDX_HIP(hipMemcpyAsync(stats, &init, sizeof(init), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_sort_pack, dim3(grid_for(n, 256, 4)), dim3(256), 0, st, reinterpret_cast<const unsigned long long*>(labels), (const uint8_t*)nullptr,
                     (int64_t)0, n, (int)PDX_UINT64, 0, flip, (uint8_t*)nullptr, lo, pay, stats);
  PDX_LAUNCH_CHECK();
  PDX_HIP(hipMemcpyAsync(&h, stats, sizeof(h), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
  const unsigned long long* sorted = nullptr;
  PDX_TRY(sort_packed64(lo, pay, n, h.bits_or ^ h.bits_and, s, st, &sorted));
  hipLaunchKernelGGL(k_pay_rows_u32, dim3(grid_for(n, 256, 4)), dim3(256), 0, st, sorted, n, perm);
  PDX_LAUNCH_CHECK();
  return PDX_OK;
}

static int check_index(const pdx_column* c, const char* what) {
  PDX_TRY(check_column(c, what));
  if (!is_int_like(c->dtype)) return fail(PDX_NOT_IMPLEMENTED, std::string(what) + ": index must be int64 / uint64 / timestamp[ns]");
  if (validity_or_null(c)) return fail(PDX_NOT_IMPLEMENTED, std::string(what) + ": null index labels are not supported");
  return PDX_OK;
}
struct GroupByOwner {  // RAII for the internal dictionary handle
  pdx_groupby* h = nullptr;
  ~GroupByOwner() {
    if (h) pdx_groupby_destroy(h);
  }
};

__global__ void k_mark_groups(const uint32_t* __restrict__ gids, int64_t from, int64_t n, uint8_t* __restrict__ flag) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = from + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) flag[gids[i]] = 1;
}
// groups are numbered by first occurrence in concat(reverse(a), b): walking them backwards walks a's LAST positions upwards
struct IntersectPred {
  const int64_t* first_rows;
  const uint8_t* in_b;
  int64_t G, n_a;
  __device__ bool operator()(int64_t i) const {
    const int64_t g = G - 1 - i;
    return first_rows[g] < n_a && in_b[g];
  }
};
struct IntersectEmit {
  const int64_t* first_rows;
  const long long* a;
  int64_t G, n_a;
  long long* out;
  __device__ void operator()(int64_t pos, int64_t i) const { out[pos] = a[n_a - 1 - first_rows[G - 1 - i]]; }
};

}  // namespace pdx

using namespace pdx;

extern "C" {

int pdx_index_union(const pdx_column* a, const pdx_column* b, int sort, pdx_mut_column* out, void* stream) {
  PDX_TRY(check_index(a, "pdx_index_union"));
  PDX_TRY(check_index(b, "pdx_index_union"));
  if (!out) return fail(PDX_INVALID, "pdx_index_union: null output");
  if (a->dtype != b->dtype) return fail(PDX_INVALID, "type(NewIndex) != type(CurrentIndex).");
  const int64_t na = a->length, nb = b->length, n = na + nb;
  if (out->dtype != a->dtype) return fail(PDX_INVALID, "pdx_index_union: output dtype must be the index dtype");
  if (out->length < n || (n && !out->values)) return fail(PDX_INVALID, "pdx_index_union: output must hold a.length + b.length labels");
  out->length = 0;
  out->null_count = 0;
  if (n == 0) return PDX_OK;
  if (n > 0x7FFFFFFFll) return fail(PDX_NOT_IMPLEMENTED, "pdx_index_union: more than 2^31-1 labels per call is not supported yet");
  hipStream_t st = as_stream(stream);
  Scratch s;
  long long* cat = s.get<long long>((size_t)n);
  PDX_SCRATCH_CHECK(s);
  hipLaunchKernelGGL(k_concat2_i64, dim3(grid_for(n, 256, 4)), dim3(256), 0, st, static_cast<const long long*>(a->values) + a->offset, na, 0,
                     static_cast<const long long*>(b->values) + b->offset, nb, cat);
  PDX_LAUNCH_CHECK();
  // Unique (first-occurrence dictionary)
  pdx_column cc{};
  cc.dtype = a->dtype;
  cc.length = n;
  cc.values = cat;
  GroupByOwner gb;
  PDX_TRY(pdx_groupby_create(&cc, stream, &gb.h));
  const int64_t G = pdx_groupby_num_groups(gb.h);
  long long* uniq = s.get<long long>((size_t)G);
  uint8_t* uniq_ok = s.get<uint8_t>((size_t)(G + 7) / 8 + 8);
  PDX_SCRATCH_CHECK(s);
  pdx_mut_column um{};
  um.dtype = a->dtype;
  um.length = G;
  um.values = uniq;
  um.validity = uniq_ok;
  PDX_TRY(pdx_groupby_unique_keys(gb.h, &um, stream));
  if (!sort) {  // Series::union_ (src/series.cpp:782-798): Unique(Concatenate) as is, first-occurrence order
    PDX_HIP(hipMemcpyAsync(out->values, uniq, (size_t)G * sizeof(long long), hipMemcpyDeviceToDevice, st));
    if (out->validity) PDX_HIP(hipMemsetAsync(out->validity, 0xFF, (size_t)((G + 7) / 8), st));
    PDX_HIP(hipStreamSynchronize(st));
    out->length = G;
    return PDX_OK;
  }
  // array_sort_indices ascending + Take: stable LSD sort of the 64-bit labels, three rounds through the 32-bit pair sort
  uint32_t* perm = s.get<uint32_t>((size_t)G);
  PDX_SCRATCH_CHECK(s);
  const unsigned long long flip = a->dtype == PDX_UINT64 ? 0ull : 0x8000000000000000ull;
  PDX_TRY(argsort_labels64(uniq, G, flip, perm, s, st));
  hipLaunchKernelGGL(k_gather_labels, dim3(grid_for(G, 256, 4)), dim3(256), 0, st, uniq, perm, G, static_cast<long long*>(out->values));
  PDX_LAUNCH_CHECK();
  if (out->validity) PDX_HIP(hipMemsetAsync(out->validity, 0xFF, (size_t)((G + 7) / 8), st));
  PDX_HIP(hipStreamSynchronize(st));
  out->length = G;
  return PDX_OK;
}

int pdx_argsort(const pdx_column* col, int ascending, pdx_mut_column* out, void* stream) {
  PDX_TRY(check_column(col, "pdx_argsort"));
  if (!out) return fail(PDX_INVALID, "pdx_argsort: null output");
  if (col->dtype != PDX_INT64 && col->dtype != PDX_UINT64 && col->dtype != PDX_FLOAT64 && col->dtype != PDX_TIMESTAMP_NS)
    return fail(PDX_NOT_IMPLEMENTED, "pdx_argsort: int64 / uint64 / float64 / timestamp[ns] columns only");
  const int64_t n = col->length;
  if (out->dtype != PDX_UINT64) return fail(PDX_INVALID, "pdx_argsort: the indices are uint64");
  if (out->length < n || (n && !out->values)) return fail(PDX_INVALID, "pdx_argsort: output too small");
  if (n > 0x7FFFFFFFll) return fail(PDX_NOT_IMPLEMENTED, "pdx_argsort: more than 2^31-1 rows per call is not supported yet");
  out->length = n;
  out->null_count = 0;
  if (n == 0) return PDX_OK;
  hipStream_t st = as_stream(stream);
  Scratch s;
  const uint8_t* valid = validity_or_null(col);
  const bool classes = valid || col->dtype == PDX_FLOAT64;  // NaNs / nulls go behind the numbers, in row order
  const unsigned long long* v = static_cast<const unsigned long long*>(col->values) + col->offset;
  unsigned long long* outp = static_cast<unsigned long long*>(out->values);
  uint32_t* lo = s.get<uint32_t>((size_t)n);
  unsigned long long* pay = s.get<unsigned long long>((size_t)n);
  uint8_t* cls8 = classes ? s.get<uint8_t>((size_t)n) : nullptr;
  SortStats* stats = s.get<SortStats>(1);
  PDX_SCRATCH_CHECK(s);
  const SortStats init{0ull, ~0ull, 0ull};
  SortStats h{};
  PDX_HIP(hipMemcpyAsync(stats, &init, sizeof(init), hipMemcpyHostToDevice, st));
  // one pass: class bytes, the packed elements at their own positions (all that is needed when every row is a number), key statistics
  hipLaunchKernelGGL(k_sort_pack, dim3(grid_for(n, 256, 4)), dim3(256), 0, st, v, valid, col->offset, n, col->dtype, ascending ? 0 : 1, 0ull, cls8, lo, pay,
                     stats);
  PDX_LAUNCH_CHECK();
  PDX_HIP(hipMemcpyAsync(&h, stats, sizeof(h), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
  int64_t n0 = n;
  if (h.others) {  // numbers compacted to the front (row order), NaN rows, then null rows straight into the output
    int64_t n1 = 0, n2 = 0;
    PDX_TRY(compact_indices(n, ClassPred{cls8, 0}, PackEmit{v, col->dtype, ascending ? 0 : 1, lo, pay}, &n0, s, st));
    PDX_TRY(compact_indices(n, ClassPred{cls8, 1}, RowEmit{outp, n0}, &n1, s, st));
    PDX_TRY(compact_indices(n, ClassPred{cls8, 2}, RowEmit{outp, n0 + n1}, &n2, s, st));
    if (n0 + n1 + n2 != n) return fail(PDX_DEVICE, "pdx_argsort: class counts do not add up");
  }
  const unsigned long long* sorted = nullptr;
  PDX_TRY(sort_packed64(lo, pay, n0, h.bits_or ^ h.bits_and, s, st, &sorted));
  if (n0) hipLaunchKernelGGL(k_pay_rows_u64, dim3(grid_for(n0, 256, 4)), dim3(256), 0, st, sorted, n0, outp);
  PDX_LAUNCH_CHECK();
  if (out->validity) PDX_HIP(hipMemsetAsync(out->validity, 0xFF, (size_t)((n + 7) / 8), st));
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}

int pdx_reindex_indices(const pdx_column* old_index, const pdx_column* new_index, pdx_mut_column* out_idx, void* stream) {
  PDX_TRY(check_index(old_index, "pdx_reindex_indices"));
  PDX_TRY(check_index(new_index, "pdx_reindex_indices"));
  if (!out_idx) return fail(PDX_INVALID, "pdx_reindex_indices: null output");
  if (old_index->dtype != new_index->dtype) return fail(PDX_INVALID, "type(NewIndex) != type(CurrentIndex).");
  const int64_t n_old = old_index->length, n_new = new_index->length, n = n_old + n_new;
  if (out_idx->dtype != PDX_INT64) return fail(PDX_INVALID, "pdx_reindex_indices: output must be int64 take indices");
  if (out_idx->length < n_new || (n_new && (!out_idx->values || !out_idx->validity)))
    return fail(PDX_INVALID, "pdx_reindex_indices: output needs new_index.length indices and a validity buffer");
  out_idx->length = n_new;
  out_idx->null_count = -1;
  if (n_new == 0) return PDX_OK;
  if (n > 0x7FFFFFFFll) return fail(PDX_NOT_IMPLEMENTED, "pdx_reindex_indices: more than 2^31-1 labels per call is not supported yet");
  hipStream_t st = as_stream(stream);
  Scratch s;
  long long* cat = s.get<long long>((size_t)n);
  uint32_t* gids = s.get<uint32_t>((size_t)n);
  uint8_t* ok = s.get<uint8_t>((size_t)n_new);
  PDX_SCRATCH_CHECK(s);
  hipLaunchKernelGGL(k_concat2_i64, dim3(grid_for(n, 256, 4)), dim3(256), 0, st, static_cast<const long long*>(old_index->values) + old_index->offset,
                     n_old, 1, static_cast<const long long*>(new_index->values) + new_index->offset, n_new, cat);
  PDX_LAUNCH_CHECK();
  pdx_column cc{};
  cc.dtype = old_index->dtype;
  cc.length = n;
  cc.values = cat;
  GroupByOwner gb;
  PDX_TRY(pdx_groupby_create(&cc, stream, &gb.h));
  const int64_t G = pdx_groupby_num_groups(gb.h);
  int64_t* first_rows = s.get<int64_t>((size_t)G);
  PDX_SCRATCH_CHECK(s);
  PDX_TRY(pdx_groupby_group_ids(gb.h, gids, stream));
  PDX_TRY(pdx_groupby_first_rows(gb.h, first_rows, stream));
  hipLaunchKernelGGL(k_reindex_emit, dim3(grid_for(n_new, 256, 4)), dim3(256), 0, st, gids, first_rows, n_old, n_new,
                     static_cast<long long*>(out_idx->values), ok);
  hipLaunchKernelGGL(k_pack_bytes, dim3(grid_for((n_new + 7) / 8, 256)), dim3(256), 0, st, ok, n_new, static_cast<uint8_t*>(out_idx->validity));
  PDX_LAUNCH_CHECK();
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}

int pdx_index_intersection(const pdx_column* a, const pdx_column* b, pdx_mut_column* out, void* stream) {
  PDX_TRY(check_index(a, "pdx_index_intersection"));
  PDX_TRY(check_index(b, "pdx_index_intersection"));
  if (!out) return fail(PDX_INVALID, "pdx_index_intersection: null output");
  if (a->dtype != b->dtype) return fail(PDX_INVALID, "type(NewIndex) != type(CurrentIndex).");
  const int64_t na = a->length, nb = b->length, n = na + nb;
  if (out->dtype != a->dtype) return fail(PDX_INVALID, "pdx_index_intersection: output dtype must be the index dtype");
  if (out->length < na || (na && !out->values)) return fail(PDX_INVALID, "pdx_index_intersection: output must hold a.length labels");
  out->length = 0;
  out->null_count = 0;
  if (na == 0 || nb == 0) return PDX_OK;
  if (n > 0x7FFFFFFFll) return fail(PDX_NOT_IMPLEMENTED, "pdx_index_intersection: more than 2^31-1 labels per call is not supported yet");
  hipStream_t st = as_stream(stream);
  Scratch s;
  long long* cat = s.get<long long>((size_t)n);
  uint32_t* gids = s.get<uint32_t>((size_t)n);
  PDX_SCRATCH_CHECK(s);
  const long long* av = static_cast<const long long*>(a->values) + a->offset;
  hipLaunchKernelGGL(k_concat2_i64, dim3(grid_for(n, 256, 4)), dim3(256), 0, st, av, na, 1, static_cast<const long long*>(b->values) + b->offset, nb, cat);
  PDX_LAUNCH_CHECK();
  pdx_column cc{};
  cc.dtype = a->dtype;
  cc.length = n;
  cc.values = cat;
  GroupByOwner gb;
  PDX_TRY(pdx_groupby_create(&cc, stream, &gb.h));
  const int64_t G = pdx_groupby_num_groups(gb.h);
  int64_t* first_rows = s.get<int64_t>((size_t)G);
  uint8_t* in_b = s.get<uint8_t>((size_t)G);
  PDX_SCRATCH_CHECK(s);
  PDX_TRY(pdx_groupby_group_ids(gb.h, gids, stream));
  PDX_TRY(pdx_groupby_first_rows(gb.h, first_rows, stream));
  PDX_HIP(hipMemsetAsync(in_b, 0, (size_t)G, st));
  hipLaunchKernelGGL(k_mark_groups, dim3(grid_for(nb, 256, 4)), dim3(256), 0, st, gids, na, n, in_b);
  PDX_LAUNCH_CHECK();
  int64_t m = 0;
  PDX_TRY(compact_indices(G, IntersectPred{first_rows, in_b, G, na}, IntersectEmit{first_rows, av, G, na, static_cast<long long*>(out->values)}, &m, s, st));
  if (out->validity) PDX_HIP(hipMemsetAsync(out->validity, 0xFF, (size_t)((m + 7) / 8), st));
  PDX_HIP(hipStreamSynchronize(st));
  out->length = m;
  return PDX_OK;
}

}  // extern "C"
