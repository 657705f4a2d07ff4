// groupby_extra.hip -- the remaining group-by aggregations of SURVEY.md 8(f)-3 on top of the grouped layout:
//   all / any        GROUPBY_NUMERIC_AGG(all, bool), GROUPBY_NUMERIC_AGG(any, bool)       reference src/dataframe.cpp:1520-1522
//   count_distinct   GROUPBY_NUMERIC_AGG(count_distinct, int64_t)                          reference src/dataframe.cpp:1526
// (GroupBy::min_max, src/dataframe.cpp:1602-1696, is MIN and MAX of one grouped pass: pdx_groupby_agg with two kinds.)
//
// Semantics (Arrow C++ 25.0.0 defaults, pinned by tests/golden/arrow_golden_r2.npz): all / any take BOOLEAN values, skip nulls,
// and are null for a group without a valid value; count_distinct counts the distinct VALID values of a group (CountOptions
// ONLY_VALID), where two float64 values are the same iff their bit patterns are (0.0 and -0.0 count twice, NaNs by payload).
//
// Both are order-free, so neither needs the row-order sort:
//   all / any       the bit-packed values become int64 0/1 (one 1/8 + 8 B/row pass) and go through the existing grouped
//                   int64 SUM + COUNT; all = (sum == count), any = (sum > 0), null when count == 0.
//   count_distinct  dictionary-encode the values with the group-by's own hash build (value id), form the composite key
//                   group id x V + value id, dictionary-encode THAT (its uniques are the distinct (group, value) pairs), and
//                   histogram the pairs by group.  Three dictionary builds, no sort, no per-group loop.
#include <memory>
#include <vector>
#include "pdx_common.hpp"

namespace pdx {

int launch_validity_and(const pdx_column* a, const pdx_column* b, int b_is_scalar, int64_t n, uint8_t* out, hipStream_t st);  // elementwise.hip

namespace {

__global__ void k_bool_to_i64(const uint8_t* __restrict__ bits, int64_t off, int64_t n, long long* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = bit_get(bits, off + i) ? 1 : 0;
}

// one thread per 64 groups: the result words of all / any and their validity
__global__ void k_all_any_finish(const long long* __restrict__ trues, const long long* __restrict__ valid_count, int64_t G,
                                 uint64_t* __restrict__ all_bits, uint64_t* __restrict__ all_ok, uint64_t* __restrict__ any_bits,
                                 uint64_t* __restrict__ any_ok, unsigned long long* __restrict__ null_groups) {
  int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t nwords = (G + 63) >> 6;
  if (w >= nwords) return;
  uint64_t a = 0, y = 0, ok = 0;
  for (int b = 0; b < 64; ++b) {
    int64_t g = (w << 6) + b;
    if (g >= G) break;
    long long c = valid_count[g], t = trues[g];  // (t is the slot of a NULL sum when c == 0: only trusted together with c > 0)
    if (c > 0) ok |= 1ull << b;
    if (c > 0 && t == c) a |= 1ull << b;
    if (c > 0 && t > 0) y |= 1ull << b;
  }
  if (all_bits) all_bits[w] = a;
  if (any_bits) any_bits[w] = y;
  if (all_ok) all_ok[w] = ok;
  if (any_ok) any_ok[w] = ok;
  int64_t in_word = G - (w << 6) < 64 ? G - (w << 6) : 64;
  int nulls = (int)in_word - __popcll(ok);
  if (nulls) atomicAdd(null_groups, (unsigned long long)nulls);
}

__global__ void k_composite(const uint32_t* __restrict__ gid, const uint32_t* __restrict__ vid, int64_t n, long long V, long long* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (long long)gid[i] * V + (long long)vid[i];
}

// the value id of the null value (the dictionary gives a null key its own group), or -1
__global__ void k_find_null_vid(const uint8_t* __restrict__ ok_bits, int64_t V, long long* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += stride)
    if (!bit_get(ok_bits, i)) *out = i;
}

__global__ void k_count_pairs(const long long* __restrict__ pairs, int64_t U, long long V, const long long* __restrict__ null_vid,
                              long long* __restrict__ out) {
  const long long nv = *null_vid;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < U; i += stride) {
    long long c = pairs[i];
    long long g = c / V, v = c - g * V;
    if (v != nv) atomicAdd(reinterpret_cast<unsigned long long*>(out + g), 1ull);
  }
}

struct HandleGuard {  // pdx_groupby_destroy on scope exit
  pdx_groupby* h = nullptr;
  ~HandleGuard() {
    if (h) pdx_groupby_destroy(h);
  }
};

int all_any(pdx_groupby* gb, const pdx_column* values, pdx_mut_column* out_all, pdx_mut_column* out_any, hipStream_t st) {
  const int64_t n = values->length, G = pdx_groupby_num_groups(gb);
  Scratch s;
  long long* as_i64 = s.get<long long>((size_t)n);
  const uint8_t* vvalid = validity_or_null(values);
  uint8_t* vbits = vvalid ? s.get<uint8_t>((size_t)(n + 7) / 8 + 16) : nullptr;
  long long* trues = s.get<long long>((size_t)G);
  long long* cnt = s.get<long long>((size_t)G);
  uint8_t* sum_ok = s.get<uint8_t>((size_t)(G + 7) / 8 + 16);
  unsigned long long* null_groups = s.get<unsigned long long>(1);
  PDX_SCRATCH_CHECK(s);
  if (n) {
    hipLaunchKernelGGL(k_bool_to_i64, dim3(grid_for(n, 256, 4)), dim3(256), 0, st, static_cast<const uint8_t*>(values->values), values->offset, n, as_i64);
    PDX_LAUNCH_CHECK();
    if (vbits) PDX_TRY(launch_validity_and(values, nullptr, 0, n, vbits, st));  // the validity bitmap re-based to offset 0
  }
  pdx_column tmp{};
  tmp.dtype = PDX_INT64;
  tmp.length = n;
  tmp.offset = 0;
  tmp.null_count = vbits ? -1 : 0;
  tmp.validity = vbits;
  tmp.values = as_i64;
  pdx_mut_column o[2] = {};
  o[0].dtype = PDX_INT64; o[0].length = G; o[0].values = trues; o[0].validity = sum_ok;
  o[1].dtype = PDX_INT64; o[1].length = G; o[1].values = cnt;
  const int kinds[2] = {PDX_AGG_SUM, PDX_AGG_COUNT};
  PDX_TRY(pdx_groupby_agg(gb, &tmp, kinds, 2, o, st));
  PDX_HIP(hipMemsetAsync(null_groups, 0, sizeof(unsigned long long), st));
  const int64_t nwords = (G + 63) >> 6;
  hipLaunchKernelGGL(k_all_any_finish, dim3((unsigned)ceil_div(nwords, 256)), dim3(256), 0, st, trues, cnt, G,
                     out_all ? static_cast<uint64_t*>(out_all->values) : nullptr, out_all ? static_cast<uint64_t*>(out_all->validity) : nullptr,
                     out_any ? static_cast<uint64_t*>(out_any->values) : nullptr, out_any ? static_cast<uint64_t*>(out_any->validity) : nullptr,
                     null_groups);
  PDX_LAUNCH_CHECK();
  unsigned long long h = 0;
  PDX_HIP(hipMemcpyAsync(&h, null_groups, sizeof(h), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
  for (pdx_mut_column* oc : {out_all, out_any})
    if (oc) {
      oc->length = G;
      oc->null_count = (int64_t)h;
    }
  return PDX_OK;
}

int count_distinct(pdx_groupby* gb, const pdx_column* values, pdx_mut_column* out, hipStream_t st) {
  const int64_t n = values->length, G = pdx_groupby_num_groups(gb);
  long long* counts = static_cast<long long*>(out->values);
  PDX_HIP(hipMemsetAsync(counts, 0, (size_t)G * sizeof(long long), st));
  out->length = G;
  out->null_count = 0;
  if (n == 0) return PDX_OK;
  // 1. value ids: the values as 64-bit patterns through the dictionary build (a null value is its own entry)
  pdx_column vkey = *values;
  vkey.dtype = PDX_INT64;
  HandleGuard gv;
  PDX_TRY(pdx_groupby_create(&vkey, st, &gv.h));
  const int64_t V = pdx_groupby_num_groups(gv.h);
  Scratch s;
  uint32_t* vid = s.get<uint32_t>((size_t)n);
  uint32_t* gid = s.get<uint32_t>((size_t)n);
  long long* comp = s.get<long long>((size_t)n);
  long long* vuniq = s.get<long long>((size_t)V);
  uint8_t* vok = s.get<uint8_t>((size_t)(V + 7) / 8 + 16);
  long long* null_vid = s.get<long long>(1);
  PDX_SCRATCH_CHECK(s);
  PDX_TRY(pdx_groupby_group_ids(gv.h, vid, st));
  pdx_mut_column vu{};
  vu.dtype = PDX_INT64; vu.length = V; vu.values = vuniq; vu.validity = vok;
  PDX_TRY(pdx_groupby_unique_keys(gv.h, &vu, st));
  const long long minus_one = -1;
  PDX_HIP(hipMemcpyAsync(null_vid, &minus_one, sizeof(long long), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_find_null_vid, dim3(grid_for(V, 256)), dim3(256), 0, st, vok, V, null_vid);
  PDX_LAUNCH_CHECK();
  // 2. composite key = group id x V + value id  (G, V < 2^31: no overflow)
  PDX_TRY(pdx_groupby_group_ids(gb, gid, st));
  hipLaunchKernelGGL(k_composite, dim3(grid_for(n, 256, 4)), dim3(256), 0, st, gid, vid, n, (long long)V, comp);
  PDX_LAUNCH_CHECK();
  pdx_column ckey{};
  ckey.dtype = PDX_INT64; ckey.length = n; ckey.values = comp;
  HandleGuard gc;
  PDX_TRY(pdx_groupby_create(&ckey, st, &gc.h));
  const int64_t U = pdx_groupby_num_groups(gc.h);
  // 3. the distinct (group, value) pairs, counted per group
  long long* pairs = s.get<long long>((size_t)U);
  uint8_t* pok = s.get<uint8_t>((size_t)(U + 7) / 8 + 16);
  PDX_SCRATCH_CHECK(s);
  pdx_mut_column pu{};
  pu.dtype = PDX_INT64; pu.length = U; pu.values = pairs; pu.validity = pok;
  PDX_TRY(pdx_groupby_unique_keys(gc.h, &pu, st));
  hipLaunchKernelGGL(k_count_pairs, dim3(grid_for(U, 256)), dim3(256), 0, st, pairs, U, (long long)V, null_vid, counts);
  PDX_LAUNCH_CHECK();
  PDX_HIP(hipStreamSynchronize(st));  // the scratch and the two dictionaries go back to the pool behind this point
  return PDX_OK;
}

}  // namespace

// Called by pdx_groupby_agg for a request that names ALL / ANY / COUNT_DISTINCT (or boolean values).  Standard kinds in the same
// request go back through pdx_groupby_agg in one grouped pass.
int groupby_agg_extra(pdx_groupby* gb, const pdx_column* values, const int* kinds, int nk, pdx_mut_column* outs, void* stream) {
  hipStream_t st = as_stream(stream);
  const int64_t G = pdx_groupby_num_groups(gb);
  std::vector<int> std_kinds;
  std::vector<pdx_mut_column> std_outs;
  std::vector<int> std_pos;
  pdx_mut_column *out_all = nullptr, *out_any = nullptr;
  for (int k = 0; k < nk; ++k) {
    pdx_mut_column* oc = &outs[k];
    if (kinds[k] == PDX_AGG_ALL || kinds[k] == PDX_AGG_ANY) {
      if (values->dtype != PDX_BOOL) return fail(PDX_INVALID, "pdx_groupby_agg: all / any need PDX_BOOL values");
      if (oc->dtype != PDX_BOOL || oc->length < G || !oc->values || !oc->validity)
        return fail(PDX_INVALID, "pdx_groupby_agg: all / any write a PDX_BOOL column with a validity buffer (a group without valid values is null)");
      (kinds[k] == PDX_AGG_ALL ? out_all : out_any) = oc;
    } else if (kinds[k] == PDX_AGG_COUNT_DISTINCT) {
      if (values->dtype == PDX_BOOL) return fail(PDX_NOT_IMPLEMENTED, "pdx_groupby_agg: count_distinct of boolean values is not implemented");
      if (oc->dtype != PDX_INT64 || oc->length < G || (G && !oc->values)) return fail(PDX_INVALID, "pdx_groupby_agg: count_distinct writes PDX_INT64");
    } else {
      if (values->dtype == PDX_BOOL) return fail(PDX_NOT_IMPLEMENTED, "pdx_groupby_agg: boolean values support all / any only");
      std_kinds.push_back(kinds[k]);
      std_outs.push_back(*oc);
      std_pos.push_back(k);
    }
  }
  if (!std_kinds.empty()) {
    PDX_TRY(pdx_groupby_agg(gb, values, std_kinds.data(), (int)std_kinds.size(), std_outs.data(), stream));
    for (size_t j = 0; j < std_pos.size(); ++j) outs[std_pos[j]] = std_outs[j];
  }
  if (out_all || out_any) {
    if (G == 0) {
      for (pdx_mut_column* oc : {out_all, out_any})
        if (oc) { oc->length = 0; oc->null_count = 0; }
    } else {
      PDX_TRY(all_any(gb, values, out_all, out_any, st));
    }
  }
  for (int k = 0; k < nk; ++k)
    if (kinds[k] == PDX_AGG_COUNT_DISTINCT) {
      if (G == 0) { outs[k].length = 0; outs[k].null_count = 0; }
      else PDX_TRY(count_distinct(gb, values, &outs[k], st));
    }
  return PDX_OK;
}

}  // namespace pdx
