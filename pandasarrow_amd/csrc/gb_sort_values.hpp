// gb_sort_values.hpp -- part of groupby.hip (textually included there, in its namespace context; split by stage, kernels unchanged):
// values stably sorted by slot: classic and narrowing LSD sorts, dispatch of the dense reducers.
#pragma once


int minmax_i64_host(const long long* v, int64_t n, long long* mn, long long* mx, Scratch& s, hipStream_t st);  // aggregate.hip
int groupby_agg_extra(pdx_groupby* gb, const pdx_column* values, const int* kinds, int nk, pdx_mut_column* outs, void* stream);  // groupby_extra.hip
int minmax_keys_host(const long long* v, const uint8_t* valid, int64_t off, int64_t n, MinMaxPartial<long long>* out, Scratch& s,
                     hipStream_t st);  // aggregate.hip

static unsigned int next_pow2(uint64_t x) {
  uint64_t p = 16;
  while (p < x) p <<= 1;
  return (unsigned int)p;
}
static int ilog2(uint64_t x) {
  int b = 0;
  while ((1ull << b) < x) ++b;
  return b;
}

// Values (8-byte payload in ROW order, optional validity) stably sorted by logical slot.  `alloc` provides the buffers
// (scratch for pdx_groupby_agg, handle-owned for pdx_groupby_group_values).  In the partitioned layout the values are first
// scattered with the stored partition offsets (the first LSD pass) and only the remaining slot bits are sorted.
template <typename Alloc>
static int sort_values_by_slot(pdx_groupby* gb, const uint64_t* vals, const uint8_t* vvalid, int64_t voff, Alloc&& alloc, Scratch& s, hipStream_t st,
                               const uint32_t** keys_sorted, const uint64_t** vals_sorted, int skip_top_bits = 0) {
  // skip_top_bits: leave the rows sorted by the LOW slot_bits - skip_top_bits bits only (the fused last-digit reduce does the rest)
  const int64_t n = gb->n;
  uint32_t* k0 = static_cast<uint32_t*>(alloc((size_t)n * 4));
  uint32_t* k1 = static_cast<uint32_t*>(alloc((size_t)n * 4));
  uint64_t* v0 = static_cast<uint64_t*>(alloc((size_t)n * 8));
  uint64_t* v1 = static_cast<uint64_t*>(alloc((size_t)n * 8));
  if (!k0 || !k1 || !v0 || !v1) return PDX_OOM;
  if (gb->slot_part) {
    PDX_TRY(ensure_slot_part(gb, st));
    uint64_t* vals_part = static_cast<uint64_t*>(alloc((size_t)n * 8));
    if (!vals_part) return PDX_OOM;
    PDX_TRY((radix_scatter_only<kPartBits, uint64_t, uint8_t>(gb->bucket8, vals, nullptr, vals_part, n, 0, false, gb->part_off, st)));
    if (gb->digit2) {  // second partition level: one more stable scatter with the stored digits / offsets
      uint64_t* vals_part2 = static_cast<uint64_t*>(alloc((size_t)n * 8));
      if (!vals_part2) return PDX_OOM;
      switch (gb->digit2_bits) {
        case 4: PDX_TRY((radix_scatter_only<4, uint64_t, uint8_t>(gb->digit2, vals_part, nullptr, vals_part2, n, 0, false, gb->part_off2, st))); break;
        case 5: PDX_TRY((radix_scatter_only<5, uint64_t, uint8_t>(gb->digit2, vals_part, nullptr, vals_part2, n, 0, false, gb->part_off2, st))); break;
        case 6: PDX_TRY((radix_scatter_only<6, uint64_t, uint8_t>(gb->digit2, vals_part, nullptr, vals_part2, n, 0, false, gb->part_off2, st))); break;
        case 7: PDX_TRY((radix_scatter_only<7, uint64_t, uint8_t>(gb->digit2, vals_part, nullptr, vals_part2, n, 0, false, gb->part_off2, st))); break;
        default: PDX_TRY((radix_scatter_only<8, uint64_t, uint8_t>(gb->digit2, vals_part, nullptr, vals_part2, n, 0, false, gb->part_off2, st))); break;
      }
      vals_part = vals_part2;
    }
    const uint32_t* kin = gb->slot_part;
    if (vvalid) {
      uint32_t* fk = static_cast<uint32_t*>(alloc((size_t)n * 4));
      if (!fk) return PDX_OOM;
      PDX_TRY(ensure_rows_part(gb, st));
      hipLaunchKernelGGL(k_flag_keys_part, dim3(grid_for(n, 256, 4)), dim3(256), 0, st, gb->slot_part, gb->rows_part, vvalid, voff, n, fk);
      kin = fk;
    }
    return radix_sort_pairs<uint64_t>(kin, vals_part, k0, v0, k1, v1, n, gb->slot_bits - gb->part_bits - skip_top_bits, keys_sorted, vals_sorted, true, s, st,
                                      gb->part_bits);
  }
  const uint32_t* kin = gb->slot_of_row;
  if (vvalid) {
    uint32_t* fk = static_cast<uint32_t*>(alloc((size_t)n * 4));
    if (!fk) return PDX_OOM;
    hipLaunchKernelGGL(k_flag_keys, dim3(grid_for(n, 256, 4)), dim3(256), 0, st, gb->slot_of_row, vvalid, voff, n, fk);
    kin = fk;
  }
  // (pass0_off describes the unflagged slots; the digit of a flagged key is the same: the flag lives in bit 31)
  return radix_sort_pairs<uint64_t>(kin, vals, k0, v0, k1, v1, n, gb->slot_bits - skip_top_bits, keys_sorted, vals_sorted, true, s, st, 0, gb->pass0_off);
}


// One pass of the narrowing sort, digit = the low `bits` of K (the digit width is a template parameter of the kernels).  offsets != nullptr:
// the scanned per-tile offsets of this pass exist already (pass 0: fused into the slot kernel); otherwise they are built in hist.
template <typename K, typename KO, bool FLAGS = false>
static int narrow_pass(int bits, const K* kin, const uint64_t* vin, KO* kout, uint64_t* vout, int64_t n, const uint32_t* offsets, uint32_t* hist,
                       uint32_t* chunk_sum, hipStream_t st, const uint8_t* valid = nullptr, int64_t valid_off = 0) {
#define NARROW_PASS(B)                                                                                   \
  {                                                                                                      \
    if (!offsets) PDX_TRY((radix_offsets<B, K>(kin, n, 0, hist, chunk_sum, true, st)));                    \
    return radix_scatter_narrow<B, uint64_t, K, KO, FLAGS>(kin, vin, kout, vout, n, offsets ? offsets : hist, st, valid, valid_off); \
  }
  switch (bits) {
    case 4: NARROW_PASS(4)
    case 5: NARROW_PASS(5)
    case 6: NARROW_PASS(6)
    case 7: NARROW_PASS(7)
    case 8: NARROW_PASS(8)
    default: return fail(PDX_INVALID, "narrowing sort: unsupported digit width");
  }
#undef NARROW_PASS
}
// Full stable sort of the values by dense slot with narrowing keys (three passes: 4 -> 2 -> 1 byte keys -> none) and every group's
// offset from the scatter offsets (two levels of k_level_starts): 22 + 19 + 17 B/row instead of 3 x 24 + 2 x 4 (histograms) and no
// search in sorted slots.  Returns PDX_OK with *done = false when the handle's layout does not fit (the caller takes the classic sort).
// `two` (the sharded sums' fused record emission, k_flr_emit): stop after the second pass -- the values stay sorted by the low b0 + b1 slot
// bits with the last digit beside them in a byte, the group offsets still come from the last digit's scatter offsets (its histogram is one
// pass over the bytes), and the kernel that reads the runs does the last digit itself.  finish_narrow_sort() makes up for the third pass.
struct NarrowTwo {
  const uint8_t* k8 = nullptr;   // last digit of every row (rows sorted by the low bits)
  const uint64_t* v1 = nullptr;  // the values in that order
  uint64_t* v0 = nullptr;        // target of the third pass, should it be needed
  uint32_t* run_start = nullptr; // starts of the 2^low_bits runs (+ n)
  int64_t nruns = 0;
  int low_bits = 0, last_bits = 0;
};
static int last_digit_offsets(int bits, const uint8_t* k8, int64_t n, uint32_t* hist, uint32_t* chunk, hipStream_t st) {
  switch (bits) {
    case 4: return radix_offsets<4, uint8_t>(k8, n, 0, hist, chunk, true, st);
    case 5: return radix_offsets<5, uint8_t>(k8, n, 0, hist, chunk, true, st);
    case 6: return radix_offsets<6, uint8_t>(k8, n, 0, hist, chunk, true, st);
    case 7: return radix_offsets<7, uint8_t>(k8, n, 0, hist, chunk, true, st);
    case 8: return radix_offsets<8, uint8_t>(k8, n, 0, hist, chunk, true, st);
    default: return fail(PDX_INVALID, "narrowing sort: unsupported digit width");
  }
}
static int finish_narrow_sort(const NarrowTwo& two, int64_t n, Scratch& s, hipStream_t st) {
  const int64_t ntiles = ceil_div(n, kSortTile), nchunks = ceil_div(ntiles, kColChunk);
  uint32_t* hist = s.get<uint32_t>((size_t)ntiles << 8);
  uint32_t* chunk = s.get<uint32_t>((size_t)(nchunks + 1) << 8);
  PDX_SCRATCH_CHECK(s);
  return narrow_pass<uint8_t, uint8_t>(two.last_bits, two.k8, two.v1, (uint8_t*)nullptr, two.v0, n, nullptr, hist, chunk, st);
}
template <typename Alloc>
static int sort_values_narrow_full(pdx_groupby* gb, const uint64_t* vin, Alloc&& alloc, Scratch& s, hipStream_t st, const uint64_t** vals_sorted,
                                   uint32_t* seg_start_out, bool* done, NarrowTwo* two = nullptr) {
  *done = false;
  const int64_t n = gb->n, G = gb->G;
  const SortPlan plan = make_sort_plan(gb->slot_bits, sort_max_bits());
  const bool env_ok = [] { const char* e = getenv("PDX_SORT_NARROW"); return !(e && e[0] == '0'); }();
  if (!env_ok || gb->slot_part || !gb->pass0_off || !gb->slot_of_row || plan.npasses != 3 || n < ((int64_t)1 << 22)) return PDX_OK;
  const int b0 = plan.bits[0], b1 = plan.bits[1], b2 = plan.bits[2];
  if (b0 > 8 || b1 > 8 || b2 > 8 || gb->slot_bits - b0 > 16 || b2 > 8 || gb->slot_bits != b0 + b1 + b2) return PDX_OK;
  const int64_t ntiles = ceil_div(n, kSortTile), nchunks = ceil_div(ntiles, kColChunk);
  const bool stop_at_two = two != nullptr && b2 <= 6;  // (= kFlrBits: one lane per value of the last digit in k_flr_emit; asserted there)
  uint16_t* k16 = s.get<uint16_t>((size_t)n);
  uint8_t* k8 = stop_at_two ? static_cast<uint8_t*>(alloc((size_t)n)) : s.get<uint8_t>((size_t)n);
  uint32_t* hist = s.get<uint32_t>((size_t)ntiles << 8);
  uint32_t* chunk = s.get<uint32_t>((size_t)(nchunks + 1) << 8);
  uint32_t* starts1 = stop_at_two ? static_cast<uint32_t*>(alloc((((size_t)1 << (b0 + b1)) + 1) * 4)) : s.get<uint32_t>(((size_t)1 << (b0 + b1)) + 1);
  uint32_t* slot_start = s.get<uint32_t>(((size_t)1 << gb->slot_bits) + 1);
  PDX_SCRATCH_CHECK(s);
  uint64_t* v0 = static_cast<uint64_t*>(alloc((size_t)n * 8));
  uint64_t* v1 = static_cast<uint64_t*>(alloc((size_t)n * 8));
  if (!v0 || !v1 || !k8 || !starts1) return PDX_OOM;
  PDX_TRY((narrow_pass<uint32_t, uint16_t>(b0, gb->slot_of_row, vin, k16, v0, n, gb->pass0_off, hist, chunk, st)));
  PDX_TRY((narrow_pass<uint16_t, uint8_t>(b1, k16, v0, k8, v1, n, nullptr, hist, chunk, st)));
  hipLaunchKernelGGL((k_level_starts<uint16_t>), dim3(1u << b0), dim3(256), 0, st, k16, n, gb->pass0_off, (int64_t)1 << b0, b0, b1, hist, starts1);
  if (stop_at_two) PDX_TRY(last_digit_offsets(b2, k8, n, hist, chunk, st));  // (the offsets of the third pass without its scatter)
  else PDX_TRY((narrow_pass<uint8_t, uint8_t>(b2, k8, v1, (uint8_t*)nullptr, v0, n, nullptr, hist, chunk, st)));
  hipLaunchKernelGGL((k_level_starts<uint8_t>), dim3((unsigned)std::min<int64_t>((int64_t)1 << (b0 + b1), 65536)), dim3(256), 0, st, k8, n, starts1,
                     (int64_t)1 << (b0 + b1), b0 + b1, b2, hist, slot_start);
  hipLaunchKernelGGL(k_seg_starts_from_slots, dim3(grid_for(G + 1, 256)), dim3(256), 0, st, slot_start, n, gb->occ_slot, G, seg_start_out);
  PDX_LAUNCH_CHECK();
  *vals_sorted = stop_at_two ? nullptr : v0;
  if (stop_at_two) {
    two->k8 = k8;
    two->v1 = v1;
    two->v0 = v0;
    two->run_start = starts1;
    two->nruns = (int64_t)1 << (b0 + b1);
    two->low_bits = b0 + b1;
    two->last_bits = b2;
  }
  *done = true;
  return PDX_OK;
}

template <typename T>
static int launch_seg_reduce_dense(const T* vals, const uint32_t* seg_start, int64_t nseg, const uint32_t* out_index, const SegOut& o,
                                   bool want_pw, bool want_mm, bool want_is, int64_t nrows, Scratch& s, hipStream_t st) {
  if (nseg == 0) return PDX_OK;
  // flag combination -> one of five instantiations (the mixed ones share <true, true, true>)
  const int combo = (want_pw && !want_mm && !want_is) ? 0 : (!want_pw && want_mm && !want_is) ? 1 : (!want_pw && !want_mm && want_is) ? 2
                    : (!want_pw && !want_mm && !want_is) ? 3 : 4;
#define SEG_DISPATCH(LAUNCH)          \
  switch (combo) {                    \
    case 0: LAUNCH(true, false, false); break;  \
    case 1: LAUNCH(false, true, false); break;  \
    case 2: LAUNCH(false, false, true); break;  \
    case 3: LAUNCH(false, false, false); break; \
    default: LAUNCH(true, true, true); break;   \
  }
  // ---- long groups first (their outputs are skipped by the per-group kernels below)
  if (nrows > kBigSeg) {
    const int64_t maxB = std::min<int64_t>(nseg, nrows / kBigSeg + 1);  // a long group has more than kBigSeg rows
    const int64_t max_items = nrows / kBigSeg + maxB;
    uint32_t* big_idx = s.get<uint32_t>((size_t)maxB);
    int64_t* nbig = s.get<int64_t>(1);
    int64_t* item_off = s.get<int64_t>((size_t)maxB + 1);
    SubState<T>* state = s.get<SubState<T>>((size_t)max_items);
    PDX_SCRATCH_CHECK(s);
    PDX_HIP(hipMemsetAsync(nbig, 0, sizeof(int64_t), st));
    hipLaunchKernelGGL(k_big_append, dim3(grid_for(nseg, 256)), dim3(256), 0, st, seg_start, nseg, big_idx, nbig);
    hipLaunchKernelGGL(k_big_offsets, dim3(1), dim3(256), 0, st, seg_start, big_idx, nbig, item_off);
    const int grid_sub = (int)std::min<int64_t>(ceil_div(max_items, kSegWaves), (int64_t)kCUs * 8);
#define SEG_SUB(PW, MM, IS)                                                                                                                      \
  hipLaunchKernelGGL((k_seg_reduce_sub<T, PW, MM, IS>), dim3(grid_sub), dim3(kSegWaves * 64), 0, st, vals, seg_start, big_idx, item_off, nbig, state); \
  hipLaunchKernelGGL((k_seg_combine_big<T, PW, MM, IS>), dim3((unsigned)maxB), dim3(64), 0, st, seg_start, big_idx, item_off, nbig, state, out_index, o)
    SEG_DISPATCH(SEG_SUB)
#undef SEG_SUB
    PDX_LAUNCH_CHECK();
  }
  // workgroups per CU: a multiple of what is resident at once (the 4-wave workgroups hold ~44 KB of LDS: 3 per CU), so that the waves'
  // static shares of the groups run in full rounds (8 per CU meant 3 + 3 + 2).  PDX_SEG_WGS_PER_CU: diagnostic.
  static const int seg_wgs_per_cu = [] { const char* e = getenv("PDX_SEG_WGS_PER_CU"); return e && atoi(e) > 0 ? atoi(e) : 9; }();
  int grid = (int)std::min<int64_t>(ceil_div(nseg, kSegWaves), (int64_t)kCUs * seg_wgs_per_cu);
  dim3 g(grid), b(kSegWaves * 64);
  // mostly short groups: one kernel that batches the groups of <= kMidLen rows per wave and chunks through the longer ones
  const int64_t mid_max = [] { const char* e = getenv("PDX_SEG_MID_MAX"); return e ? atoll(e) : 1100ll; }();
  const int64_t min_len = -1;
  if (nrows / nseg < mid_max) {
    const int64_t nwaves = (int64_t)kCUs * seg_wgs_per_cu * kSegWaves;
    // groups per wave: at least a batch of short groups (64), but with long groups few of them are a wave's worth of rows already -- a
    // thousand 1000-row groups are a thousand waves, not sixteen
    const int64_t avg_rows = std::max<int64_t>(1, nrows / nseg);
    const int64_t min_gpw = std::max<int64_t>(1, std::min<int64_t>(64, 4096 / avg_rows));
    const int64_t gpw = std::max<int64_t>(min_gpw, ceil_div(nseg, nwaves));
    const int grid_mid = (int)ceil_div(ceil_div(nseg, gpw), kSegWaves);
#define SEG_MID(PW, MM, IS) \
  hipLaunchKernelGGL((k_seg_reduce_mid<T, PW, MM, IS>), dim3(grid_mid), b, 0, st, vals, seg_start, nseg, out_index, o, gpw)
    SEG_DISPATCH(SEG_MID)
#undef SEG_MID
    PDX_LAUNCH_CHECK();
    return PDX_OK;
  }
#define SEG_LAUNCH(PW, MM, IS) hipLaunchKernelGGL((k_seg_reduce<T, PW, MM, IS>), g, b, 0, st, vals, seg_start, nseg, out_index, o, min_len)
  SEG_DISPATCH(SEG_LAUNCH)
#undef SEG_LAUNCH
#undef SEG_DISPATCH
  PDX_LAUNCH_CHECK();
  return PDX_OK;
}
