// gb_layout.hpp -- part of groupby.hip: the grouped layout of ONE value column (GroupedLayout, gb_handle.hpp) and how it is built.
//
// Reference: GroupBy's constructor materialises every column's per-group arrays once (processEach, src/dataframe.cpp:1539-1554:
// Grouper::MakeGroupings + ApplyGroupings) and sum() / mean() / count() reuse them (src/group_by.h:85-139).  Here the same
// split: build_layout sorts the values by slot (stage 1), the reducers in groupby.hip consume the layout (stage 2); a bound column
// (pdx_groupby_bind) keeps its layout in the handle, so the reference's call pattern gb.sum(c); gb.mean(c); gb.count(c) sorts once.
#pragma once

// Diagnostic switches of pdx_groupby_agg, read once per call (tests flip them inside one process).  Defaults are the product path;
// pdx_groupby_last_plan reports the path a call actually took, so a test can assert it instead of trusting the thresholds.
struct AggTuning {
  bool fused = true;        // PDX_FUSED_LAST_DIGIT=0: never fuse the last digit into the reduce
  bool fused_hash = true;   // PDX_FUSED_LAST_DIGIT_HASH=0: not on hash-partitioned slots
  bool narrow = true;       // PDX_SORT_NARROW=0: 4-byte slots through every pass
  bool null_pw = false;     // PDX_FLR_NULL_PW=1: thread-per-leaf form for nullable sum / mean / count
  int64_t min_rows = (int64_t)1 << 22;  // PDX_FUSED_LAST_DIGIT_MIN_ROWS
  // PDX_FUSED_LAST_DIGIT_MIN_RUN: a run is one workgroup's sequential work (1e8 groups: 119-row runs).  Was 8192 until round 4: the
  // 1e6-key query at an 8-GPU shard's size (1.25e8 rows: 7629-row runs) fell just below it and took the classic full sort, 4.18 ms
  // against 2.69 ms fused; fused also wins at 5e7 / 2e7 / 3e7 rows (tools/sweep_min_run.sh: 2.25 -> 1.61, 1.60 -> 1.28, 1.46 -> 1.07 ms)
  int64_t min_run = 1024;
  int min_low = 10;         // PDX_FUSED_LAST_DIGIT_MIN_LOW_BITS
  int flr_wgs_per_cu = 24;  // PDX_FLR_WGS_PER_CU
  int fw_wgs_per_cu = 48;   // PDX_FLR_WAVE_WGS_PER_CU
  int wave_force = -1;      // PDX_FLR_WAVE=0 / 1: force the workgroup / wave form of the fused kernel
  bool hybrid = true;       // PDX_FLR_HYBRID=0: any run over the limit sends the whole column down the classic path
  unsigned int max_run = 1u << 19;  // PDX_FLR_MAX_RUN (<= 2^19): rows of the longest run the fused kernels take
  static AggTuning read() {
    AggTuning t;
    auto off = [](const char* name) { const char* e = getenv(name); return e && e[0] == '0'; };
    auto on = [](const char* name) { const char* e = getenv(name); return e && e[0] == '1'; };
    t.fused = !off("PDX_FUSED_LAST_DIGIT");
    t.fused_hash = !off("PDX_FUSED_LAST_DIGIT_HASH");
    t.narrow = !off("PDX_SORT_NARROW");
    t.null_pw = on("PDX_FLR_NULL_PW");
    if (const char* e = getenv("PDX_FUSED_LAST_DIGIT_MIN_ROWS")) t.min_rows = atoll(e);
    if (const char* e = getenv("PDX_FUSED_LAST_DIGIT_MIN_RUN")) t.min_run = atoll(e);
    if (const char* e = getenv("PDX_FUSED_LAST_DIGIT_MIN_LOW_BITS")) t.min_low = atoi(e);
    if (const char* e = getenv("PDX_FLR_WGS_PER_CU")) if (atoi(e) > 0) t.flr_wgs_per_cu = atoi(e);
    if (const char* e = getenv("PDX_FLR_WAVE_WGS_PER_CU")) if (atoi(e) > 0) t.fw_wgs_per_cu = atoi(e);
    if (const char* e = getenv("PDX_FLR_WAVE")) t.wave_force = e[0] != '0' ? 1 : 0;
    t.hybrid = !off("PDX_FLR_HYBRID");
    if (const char* e = getenv("PDX_FLR_MAX_RUN")) if (atoll(e) >= 64 && atoll(e) <= (1ll << 19)) t.max_run = (unsigned int)atoll(e);
    return t;
  }
};

constexpr unsigned int kFlrMaxRun = 1u << 19;  // a run is walked by ONE workgroup / wave (and sizes their counter levels): longer runs
                                               // (skewed keys) go through the side form below, or the whole column takes the classic path

// The side form of a fused layout: the rows of the (few) runs longer than t.max_run, gathered with their full slots, sorted by the top
// digit (one pass: within a digit they are in run order already, so the result is in slot order) and cut into segments for all G
// groups.  `keys` = the layout's top-digit bytes or 4-byte slots, `vals` its values, both sorted by the low slot bits.
template <typename KT>
static int build_side(pdx_groupby* gb, GroupedLayout& L, const KT* keys, const uint64_t* vals, const uint32_t* run_start, int64_t nruns, int low_bits,
                      unsigned int limit, int n_long, int64_t rows_long, bool nullable, Scratch& s, hipStream_t st) {
  PDX_PROFILE("side_layout", st);
  const int64_t G = gb->G, m = rows_long;
  uint32_t* unordered = s.get<uint32_t>(kMaxLongRuns);
  uint32_t* list = s.get<uint32_t>(kMaxLongRuns);
  uint32_t* off = s.get<uint32_t>(kMaxLongRuns + 1);
  unsigned int* cnt = s.get<unsigned int>(1);
  uint32_t* k_in = s.get<uint32_t>((size_t)m);
  uint64_t* v_in = s.get<uint64_t>((size_t)m);
  PDX_SCRATCH_CHECK(s);
  uint32_t* k_out = L.own<uint32_t>((size_t)m);
  uint64_t* v_out = L.own<uint64_t>((size_t)m);
  uint32_t* seg = L.own<uint32_t>((size_t)G + 1);
  if (!k_out || !v_out || !seg) return PDX_OOM;
  PDX_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned int), st));
  hipLaunchKernelGGL(k_long_runs_append, dim3(grid_for(nruns, 256)), dim3(256), 0, st, run_start, nruns, limit, (unsigned int)kMaxLongRuns, unordered, cnt);
  hipLaunchKernelGGL(k_long_runs_order, dim3(1), dim3(256), 0, st, unordered, n_long, run_start, list, off);
  hipLaunchKernelGGL((k_side_gather<KT>), dim3(grid_for(m, 256, 4)), dim3(256), 0, st, keys, vals, run_start, list, off, n_long, low_bits, m, k_in, v_in);
  PDX_LAUNCH_CHECK();
  const uint32_t* ks = nullptr;
  const uint64_t* vs = nullptr;
  PDX_TRY((radix_sort_pairs<uint64_t>(k_in, v_in, k_out, v_out, k_out, v_out, m, kFlrBits, &ks, &vs, true, s, st, low_bits)));
  hipLaunchKernelGGL(k_seg_starts, dim3(grid_for(G + 1, 256)), dim3(256), 0, st, ks, m, gb->occ_slot, G, seg);
  PDX_LAUNCH_CHECK();
  L.side_rows = m;
  L.side_runs = n_long;
  L.side_vals = vs;
  L.side_seg = seg;
  if (nullable) L.side_keys = ks;
  else L.disown(k_out);
  return PDX_OK;
}

// Which sort feeds the fused last digit on this handle (all false: full sort + classic reducers)
struct FusedPlan {
  bool flr = false, narrow = false, narrow_part = false;
  int low_bits = 0, eff_bits = 0, part = 0, mid_bits = 0;
  SortPlan low_plan{};
};
static FusedPlan plan_fused(const pdx_groupby* gb, bool nullable, const AggTuning& t) {
  FusedPlan p;
  const int64_t n = gb->n;
  p.part = gb->slot_part ? gb->part_bits : 0;
  // partitioned hash slots: the table itself is a power of two; only the two special slots (null key, INT64_MIN key) need one
  // more bit, so without them the top digit is drawn from the table's own bits (all 64 values used)
  p.eff_bits = (gb->slot_part && !gb->special_slots) ? gb->slot_bits - 1 : gb->slot_bits;
  p.low_bits = p.eff_bits - kFlrBits;
  p.mid_bits = p.low_bits - p.part;
  if (p.low_bits < 0) return p;
  // (hash-partitioned slots: only without the special slots -- with them the top digit is half empty and the runs half as long,
  //  measured slower than the classic path)
  p.flr = t.fused && n >= t.min_rows && p.low_bits - p.part >= 4 && p.low_bits >= t.min_low && p.low_bits <= 26 &&
          (!gb->slot_part || (t.fused_hash && !gb->special_slots)) && (n >> p.low_bits) >= t.min_run;
  if (p.flr && !gb->slot_part && gb->pass0_off)  // the stored pass-0 offsets belong to the first digit of the FULL plan
    p.flr = make_sort_plan(gb->slot_bits, sort_max_bits()).bits[0] == make_sort_plan(p.low_bits, sort_max_bits()).bits[0];
  if (!p.flr) return p;
  // Narrowing sort (dense slots, two passes below the fused digit): a digit that has been sorted on is dropped from the key, so
  // pass 0 writes 2-byte keys, pass 1 reads them and writes the top digit alone in a byte, which is all the fused kernel reads:
  // 12 B/row less traffic than carrying the 4-byte slot through.  Run starts then come from the scatter offsets (k_level_starts)
  // instead of a search in the sorted slots.  (values with nulls: the null flag rides in the narrow key's top bit, read from the
  // validity bitmap by pass 0 itself)
  p.low_plan = make_sort_plan(p.low_bits, sort_max_bits());
  p.narrow = t.narrow && !gb->slot_part && gb->pass0_off && p.low_plan.npasses == 2 && gb->slot_bits - p.low_plan.bits[0] <= (nullable ? 15 : 16) &&
             p.low_plan.bits[0] <= 8 && p.low_plan.bits[1] <= 8 && p.eff_bits == gb->slot_bits;
  // The same for the hash-partitioned layout (LDS build, one level, values without nulls): the value partition by bucket plays
  // pass 0, the build left every row's 13-bit index inside its bucket's region as a 2-byte key (idx16_part), so the one sort pass
  // below the fused digit reads 2-byte keys and writes the top digit alone: 8 B/row less than carrying the 4-byte logical slot.
  p.narrow_part = t.narrow && gb->slot_part && gb->idx16_part && !gb->digit2 && !nullable && p.part == kPartBits && p.mid_bits >= 4 && p.mid_bits <= 8 &&
                  p.eff_bits - p.part <= 16;
  return p;
}

static const char* slots_name(const pdx_groupby* gb) {
  if (gb->mode != 0) return gb->resample ? "bins" : "runs";
  if (gb->dense) return "dense";
  if (gb->slot_part) return gb->digit2 ? "hash_part2" : (gb->idx16_part ? "hash_lds" : "hash_part");
  return "hash_global";
}

// frees every block the layout took since `mark` except the ones named (a sort's ping-pong partners, partition copies)
static void keep_only(GroupedLayout& L, size_t mark, std::initializer_list<const void*> keep) {
  std::vector<void*> drop;
  for (size_t i = mark; i < L.owned.size(); ++i) {
    bool k = false;
    for (const void* q : keep) k = k || q == L.owned[i];
    if (!k) drop.push_back(L.owned[i]);
  }
  for (void* q : drop) L.disown(q);
}

// Stage 1 of pdx_groupby_agg for one column.  allow_fused: the request can be served by the fused last-digit kernels (the five
// standard kinds, variance, stddev); the layout then stops LOW of the top 6 slot bits when the handle qualifies (L.fused), unless a
// run turns out longer than kFlrMaxRun (skewed keys), in which case the sort is finished here (L.full).  Otherwise a full sort.
static int build_layout(pdx_groupby* gb, const pdx_column* values, bool allow_fused, const AggTuning& t, GroupedLayout& L, hipStream_t st) {
  const int64_t n = gb->n, G = gb->G;
  const uint8_t* vvalid = validity_or_null(values);
  L.values = values->values;
  L.validity = vvalid;
  L.offset = values->offset;
  L.dtype = values->dtype;
  L.stream = st;
  const std::string slots = std::string("slots=") + slots_name(gb);
  if (gb->mode != 0) {  // segments: the rows are grouped as they stand
    L.full = true;
    L.vals_sorted = static_cast<const uint64_t*>(values->values) + values->offset;
    L.seg_start = gb->seg_start;
    L.row_valid = vvalid;
    L.plan_full = slots + " sort=none layout=full";
    return PDX_OK;
  }
  Scratch s;
  const uint64_t* vin = static_cast<const uint64_t*>(values->values) + values->offset;
  const FusedPlan fp = allow_fused && !L.fused ? plan_fused(gb, vvalid != nullptr, t) : FusedPlan{};
  const int low_bits = fp.low_bits;
  const uint32_t* keys_sorted = nullptr;
  const uint64_t* vs = nullptr;
  bool sorted_done = false;  // narrowing sort, skewed keys: full layout built in the fused branch
  bool partial_lsd = false;  // 4-byte slots sorted by the low bits only; a skewed input finishes with one more pass
  std::string sort_desc;
  if (fp.flr) {
    const int64_t nruns = (int64_t)1 << low_bits;
    uint32_t* run_start = L.own<uint32_t>((size_t)nruns + 1);
    unsigned int* dmax = s.get<unsigned int>(3);  // longest run, runs over the limit, their rows
    if (!run_start) return PDX_OOM;
    PDX_SCRATCH_CHECK(s);
    unsigned int hmax = 0, n_long = 0, rows_long = 0;
    const unsigned int max_run = std::min(t.max_run, kFlrMaxRun);
    // a few long runs (one key with a large share of the rows is enough to make one) are reduced from a side form, the rest by the
    // fused kernels; many or very long ones: the whole column takes the classic path
    auto side_ok = [&] { return t.hybrid && n_long > 0 && n_long <= (unsigned int)kMaxLongRuns && (int64_t)rows_long <= n / 4; };
    const size_t mark = L.owned.size();
    if (fp.narrow || fp.narrow_part) {
      const int b0 = fp.narrow ? fp.low_plan.bits[0] : kPartBits, b1 = fp.narrow ? fp.low_plan.bits[1] : fp.mid_bits;
      sort_desc = std::string(fp.narrow ? "narrow:" : "narrow_part:") + std::to_string(b0) + "+" + std::to_string(b1);
      const int64_t ntiles = ceil_div(n, kSortTile), nchunks = ceil_div(ntiles, kColChunk);
      uint16_t* k16 = fp.narrow ? s.get<uint16_t>((size_t)n) : gb->idx16_part;
      const uint32_t* prev_off = fp.narrow ? gb->pass0_off : gb->part_off;  // (row 0 = where every first digit's rows begin in the input of pass 1)
      uint8_t* k8 = L.own<uint8_t>((size_t)n);
      uint64_t* nv0 = L.own<uint64_t>((size_t)n);
      uint64_t* nv1 = L.own<uint64_t>((size_t)n);
      uint32_t* nhist = s.get<uint32_t>((size_t)ntiles << 8);
      uint32_t* nchunk = s.get<uint32_t>((size_t)(nchunks + 1) << 8);
      if (!k8 || !nv0 || !nv1) return PDX_OOM;
      PDX_SCRATCH_CHECK(s);
      int rcn = PDX_OK;
#define NARROW_P0(B)                                                                                                                         \
  rcn = vvalid ? radix_scatter_narrow<B, uint64_t, uint32_t, uint16_t, true>(gb->slot_of_row, vin, k16, nv0, n, gb->pass0_off, st, vvalid, values->offset) \
               : radix_scatter_narrow<B, uint64_t, uint32_t, uint16_t>(gb->slot_of_row, vin, k16, nv0, n, gb->pass0_off, st)
      if (fp.narrow_part) rcn = radix_scatter_only<kPartBits, uint64_t, uint8_t>(gb->bucket8, vin, nullptr, nv0, n, 0, false, gb->part_off, st);
      else
        switch (b0) {
          case 4: NARROW_P0(4); break;
          case 5: NARROW_P0(5); break;
          case 6: NARROW_P0(6); break;
          case 7: NARROW_P0(7); break;
          default: NARROW_P0(8); break;
        }
#undef NARROW_P0
      PDX_TRY(rcn);
      // pass 1 in two halves: its offsets first -- the run starts and the longest run (the host needs that number to choose the
      // reducer) follow from them alone -- then the scatter, which runs while the host reads the number back
#define NARROW_P1_OFFSETS(B) rcn = radix_offsets<B, uint16_t>(k16, n, 0, nhist, nchunk, true, st)
      switch (b1) {
        case 4: NARROW_P1_OFFSETS(4); break;
        case 5: NARROW_P1_OFFSETS(5); break;
        case 6: NARROW_P1_OFFSETS(6); break;
        case 7: NARROW_P1_OFFSETS(7); break;
        default: NARROW_P1_OFFSETS(8); break;
      }
#undef NARROW_P1_OFFSETS
      PDX_TRY(rcn);
      struct EventGuard {  // the event must not outlive an early return
        hipEvent_t ev = nullptr;
        ~EventGuard() { if (ev) (void)hipEventDestroy(ev); }
      } hmax_ready;
      {
        PDX_PROFILE("run_starts", st);
        // (row 0 of the pass-0 offsets = where every first digit's rows begin in the input of pass 1)
        hipLaunchKernelGGL((k_level_starts<uint16_t>), dim3(1u << b0), dim3(256), 0, st, k16, n, prev_off, (int64_t)1 << b0, b0, b1, nhist, run_start);
        PDX_HIP(hipMemsetAsync(dmax, 0, 3 * sizeof(unsigned int), st));
        hipLaunchKernelGGL(k_run_max_len, dim3(grid_for(nruns, 256)), dim3(256), 0, st, run_start, nruns, dmax, max_run);
        PDX_LAUNCH_CHECK();
        PDX_HIP(hipMemcpyAsync(hmax_pinned(), dmax, 3 * sizeof(unsigned int), hipMemcpyDeviceToHost, st));
        PDX_HIP(hipEventCreateWithFlags(&hmax_ready.ev, hipEventDisableTiming));
        PDX_HIP(hipEventRecord(hmax_ready.ev, st));
      }
#define NARROW_P1(B)                                                                    \
    rcn = vvalid ? radix_scatter_narrow<B, uint64_t, uint16_t, uint8_t, true>(k16, nv0, k8, nv1, n, nhist, st) \
                 : radix_scatter_narrow<B, uint64_t, uint16_t, uint8_t>(k16, nv0, k8, nv1, n, nhist, st)
      switch (b1) {
        case 4: NARROW_P1(4); break;
        case 5: NARROW_P1(5); break;
        case 6: NARROW_P1(6); break;
        case 7: NARROW_P1(7); break;
        default: NARROW_P1(8); break;
      }
#undef NARROW_P1
      PDX_TRY(rcn);
      {
        const hipError_t ew = hipEventSynchronize(hmax_ready.ev);
        if (ew != hipSuccess) return hip_fail(ew, "pdx_groupby_agg");
        hmax = hmax_pinned()[0];
        n_long = hmax_pinned()[1];
        rows_long = hmax_pinned()[2];
      }
      if (hmax > max_run && side_ok()) {
        PDX_TRY((build_side<uint8_t>(gb, L, k8, nv1, run_start, nruns, low_bits, max_run, (int)n_long, (int64_t)rows_long, vvalid != nullptr, s, st)));
        L.fused = true;
        L.keys8 = k8;
        L.fvals = nv1;
        L.disown(nv0);
      } else if (hmax > max_run) {
        // skewed keys: the fused kernel is skipped.  Finish the sort with the one pass that is left (on the byte digits) and take the
        // group offsets from its scatter offsets: one more level of k_level_starts gives the start of every slot's rows
        uint32_t* ss_narrow = L.own<uint32_t>((size_t)G + 1);
        uint32_t* slot_start = s.get<uint32_t>(((size_t)nruns << kFlrBits) + 1);
        if (!ss_narrow) return PDX_OOM;
        PDX_SCRATCH_CHECK(s);
        PDX_TRY((radix_offsets<kFlrBits, uint8_t>(k8, n, 0, nhist, nchunk, true, st)));
        const uint32_t* fkeys = nullptr;
        if (vvalid) {  // the classic nullable reducers read the null flag from bit 31 of a 4-byte key per grouped row: write flags only
          uint32_t* fk = L.own<uint32_t>((size_t)n);
          if (!fk) return PDX_OOM;
          PDX_TRY((radix_scatter_narrow<kFlrBits, uint64_t, uint8_t, uint32_t, true>(k8, nv1, fk, nv0, n, nhist, st)));
          fkeys = fk;
        } else {
          PDX_TRY((radix_scatter_narrow<kFlrBits, uint64_t, uint8_t, uint8_t>(k8, nv1, (uint8_t*)nullptr, nv0, n, nhist, st)));
        }
        {
          PDX_PROFILE("seg_starts", st);
          hipLaunchKernelGGL((k_level_starts<uint8_t>), dim3((unsigned)std::min<int64_t>(nruns, 65536)), dim3(256), 0, st, k8, n, run_start, nruns, low_bits,
                             kFlrBits, nhist, slot_start);
          hipLaunchKernelGGL(k_seg_starts_from_slots, dim3(grid_for(G + 1, 256)), dim3(256), 0, st, slot_start, n, gb->occ_slot, G, ss_narrow);
        }
        PDX_LAUNCH_CHECK();
        L.full = true;
        L.vals_sorted = nv0;
        L.flag_keys = fkeys;
        L.seg_start = ss_narrow;
        L.out_index = gb->gid_of_occ;
        sorted_done = true;
        keep_only(L, mark, {nv0, ss_narrow, fkeys});
        L.disown(run_start);
      } else {
        L.fused = true;
        L.keys8 = k8;
        L.fvals = nv1;
        L.disown(nv0);
      }
    } else {
      sort_desc = "lsd:skip" + std::to_string(gb->slot_bits - low_bits);
      PDX_TRY(sort_values_by_slot(gb, vin, vvalid, values->offset, [&](size_t bytes) { return (void*)L.own<uint8_t>(bytes); }, s, st, &keys_sorted, &vs,
                                  gb->slot_bits - low_bits));
      keep_only(L, mark, {keys_sorted, vs});
      {
        PDX_PROFILE("run_starts", st);
        PDX_HIP(hipMemsetAsync(dmax, 0, 3 * sizeof(unsigned int), st));
        hipLaunchKernelGGL(k_run_starts, dim3(grid_for(nruns + 1, 256)), dim3(256), 0, st, keys_sorted, n, low_bits, nruns, run_start, dmax);
        hipLaunchKernelGGL(k_run_max_len, dim3(grid_for(nruns, 256)), dim3(256), 0, st, run_start, nruns, dmax, max_run);
        PDX_LAUNCH_CHECK();
        unsigned int h3[3] = {0, 0, 0};
        PDX_HIP(hipMemcpyAsync(h3, dmax, sizeof(h3), hipMemcpyDeviceToHost, st));
        PDX_HIP(hipStreamSynchronize(st));
        hmax = h3[0];
        n_long = h3[1];
        rows_long = h3[2];
      }
      if (hmax > max_run && side_ok())
        PDX_TRY((build_side<uint32_t>(gb, L, keys_sorted, vs, run_start, nruns, low_bits, max_run, (int)n_long, (int64_t)rows_long, vvalid != nullptr, s, st)));
      if (hmax <= max_run || L.side_rows) {
        L.fused = true;
        L.fkeys = keys_sorted;
        L.fvals = vs;
      } else {
        partial_lsd = true;
        L.disown(run_start);
      }
    }
    if (L.fused) {
      L.low_bits = low_bits;
      L.nruns = nruns;
      L.hmax = std::min(hmax, max_run);  // (rows of the longest run the fused kernels walk)
      L.max_run = max_run;
      L.run_start = run_start;
      L.plan_fused = slots + " sort=" + sort_desc + " layout=fused" + (L.side_rows ? " side=" + std::to_string(L.side_runs) : std::string());
      return PDX_OK;
    }
  }
  // ---- full form: stable sort of (slot, value) by slot, each group's values contiguous in row order
  if (!sorted_done) {
    const size_t mark = L.owned.size();
    if (partial_lsd) {
      // the fused kernel was skipped (a run longer than 2^19 rows: skewed keys): finish the sort with the one pass that is left
      uint32_t* k2 = L.own<uint32_t>((size_t)n);
      uint64_t* v2 = L.own<uint64_t>((size_t)n);
      if (!k2 || !v2) return PDX_OOM;
      const uint32_t* ks2 = nullptr;
      const uint64_t* vs2 = nullptr;
      PDX_TRY((radix_sort_pairs<uint64_t>(keys_sorted, vs, k2, v2, k2, v2, n, kFlrBits, &ks2, &vs2, true, s, st, low_bits)));
      L.disown(keys_sorted);
      L.disown(vs);
      keys_sorted = ks2;
      vs = vs2;
      sort_desc += "+finish";
    } else if (G == 1 && gb->slot_of_row) {
      // one group: the rows are grouped as they stand (no sort); null flags, if any, still go into a key per row
      keys_sorted = gb->slot_of_row;
      if (vvalid) {
        uint32_t* fk1 = L.own<uint32_t>((size_t)n);
        if (!fk1) return PDX_OOM;
        hipLaunchKernelGGL(k_flag_keys, dim3(grid_for(n, 256, 4)), dim3(256), 0, st, gb->slot_of_row, vvalid, values->offset, n, fk1);
        keys_sorted = fk1;
      }
      vs = vin;
      sort_desc = "none";
    } else {
      PDX_TRY(sort_values_by_slot(gb, vin, vvalid, values->offset, [&](size_t bytes) { return (void*)L.own<uint8_t>(bytes); }, s, st, &keys_sorted, &vs));
      keep_only(L, mark, {keys_sorted, vs});
      sort_desc = "lsd";
    }
    uint32_t* ss = L.own<uint32_t>((size_t)G + 1);
    if (!ss) return PDX_OOM;
    {
      PDX_PROFILE("seg_starts", st);
      hipLaunchKernelGGL(k_seg_starts, dim3(grid_for(G + 1, 256)), dim3(256), 0, st, keys_sorted, n, gb->occ_slot, G, ss);
    }
    PDX_LAUNCH_CHECK();
    L.full = true;
    L.vals_sorted = vs;
    L.flag_keys = vvalid ? keys_sorted : nullptr;
    if (!vvalid && keys_sorted != gb->slot_of_row) L.disown(keys_sorted);  // the sorted slots were only needed for the group offsets
    L.seg_start = ss;
    L.out_index = gb->gid_of_occ;
  }
  L.plan_full = slots + " sort=" + sort_desc + " layout=full" + (fp.flr ? " skew=1" : "");
  return PDX_OK;
}
