// groupby.hip -- hash group-by with sum/mean/min/max/count, and the resample front-end, for gfx950.
//
// Replaces (reference file:line)
//   GroupBy::makeGroups  src/dataframe.cpp:1571-1600   Grouper::Make/Consume/GetUniques  -> pdx_groupby_create
//   processIndex/processEach (MakeGroupings + ApplyGroupings of every column, 1539-1569)  -> deferred, per aggregated
//                                                                                            column, to pdx_groupby_agg
//   GROUPBY_AGG / GROUPBY_NUMERIC_AGG  src/pd_core_macros.h:5-147 (one CallFunction per group)  -> pdx_groupby_agg
//   GroupBy::group / MakeSubDataFrame / apply (src/group_by.h:38-77: walk the per-group arrays)  -> pdx_groupby_groupings
//   pd::resample / makeGroupInfo / generate_bins_dt64 / GroupInfo::downsample
//        src/resample.h:19-43,91-122  src/resample.cpp:11-83,85-178,202-295               -> pdx_resample_create
//
// Data path for N rows, G groups (all arrays in HBM; DESIGN.md section 3 has the kernel-by-kernel accounting):
//   1. key -> slot.  Dense integer key domain (span < min(2^26, 4N)): slot = key - min, or its residue form key & (2^b - 1) built
//      in ONE speculative pass together with the exact min/max, first rows through an LDS-resident "seen" bitmap
//      (k_dense_slots*, k_sample_key_range).  General keys: hash -> stable partition of (key, row id) by the low 8 hash bits
//      (k_hash_bucket_hist + radix scatter; a second level for very many groups) -> one workgroup per bucket builds the
//      bucket's table region in LDS (k_hash_probe_lds; skewed buckets: head rows there, the rest in chunks against an LDS
//      snapshot, k_hash_probe_lds_tail; k_hash_probe_part is the memory-side fallback).  Tiny inputs: one global
//      open-addressing table (k_hash_insert).
//   2. occupied slots are compacted (slot order) and sorted by first_row -> dense gid in FIRST-OCCURRENCE order.
//   3. per aggregated column: stable LSD radix sort of (slot, value) by slot (radix_sort.hpp) -> every group's values
//      contiguous IN ROW ORDER.  For sum/mean/min/max/count (and variance) the last 6 slot bits are not sorted: k_flr_reduce
//      ranks each 2560-row tile by them in LDS, stages the rows by 16-value leaf, sums one leaf per thread and pushes the leaves
//      through Arrow's binary counter with one lane per group (k_flr_wave: one wave per run, for nullable values / min / max).
//      Dense slots + values without nulls: narrowing sort (the key shrinks 4 -> 2 -> 1 byte as digits are consumed; run and
//      group starts come from the scatter offsets, k_level_starts).  Runs longer than 2^19 rows (hot keys) are skipped by the
//      fused kernels and reduced from a side form of the layout (build_side, gb_layout.hpp).
//   4. classic reducers on fully sorted values (skewed keys, small inputs, resample, product/first/last): k_seg_reduce (one wave
//      per group, 16-value leaves + shuffle tree + counter), k_seg_reduce_mid (batches of short groups per wave),
//      k_seg_reduce_sub + k_seg_combine_big (many waves per long group), k_seg_reduce_nullable.
// All fp64 sums reproduce Arrow's pairwise summation bit for bit.  Algorithmic bytes: 16 B/row (8 key + 8 value).
#include <stdlib.h>
#include <string.h>
#include <string>
#include <algorithm>
#include <memory>
#include <vector>
#include "compact.hpp"
#include "minmax.hpp"
#include "pairwise.hpp"
#include "radix_sort.hpp"
#include "scan.hpp"
#include "temporal_round.hpp"

namespace pdx {
#include "gb_hash_build.hpp"
#include "gb_dense_slots.hpp"
#include "gb_group_ids.hpp"
#include "gb_seg_reduce.hpp"
#include "gb_resample_kernels.hpp"

}  // namespace pdx

using namespace pdx;

#include "gb_handle.hpp"

// the 4-byte logical slots of a hash-partitioned handle, rebuilt from the 2-byte region indexes the LDS build wrote (on demand)
static int ensure_slot_part(pdx_groupby* gb, hipStream_t st) {
  if (!gb->slot_part || gb->slot_part_ready) return PDX_OK;
  hipLaunchKernelGGL(k_slot_part_from_idx16, dim3(grid_for(gb->n, 256, 8)), dim3(256), 0, st, gb->idx16_part, gb->part_off, gb->n, gb->slot_part);
  PDX_LAUNCH_CHECK();
  gb->slot_part_ready = true;
  return PDX_OK;
}

// the original row of every partitioned row (no null keys: the hash build's partition moves the keys alone and finds the groups' first rows
// from positions; the few callers that want every row's origin -- group ids per row, validity flags of nullable values, the memory-side /
// split / skewed-bucket forms of the build -- replay the partition with the row number as the payload)
static int ensure_rows_part(pdx_groupby* gb, hipStream_t st) {
  if (gb->rows_part || !gb->bucket8) return PDX_OK;
  gb->rows_part = gb->own<uint32_t>((size_t)gb->n);
  if (!gb->rows_part) return PDX_OOM;
  return radix_scatter_iota<kPartBits, uint8_t>(gb->bucket8, nullptr, gb->rows_part, gb->n, 0, false, gb->part_off, nullptr, 0, st);
}

namespace pdx {
#include "gb_sort_values.hpp"
#include "gb_flr_reduce.hpp"
#include "gb_more_aggs.hpp"

}  // namespace pdx

// Groups = runs of equal labels over the rows as they stand (segments mode).  *done = false when the labels descend somewhere in
// both integer orders (the caller then builds a dictionary); otherwise the handle is complete.
template <typename LabelFn>
static int build_label_runs(pdx_groupby* gb, int64_t n, LabelFn fn, long long shift, Scratch& s, hipStream_t st, bool* done) {
  *done = false;
  const int64_t nblocks = ceil_div(n, (int64_t)kRunTile), nwaves = ceil_div(n, (int64_t)kRunWaveRows);
  unsigned int* flags = s.get<unsigned int>(2);
  unsigned long long* marks = s.get<unsigned long long>((size_t)((n + 63) >> 6));
  int64_t* offsets = s.get<int64_t>((size_t)nwaves);  // run starts in front of every 1024-row slice
  int64_t* total = s.get<int64_t>(1);
  PDX_SCRATCH_CHECK(s);
  unsigned int hflags[2] = {0, 0};
  int64_t G = 0;
  {
    PDX_PROFILE("label_runs", st);
    PDX_HIP(hipMemsetAsync(flags, 0, sizeof(hflags), st));
    hipLaunchKernelGGL((k_label_run_count<LabelFn, LabelFn::kBatch>), dim3((unsigned)nblocks), dim3(kRunBlock), 0, st, n, fn, marks, offsets, flags);
    PDX_TRY((device_exclusive_scan<int64_t, SumOp>(offsets, offsets, nwaves, total, s, st)));
    PDX_LAUNCH_CHECK();
  }
  PDX_HIP(hipMemcpyAsync(hflags, flags, sizeof(hflags), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipMemcpyAsync(&G, total, sizeof(G), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
  if (hflags[0] && hflags[1]) return PDX_OK;
  gb->mode = 1;
  gb->G = G;
  gb->seg_start = gb->own<uint32_t>((size_t)G + 1);
  gb->uniques = gb->own<int64_t>((size_t)G);
  gb->first_rows = gb->own<int64_t>((size_t)G);
  gb->unique_ok = gb->own<uint8_t>((size_t)G);
  gb->gid_of_occ = gb->own<uint32_t>((size_t)G);
  if (!gb->seg_start || !gb->uniques || !gb->first_rows || !gb->unique_ok || !gb->gid_of_occ) return PDX_OOM;
  {
    PDX_PROFILE("label_runs", st);
    hipLaunchKernelGGL((k_label_run_write<RunEmit<LabelFn>>), dim3((unsigned)nblocks), dim3(kRunBlock), 0, st, n, marks,
                       RunEmit<LabelFn>{fn, shift, gb->seg_start, gb->uniques, gb->first_rows, gb->gid_of_occ}, offsets, nwaves, total);
    hipLaunchKernelGGL(k_set_last, dim3(1), dim3(64), 0, st, gb->seg_start, G, (uint32_t)n);
    PDX_HIP(hipMemsetAsync(gb->unique_ok, 1, (size_t)G, st));
    PDX_LAUNCH_CHECK();
  }
  PDX_HIP(hipStreamSynchronize(st));
  *done = true;
  return PDX_OK;
}

template <int MODE, bool CEIL>
struct RoundLabel {
  static constexpr int kBatch = 4;
  const long long* ts;
  RoundParams q;
  __device__ long long operator()(int64_t i) const { return round_one<MODE, CEIL>(ts[i], q); }
};
__global__ void k_shift_labels(int64_t* __restrict__ labels, int64_t G, long long shift) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < G; i += stride)
    labels[i] = (long long)((unsigned long long)labels[i] + (unsigned long long)shift);
}

// one pinned word per host thread: the target of small device-to-host reads that must not block the launches queued behind them
// (a copy into pageable memory is staged and waits; pinned memory makes hipMemcpyAsync + an event a real overlap)
static unsigned int* hmax_pinned() {
  static thread_local unsigned int* p = nullptr;
  if (!p) {
    void* q = nullptr;
    if (hipHostMalloc(&q, 64, hipHostMallocPortable) != hipSuccess) {
      (void)hipGetLastError();
      q = new unsigned int[16];
    }
    p = static_cast<unsigned int*>(q);
  }
  return p;
}

namespace pdx {
#include "gb_layout.hpp"
#include "gb_acc.hpp"
}  // namespace pdx

extern "C" {

int pdx_groupby_create(const pdx_column* key, void* stream, pdx_groupby** out) {
  PDX_TRY(check_column(key, "pdx_groupby_create"));
  if (!out) return fail(PDX_INVALID, "pdx_groupby_create: null output");
  if (!is_int_like(key->dtype)) return fail(PDX_NOT_IMPLEMENTED, "pdx_groupby_create: key must be int64/uint64/timestamp");
  const int64_t n = key->length;
  if (n > 0x7FFFFFFFll) return fail(PDX_NOT_IMPLEMENTED, "pdx_groupby_create: more than 2^31-1 rows per call is not supported yet");
  hipStream_t st = as_stream(stream);
  std::unique_ptr<pdx_groupby> owner(new pdx_groupby());  // released into *out on success
  owner->stream = st;
  pdx_groupby* gb = owner.get();
  gb->n = n;
  gb->key_dtype = key->dtype;
  *out = nullptr;
  if (n == 0) {
    *out = owner.release();
    return PDX_OK;
  }
  Scratch s;
  const long long* keys = static_cast<const long long*>(key->values) + key->offset;
  const uint8_t* valid = validity_or_null(key);
  HashCtl* ctl = s.get<HashCtl>(1);
  if (s.failed) return PDX_OOM;
  // ---- keys that arrive grouped (non-decreasing, no nulls: a sorted id column, a time index rounded to calendar bins): the groups
  // are the runs of equal keys where they stand -- no dictionary, and no value sort in pdx_groupby_agg (segments mode, as resample).
  // 65536 sampled neighbour pairs rule the path out for anything else at the price of one small kernel; the counting pass of the
  // run-start compaction then proves the order over all rows (a sample that lied costs that one pass: the build below takes over).
  const bool sorted_env = [] { const char* e = getenv("PDX_GROUPBY_SORTED"); return !(e && e[0] == '0'); }();
  const char* denv = getenv("PDX_GROUPBY_DENSE");
  const bool allow_dense = !(denv && denv[0] == '0');
  const bool lds_ok = [] { const char* e = getenv("PDX_DENSE_LDS"); return !(e && e[0] == '0'); }();
  const bool spec_ok = [] { const char* e = getenv("PDX_DENSE_SPECULATE"); return !(e && e[0] == '0'); }();
  const bool want_sorted_sample = sorted_env && n >= 2 && !valid;
  const bool want_range_sample = allow_dense && spec_ok && lds_ok && n >= ((int64_t)1 << 23);  // (the speculative dense build below)
  KeyRange hsv[64];
  unsigned int hflags[4] = {0, 0, 0, 4u};
  unsigned int* flags = nullptr;
  if (want_sorted_sample || want_range_sample) {  // both samples, one host round trip
    flags = s.get<unsigned int>(4);
    KeyRange* dsample = s.get<KeyRange>(64);
    if (s.failed) return PDX_OOM;
    if (want_sorted_sample) {
      PDX_HIP(hipMemsetAsync(flags, 0, 4 * sizeof(unsigned int), st));
      hipLaunchKernelGGL(k_sample_descents, dim3(64), dim3(1024), 0, st, keys, valid, key->offset, n, flags + 3);
      PDX_HIP(hipMemcpyAsync(hflags, flags, sizeof(hflags), hipMemcpyDeviceToHost, st));
    }
    if (want_range_sample) {
      hipLaunchKernelGGL(k_sample_key_range, dim3(64), dim3(1024), 0, st, keys, valid, key->offset, n, dsample);
      PDX_HIP(hipMemcpyAsync(hsv, dsample, sizeof(hsv), hipMemcpyDeviceToHost, st));
    }
    PDX_LAUNCH_CHECK();
    PDX_HIP(hipStreamSynchronize(st));
  }
  if (want_sorted_sample && !(hflags[3] & 4u) && (hflags[3] & 3u) != 3u) {
    bool done = false;
    PDX_TRY(build_label_runs(gb, n, KeyLabel{keys}, 0, s, st, &done));
    if (done) {
      *out = owner.release();
      return PDX_OK;
    }
  }
  // ---- dense-domain fast path: valid keys span a small integer range -> slot = key - min (or its residue form), no table
  Slot* table = nullptr;
  unsigned int* dense_first = nullptr;
  long long dense_min = 0;
  unsigned int dense_mask = 0;
  unsigned int null_slot = 0;
  int64_t nslots = 0;
  const unsigned long long dense_lim = std::min<unsigned long long>(1ull << 26, (unsigned long long)n * 4 + 1024);

  // Builds slot_of_row, first[] and (when the first sort digit is 4..8 bits wide) the scanned pass-0 offsets for the dense domain
  // described by (mn, mask, null_slot, nslots).  range_out != nullptr: also the exact key range, computed by the same pass.
  auto dense_build = [&](long long mn, unsigned int mask, KeyRange* range_out) -> int {
    dense_first = s.get<unsigned int>((size_t)nslots);
    if (s.failed) return PDX_OOM;
    hipMemsetAsync(dense_first, 0xFF, (size_t)nslots * sizeof(unsigned int), st);
    // prefix with the full first-row protocol, then the bitmap-filtered tail; the tail kernel is tile shaped and also emits the
    // per-tile histogram of the first sort digit
    const int64_t ntiles = ceil_div(n, kSortTile), nchunks = ceil_div(ntiles, kColChunk);
    const int bits0 = make_sort_plan(ilog2((uint64_t)nslots), sort_max_bits()).bits[0];
    // rows that go through the full first-row protocol before the tail's bitmap is taken: a slot unseen by then costs one global atomic per
    // later row of its key.  16 rows per slot leave e^-16 of uniformly spread keys unseen; below 2^29 rows 8 per slot (e^-8 = 3e-4 of the
    // slots, a few 10^4 atomics) -- the prefix pass is the slow one (60 Grows/s: a random read of first[] per row) and at an 8-GPU shard's
    // size 16 per slot was 13 % of the rows, 0.28 ms of a 0.84 ms create
    const int64_t per_slot = [n] { const char* e = getenv("PDX_DENSE_PREFIX_PER_SLOT"); return e && atoi(e) > 0 ? (int64_t)atoi(e) : (n < ((int64_t)1 << 29) ? 8 : 16); }();
    int64_t prefix = std::min<int64_t>(n, std::max<int64_t>((int64_t)1 << 22, per_slot * nslots));
    prefix = std::min<int64_t>(n, round_up(prefix, kSortTile));
    const bool fuse = bits0 >= 4 && bits0 <= 8;
    const int64_t nwords = (nslots + 31) >> 5;
    const int64_t tile_first_tail = prefix / kSortTile;
    const bool lds_bitmap = fuse && lds_ok && prefix < n && nwords <= kDenseLdsWords && ntiles - tile_first_tail >= 256;
    if (range_out && !(fuse && prefix < n)) return fail(PDX_DEVICE, "dense_build: fused key range needs the histogram tail");
    uint32_t* chunk_sum = nullptr;
    if (fuse) {
      gb->pass0_off = gb->own<uint32_t>((size_t)ntiles << bits0);
      chunk_sum = s.get<uint32_t>((size_t)(nchunks + 1) << bits0);  // + digit totals row
      if (!gb->pass0_off || s.failed) return PDX_OOM;
    }
    KeyRange* wg_range = nullptr;
    long long *tile_min = nullptr, *tile_max = nullptr;
    if (range_out) {
      wg_range = s.get<KeyRange>((size_t)kCUs);
      if (s.failed) return PDX_OOM;
      if (!lds_bitmap) {  // the global-bitmap tail writes one (min, max) per tile
        tile_min = s.get<long long>((size_t)ntiles);
        tile_max = s.get<long long>((size_t)ntiles);
        if (s.failed) return PDX_OOM;
      }
    }
    {
      PDX_PROFILE("dense_slots", st);
      hipLaunchKernelGGL(k_dense_slots, dim3(grid_for(prefix, 256, 8)), dim3(256), 0, st, keys, valid, key->offset, prefix, mn, mask, null_slot,
                         dense_first, gb->slot_of_row);
      if (prefix < n) {
        uint32_t* seen = s.get<uint32_t>((size_t)nwords);
        if (s.failed) return PDX_OOM;
        hipLaunchKernelGGL(k_seen_bitmap, dim3(grid_for(nwords, 256)), dim3(256), 0, st, dense_first, nslots, seen);
        const unsigned tail_tiles = (unsigned)(ntiles - tile_first_tail);
        // the LDS tail walks ALL tiles (slots, histogram and key range of the prefix rows too; first-row tracking from `prefix` on)
#define TAIL_HIST(B)                                                                                                                              \
  if (lds_bitmap)                                                                                                                                 \
    hipLaunchKernelGGL((k_dense_slots_tail_hist_lds<B>), dim3(kCUs), dim3(kDenseLdsBlock), 0, st, keys, valid, key->offset, (int64_t)0, ntiles, n,  \
                       mn, mask, null_slot, seen, (int)nwords, prefix, dense_first, gb->slot_of_row, gb->pass0_off, wg_range);                    \
  else if (range_out) /* all tiles: slots, histogram and key range of the prefix rows too */                                                     \
    hipLaunchKernelGGL((k_dense_slots_tail_hist<B>), dim3((unsigned)ntiles), dim3(kSortBlock), 0, st, keys, valid, key->offset, (int64_t)0, n, mn,  \
                       mask, null_slot, seen, prefix, dense_first, gb->slot_of_row, gb->pass0_off, tile_min, tile_max);                           \
  else                                                                                                                                            \
    hipLaunchKernelGGL((k_dense_slots_tail_hist<B>), dim3(tail_tiles), dim3(kSortBlock), 0, st, keys, valid, key->offset, tile_first_tail, n, mn,  \
                       mask, null_slot, seen, prefix, dense_first, gb->slot_of_row, gb->pass0_off, (long long*)nullptr, (long long*)nullptr)
        if (!fuse)
          hipLaunchKernelGGL(k_dense_slots_tail, dim3(grid_for(n - prefix, 256, 8)), dim3(256), 0, st, keys, valid, key->offset, prefix, n, mn, mask,
                             null_slot, seen, dense_first, gb->slot_of_row);
        else if (bits0 == 4) TAIL_HIST(4);
        else if (bits0 == 5) TAIL_HIST(5);
        else if (bits0 == 6) TAIL_HIST(6);
        else if (bits0 == 7) TAIL_HIST(7);
        else TAIL_HIST(8);
#undef TAIL_HIST
      }
    }
    PDX_LAUNCH_CHECK();
    if (fuse) {
      if (!lds_bitmap && !range_out) {
        // histogram of the prefix tiles (the prefix is a whole number of tiles unless it is the whole input)
#define PREFIX_HIST(B) hipLaunchKernelGGL((k_radix_hist<B>), dim3((unsigned)ceil_div(prefix, kSortTile)), dim3(kSortBlock), 0, st, gb->slot_of_row, prefix, 0, \
                                          gb->pass0_off)
        if (bits0 == 4) PREFIX_HIST(4);
        else if (bits0 == 5) PREFIX_HIST(5);
        else if (bits0 == 6) PREFIX_HIST(6);
        else if (bits0 == 7) PREFIX_HIST(7);
        else PREFIX_HIST(8);
#undef PREFIX_HIST
      }
      int rcs = radix_scan_dispatch(bits0, gb->pass0_off, ntiles, chunk_sum, true, st);
      if (rcs != PDX_OK) return rcs;
    }
    if (range_out && !lds_bitmap) {
      long long lo = 0, hi = 0, dummy = 0;
      PDX_TRY(minmax_i64_host(tile_min, ntiles, &lo, &dummy, s, st));
      PDX_TRY(minmax_i64_host(tile_max, ntiles, &dummy, &hi, s, st));
      *range_out = KeyRange{lo, hi, lo <= hi ? 1 : 0, 0};
    } else if (range_out) {
      std::vector<KeyRange> h((size_t)kCUs);
      PDX_HIP(hipMemcpyAsync(h.data(), wg_range, sizeof(KeyRange) * kCUs, hipMemcpyDeviceToHost, st));
      PDX_HIP(hipStreamSynchronize(st));
      KeyRange r{0x7FFFFFFFFFFFFFFFll, (long long)0x8000000000000000ull, 0, 0};
      for (size_t wi = 0; wi < h.size(); ++wi)
        if (h[wi].any) {
          r.vmin = std::min(r.vmin, h[wi].vmin);
          r.vmax = std::max(r.vmax, h[wi].vmax);
          r.any = 1;
        }
      *range_out = r;
    }
    return PDX_OK;
  };

  MinMaxPartial<long long> mm;
  bool have_mm = false;
  bool sample_rules_out_dense = !allow_dense;
  // ---- speculative single pass: guess the width of the key window from a sample, build the residue-form dense domain and the
  // exact min/max together (saves the separate 8 B/row min/max pass), accept when the exact span fits the guessed width
  if (want_range_sample) {
    KeyRange hs{0x7FFFFFFFFFFFFFFFll, (long long)0x8000000000000000ull, 0, 0};
    for (const KeyRange& w : hsv)
      if (w.any) {
        hs.vmin = std::min(hs.vmin, w.vmin);
        hs.vmax = std::max(hs.vmax, w.vmax);
        hs.any = 1;
      }
    int b = 64;
    if (hs.any) {
      const unsigned long long span_s = (unsigned long long)hs.vmax - (unsigned long long)hs.vmin;
      b = 4;
      while (b < 64 && (span_s >> b)) ++b;  // smallest width (>= 4 bits) with sample span < 2^b
      // the sample range lies inside the exact range: a sample already too wide for the dense domain settles the question
      // without the full min/max pass
      if (span_s >= dense_lim) sample_rules_out_dense = true;
    }
    // Only windows of <= 20 bits (LDS-resident bitmap).  Wider windows work too (global bitmap, tile min/max in the same pass) but
    // do not pay: the power-of-two residue domain makes the bitmap up to 2x larger than key - min and costs more than the
    // saved min/max pass (measured at 1e7 keys: 42.6 vs 39.7 ms).
    const int spec_max_bits = [] { const char* e = getenv("PDX_DENSE_SPECULATE_BITS"); return e ? atoi(e) : 20; }();
    if (b <= spec_max_bits && b <= 26 && (1ull << b) <= dense_lim && (int64_t)16 * (((int64_t)1 << b) + 1) + 2 * kSortTile < n) {
      dense_mask = (1u << b) - 1;
      null_slot = 1u << b;
      nslots = (int64_t)null_slot + (valid ? 1 : 0);
      gb->slot_of_row = gb->own<uint32_t>((size_t)n);
      if (!gb->slot_of_row) return PDX_OOM;
      KeyRange exact;
      PDX_TRY(dense_build(0, dense_mask, &exact));
      mm.vmin = exact.vmin;
      mm.vmax = exact.vmax;
      mm.rmin = mm.rmax = exact.any ? 0 : -1;
      have_mm = true;
      if (exact.any && (unsigned long long)exact.vmax - (unsigned long long)exact.vmin <= dense_mask) {
        gb->dense = 1;
        dense_min = exact.vmin;
      } else {
        // the window is wider than the sample suggested (or there is no valid key at all): redo on the exact range below
        dense_mask = 0;
        dense_first = nullptr;
        while (!gb->owned.empty()) {
          pool_free(gb->owned.back());
          gb->owned.pop_back();
        }
        gb->slot_of_row = nullptr;
        gb->pass0_off = nullptr;
      }
    }
  }
  if (!gb->dense && !sample_rules_out_dense) {
    if (!have_mm) {
      int rc0 = minmax_keys_host(keys, valid, key->offset, n, &mm, s, st);
      if (rc0 != PDX_OK) return rc0;
    }
    if (allow_dense && mm.rmin >= 0) {
      unsigned long long span = (unsigned long long)mm.vmax - (unsigned long long)mm.vmin;  // range - 1
      if (span < dense_lim) {
        gb->dense = 1;
        dense_min = mm.vmin;
        null_slot = (unsigned int)span + 1;
        nslots = (int64_t)null_slot + 1;
      }
    } else if (allow_dense && mm.rmin < 0) {  // every key is null: one group
      gb->dense = 1;
      null_slot = 0;
      nslots = 1;
    }
    if (gb->dense) {
      gb->slot_of_row = gb->own<uint32_t>((size_t)n);
      if (!gb->slot_of_row) return PDX_OOM;
      PDX_TRY(dense_build(dense_min, 0, nullptr));
    }
  }
  unsigned int region = 0;  // != 0: partitioned hash table (logical slot = (index in region << kPartBits) | bucket)
  const char* penv = getenv("PDX_HASH_PARTITION");
  const bool use_partition = !gb->dense && !(penv && penv[0] == '0') && (n >= ((int64_t)1 << 18) || (penv && penv[0] == '2'));
  if (!use_partition && !gb->dense) {
    gb->slot_of_row = gb->own<uint32_t>((size_t)n);
    if (!gb->slot_of_row) return PDX_OOM;
  }
  if (gb->dense) {
    // (built above)
  } else if (use_partition) {
    // ---- general keys, partitioned build: hash -> stable partition by the low 8 hash bits (== first LSD pass of the sort by slot)
    const int64_t ntiles = ceil_div(n, kSortTile), nchunks = ceil_div(ntiles, kColChunk);
    gb->bucket8 = gb->own<uint8_t>((size_t)n);
    gb->part_off = gb->own<uint32_t>((size_t)ntiles << kPartBits);
    gb->slot_part = gb->own<uint32_t>((size_t)n);
    // row ids ride through the partition only when there are null keys (their flag is bit 31 of the row id); otherwise on demand
    const bool rows_in_partition = valid != nullptr || [] { const char* e = getenv("PDX_HASH_ROWS"); return e && e[0] == '1'; }();
    if (rows_in_partition) gb->rows_part = gb->own<uint32_t>((size_t)n);
    gb->idx16_part = gb->own<uint16_t>((size_t)n);
    bool idx16_written = false;  // by the attempt that succeeded (the LDS build at the first partition level)
    uint32_t* chunk_sum = s.get<uint32_t>((size_t)(nchunks + 1) << kPartBits);  // + digit totals row
    long long* keys_part = s.get<long long>((size_t)n);
    if (s.failed || !gb->bucket8 || !gb->part_off || !gb->slot_part || (rows_in_partition && !gb->rows_part) || !gb->idx16_part) return PDX_OOM;
    {
      // one pass over the keys: bucket byte per row (kept: it is the digit of the value partition of every later aggregation)
      // + the partition histogram
      PDX_PROFILE("hash_bucket_hist", st);
      hipLaunchKernelGGL(k_hash_bucket_hist, dim3((unsigned)ntiles), dim3(kSortBlock), 0, st, keys, valid, key->offset, n, gb->bucket8, gb->part_off);
    }
    int rcp = radix_scan_only<kPartBits>(gb->part_off, ntiles, chunk_sum, true, st);
    if (rcp == PDX_OK && rows_in_partition)  // keys and row ids in ONE scatter (they used to be two kernels ranking the same bucket bytes: 5.0 + 2.5 ms per 1e9 rows)
      rcp = radix_scatter_with_rows<kPartBits>(gb->bucket8, reinterpret_cast<const uint64_t*>(keys), reinterpret_cast<uint64_t*>(keys_part),
                                               gb->rows_part, n, gb->part_off, valid, key->offset, st);
    else if (rcp == PDX_OK)  // the keys alone: 17 instead of 21 B/row, and the pass of the value partition (4 waves per SIMD)
      rcp = radix_scatter_only<kPartBits, uint64_t, uint8_t>(gb->bucket8, reinterpret_cast<const uint64_t*>(keys), nullptr, reinterpret_cast<uint64_t*>(keys_part), n,
                                                             0, false, gb->part_off, st);
    if (rcp != PDX_OK) return rcp;
    uint64_t want = std::max<uint64_t>(next_pow2((uint64_t)n * 2), 1u << 16);
    unsigned int cap = (unsigned int)std::min<uint64_t>(want, 1u << 21);
    unsigned int pb = kPartBits;              // hash bits the rows are currently partitioned by
    uint32_t* bucket_starts = nullptr;        // starts of the 2^pb buckets after a second partition level
    bool lds_failed_at_pb = false;
    for (;;) {
      table = static_cast<Slot*>(pool_alloc(((size_t)cap + 2) * sizeof(Slot)));
      if (!table) return PDX_OOM;
      region = cap >> pb;
      hipLaunchKernelGGL(k_table_init, dim3(grid_for((int64_t)cap + 2, 256, 4)), dim3(256), 0, st, table, (int64_t)cap + 2);
      hipMemsetAsync(ctl, 0, sizeof(HashCtl), st);
      unsigned int limit = (unsigned int)((uint64_t)cap * 7 / 10);
      const char* lenv = getenv("PDX_HASH_LDS");
      const bool use_lds = region <= (unsigned)kLdsRegionMax && !(lenv && lenv[0] == '0') && !lds_failed_at_pb;
      if (use_lds) {
        PDX_PROFILE("hash_probe_lds", st);
        // bucket starts: the offsets row of tile 0 of the partition pass, or the searched starts after a second level
        // skewed buckets (longer than 1.5x the average and 2^20 rows): head rows here, the rest in chunks (k_hash_probe_lds_tail)
        const uint32_t* boff = bucket_starts ? bucket_starts : gb->part_off;
        const size_t nb = (size_t)1 << pb;
        int64_t head_rows = std::max<int64_t>((int64_t)1 << 20, (n >> pb) + (n >> (pb + 1)));
        if (const char* e = getenv("PDX_HASH_HEAD_ROWS")) head_rows = std::max<int64_t>(atoll(e), 1);
        head_rows = (head_rows + 4 * kProbeBlock - 1) / (4 * kProbeBlock) * (4 * kProbeBlock);
        std::vector<TailChunk> chunks;
        if (n > head_rows) {
          std::vector<uint32_t> hoff(nb);
          PDX_HIP(hipMemcpyAsync(hoff.data(), boff, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
          PDX_HIP(hipStreamSynchronize(st));
          for (size_t bi = 0; bi < nb; ++bi) {
            const int64_t bs = hoff[bi], be = bi + 1 < nb ? (int64_t)hoff[bi + 1] : n;
            for (int64_t c0 = bs + head_rows; c0 < be; c0 += kTailChunkRows)
              chunks.push_back(TailChunk{(uint32_t)bi, (uint32_t)c0, (uint32_t)std::min<int64_t>(c0 + kTailChunkRows, be)});
          }
        }
        uint16_t* idx16 = pb == (unsigned)kPartBits ? gb->idx16_part : nullptr;
        idx16_written = idx16 != nullptr;
        if (!chunks.empty() || pb != (unsigned)kPartBits) PDX_TRY(ensure_rows_part(gb, st));  // (the tail kernel tracks first rows per row)
        const bool first_as_pos = !valid && !gb->rows_part;
        if (valid)
          hipLaunchKernelGGL((k_hash_probe_lds<true>), dim3(1u << pb), dim3(kProbeBlock), 0, st, keys_part, gb->rows_part, boff, n, table, cap, region,
                             gb->slot_part, ctl, pb, head_rows, idx16);
        else  // no null keys: rows_part is not read per row (first rows as positions), and with idx16 the 4-byte slot is not written
          hipLaunchKernelGGL((k_hash_probe_lds<false>), dim3(1u << pb), dim3(kProbeBlock), 0, st, keys_part, gb->rows_part, boff, n, table, cap, region,
                             gb->slot_part, ctl, pb, head_rows, idx16);
        if (first_as_pos)
          hipLaunchKernelGGL(k_first_rows_from_pos, dim3((unsigned)(((int64_t)cap + 2 + 255) / 256)), dim3(256), 0, st, table, cap, region,
                             gb->part_off, ntiles, gb->bucket8, n);
        if (!chunks.empty()) {
          TailChunk* dchunks = s.get<TailChunk>(chunks.size());
          if (s.failed) {
            pool_free(table);
            return PDX_OOM;
          }
          PDX_HIP(hipMemcpyAsync(dchunks, chunks.data(), chunks.size() * sizeof(TailChunk), hipMemcpyHostToDevice, st));
          hipLaunchKernelGGL(k_hash_probe_lds_tail, dim3((unsigned)chunks.size()), dim3(kProbeBlock), 0, st, keys_part, gb->rows_part, dchunks, table, cap,
                             region, gb->slot_part, ctl, pb, idx16);
          PDX_HIP(hipStreamSynchronize(st));  // `chunks` (pageable host memory) must outlive the copy
        }
      } else {
        idx16_written = false;
        PDX_TRY(ensure_rows_part(gb, st));
        PDX_PROFILE("hash_probe_part", st);
        constexpr unsigned int kWindowBits = 16;  // 2^16 slots = 1 MB per bucket window; about two buckets are active at a time
        const unsigned int nsweeps = region > (1u << kWindowBits) ? region >> kWindowBits : 1u;
        for (unsigned int sw = 0; sw < nsweeps; ++sw)
          hipLaunchKernelGGL((k_hash_probe_part<4>), dim3((unsigned int)ceil_div(n, 1024 * kProbeTiles)), dim3(256), 0, st, keys_part, gb->rows_part, n, table,
                             cap, region, limit, gb->slot_part, ctl, nsweeps > 1 ? kWindowBits : 31u, sw, pb);
      }
      HashCtl h;
      hipError_t e = hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e != hipSuccess) {
        pool_free(table);
        return hip_fail(e, "k_hash_probe_part");
      }
      if (getenv("PDX_HASH_DEBUG"))
        fprintf(stderr, "[pdx] hash build attempt: pb=%u cap=%u region=%u lds=%d inserted=%u overflow=%u rows_seen=%llu distinct_seen=%llu\n", pb, cap,
                region, (int)use_lds, h.inserted, h.overflow, (unsigned long long)h.rows_seen, (unsigned long long)h.est_distinct);
      if (!h.overflow && h.inserted <= limit) break;
      pool_free(table);
      table = nullptr;
      if (cap >= want * 4 || cap >= (1u << 30)) return fail(PDX_DEVICE, "pdx_groupby_create: hash table overflow at maximum capacity");
      if (use_lds && pb > (unsigned)kPartBits) lds_failed_at_pb = true;  // a skewed bucket outgrew LDS even after the split: memory-side build
      uint64_t next = (uint64_t)cap * 8;
      double groups = 0.0;
      if (h.rows_seen) {
        // the LDS attempt saw d distinct keys in its first r rows (an exact pair, taken after every bucket's first 4096 rows):
        // size the retry for the cardinality K that predicts, d = K (1 - exp(-r / K)) for uniformly mixed keys.  Too few
        // repeats to tell means "more groups than the sample can resolve": plan for min(n, 2^28) keys.
        const double d = (double)h.est_distinct, r = (double)h.rows_seen;
        groups = std::min((double)n, 268435456.0);
        if (r - d > 0.001 * r && r - d >= 512.0) {
          double lo = d, hi = 1e18;  // bisection on K
          for (int it = 0; it < 200; ++it) {
            double mid = std::sqrt(lo * hi);
            if (mid * -std::expm1(-r / mid) < d) lo = mid;
            else hi = mid;
          }
          groups = std::min((double)n, hi * -std::expm1(-(double)n / hi));
        }
        next = std::max<uint64_t>(next, next_pow2((uint64_t)(groups / 0.4) + 1));  // aim at <= 40 % load: the estimate is noisy
      }
      // Very many groups: instead of a table that falls out of the L2 (every probe and atomic then goes to memory: 1.4 s per 1e9
      // rows measured), split the buckets by a second partition level until a bucket's table (<= 8192 slots at <= ~35 % load)
      // fits in LDS again.  The second level is one more stable scatter of (key, row) by the next hash bits; the logical slot id
      // keeps the (index << pb) | bucket form with pb = 8 + extra, so the later sort by slot just sees a longer partitioned prefix.
      const bool split_ok = [] { const char* e = getenv("PDX_HASH_SPLIT"); return !(e && e[0] == '0'); }();
      if (split_ok && pb == (unsigned)kPartBits && groups > 0.0 && next > (1u << 21) && !(lenv && lenv[0] == '0')) {
        int extra = 4;
        while (extra < 8 && groups / (double)((uint64_t)1 << (kPartBits + extra)) > 2800.0) ++extra;  // ~35 % load when there is room ...
        if (groups / (double)((uint64_t)1 << (kPartBits + extra)) <= 4200.0) {  // ... up to ~51 % at the last level (2.7e8 groups)
          const int64_t ntiles2 = ntiles, nchunks2 = nchunks;
          PDX_TRY(ensure_rows_part(gb, st));
          gb->digit2 = gb->own<uint8_t>((size_t)n);
          gb->part_off2 = gb->own<uint32_t>((size_t)ntiles2 << extra);
          uint32_t* chunk_sum2 = s.get<uint32_t>((size_t)(nchunks2 + 1) << extra);
          long long* keys_part2 = s.get<long long>((size_t)n);
          uint32_t* rows_part2 = gb->own<uint32_t>((size_t)n);
          bucket_starts = s.get<uint32_t>(((size_t)1 << (kPartBits + extra)) + 1);
          if (s.failed || !gb->digit2 || !gb->part_off2 || !rows_part2) return PDX_OOM;
          int rc2 = PDX_OK;
          {
            PDX_PROFILE("hash_bucket_hist", st);
#define DIGIT_HIST(B) hipLaunchKernelGGL((k_hash_digit_hist<B>), dim3((unsigned)ntiles2), dim3(kSortBlock), 0, st, keys_part, gb->rows_part, n, kPartBits, \
                                         gb->digit2, gb->part_off2)
            switch (extra) {
              case 4: DIGIT_HIST(4); break;
              case 5: DIGIT_HIST(5); break;
              case 6: DIGIT_HIST(6); break;
              case 7: DIGIT_HIST(7); break;
              default: DIGIT_HIST(8); break;
            }
#undef DIGIT_HIST
          }
          rc2 = radix_scan_dispatch(extra, gb->part_off2, ntiles2, chunk_sum2, true, st);
#define SCATTER2(B)                                                                                                                                   \
  if (rc2 == PDX_OK)                                                                                                                                  \
    rc2 = radix_scatter_only<B, uint64_t, uint8_t>(gb->digit2, reinterpret_cast<const uint64_t*>(keys_part), nullptr,                                   \
                                                   reinterpret_cast<uint64_t*>(keys_part2), n, 0, false, gb->part_off2, st);                           \
  if (rc2 == PDX_OK) rc2 = radix_scatter_only<B, uint32_t, uint8_t>(gb->digit2, gb->rows_part, nullptr, rows_part2, n, 0, false, gb->part_off2, st)
          switch (extra) {
            case 4: SCATTER2(4); break;
            case 5: SCATTER2(5); break;
            case 6: SCATTER2(6); break;
            case 7: SCATTER2(7); break;
            default: SCATTER2(8); break;
          }
#undef SCATTER2
          if (rc2 != PDX_OK) return rc2;
          keys_part = keys_part2;
          gb->rows_part = rows_part2;  // (the first-level array stays owned by the handle until it is destroyed)
          pb = kPartBits + extra;
          gb->digit2_bits = extra;
          hipLaunchKernelGGL(k_bucket_starts, dim3(grid_for((int64_t)1 << pb, 256)), dim3(256), 0, st, keys_part, gb->rows_part, n, pb, bucket_starts);
          PDX_LAUNCH_CHECK();
          cap = (unsigned int)kLdsRegionMax << pb;
          continue;
        }
      }
      cap = (unsigned int)std::min<uint64_t>(std::max<uint64_t>(next, (uint64_t)cap * 2), std::max<uint64_t>(want * 4, 1u << 16));
    }
    gb->part_bits = (int)pb;
    if (!idx16_written) gb->idx16_part = nullptr;  // (the block stays owned by the handle; nothing reads it)
    gb->slot_part_ready = !idx16_written;          // the LDS build at the first level writes the 2-byte region indexes only
    gb->owned.push_back(table);
    {
      Slot sp[2];
      PDX_HIP(hipMemcpyAsync(sp, table + cap, sizeof(sp), hipMemcpyDeviceToHost, st));
      PDX_HIP(hipStreamSynchronize(st));
      gb->special_slots = sp[0].first != kNoRow || sp[1].first != kNoRow;
    }
    null_slot = cap;
    nslots = (int64_t)cap + 2;
  } else {
  // table capacity: start at min(2^21, pow2 >= 2n) and grow x8 whenever more than 70 % of the slots fill up
  uint64_t want = next_pow2((uint64_t)n * 2);
  unsigned int cap = (unsigned int)std::min<uint64_t>(want, 1u << 21);
  for (;;) {
    table = static_cast<Slot*>(pool_alloc(((size_t)cap + 2) * sizeof(Slot)));
    if (!table) return PDX_OOM;
    hipLaunchKernelGGL(k_table_init, dim3(grid_for((int64_t)cap + 2, 256, 4)), dim3(256), 0, st, table, (int64_t)cap + 2);
    hipMemsetAsync(ctl, 0, sizeof(HashCtl), st);
    unsigned int limit = (unsigned int)((uint64_t)cap * 7 / 10);
    if (cap >= want) limit = cap;  // a table of >= 2n slots can never overflow
    {
      PDX_PROFILE("hash_insert", st);
      hipLaunchKernelGGL(k_hash_insert, dim3(grid_for(n, 256, 8)), dim3(256), 0, st, keys, valid, key->offset, n, table, cap, limit,
                         gb->slot_of_row, ctl);
    }
    HashCtl h;
    hipError_t e = hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        pool_free(table);
        return hip_fail(e, "k_hash_insert");
      }
    if (!h.overflow) break;
    pool_free(table);
    if (cap >= want) return fail(PDX_DEVICE, "pdx_groupby_create: hash table overflow at maximum capacity");
    cap = (unsigned int)std::min<uint64_t>((uint64_t)cap * 8, want);
  }
  gb->owned.push_back(table);
  null_slot = cap;
  nslots = (int64_t)cap + 2;
  }
  gb->nslots = nslots;
  gb->slot_bits = ilog2((uint64_t)nslots);
  gb->gid_of_slot = gb->own<uint32_t>((size_t)nslots);
  // occupied slots in slot order
  uint32_t* occ_slot_tmp = s.get<uint32_t>((size_t)std::min<int64_t>(nslots, n + 2));
  uint32_t* occ_first_tmp = s.get<uint32_t>((size_t)std::min<int64_t>(nslots, n + 2));
  if (s.failed || !gb->gid_of_slot) return PDX_OOM;
  int64_t G = 0;
  int rc = compact_indices(nslots, OccPred{table, dense_first, region, null_slot}, OccEmit{table, dense_first, region, null_slot, occ_slot_tmp, occ_first_tmp},
                           &G, s, st);
  if (rc != PDX_OK) return rc;
  gb->G = G;
  gb->occ_slot = gb->own<uint32_t>((size_t)G);
  gb->gid_of_occ = gb->own<uint32_t>((size_t)G);
  gb->uniques = gb->own<int64_t>((size_t)G);
  gb->unique_ok = gb->own<uint8_t>((size_t)G);
  gb->first_rows = gb->own<int64_t>((size_t)G);
  if (s.failed || !gb->occ_slot || !gb->gid_of_occ || !gb->uniques || !gb->unique_ok || !gb->first_rows) return PDX_OOM;
  hipMemcpyAsync(gb->occ_slot, occ_slot_tmp, (size_t)G * sizeof(uint32_t), hipMemcpyDeviceToDevice, st);
  // order groups by first occurrence: gid = rank of the group's first row among all first rows (bit map + block prefix, no sort).
  // Nearly every row its own group: the random bit sets / word reads of the map cost more than sorting the (first row, slot)
  // pairs (measured at 1.5e8 groups of 2.5e8 rows, pdx_reindex_indices: 65 ms with the map -- 5.6 ms of bit sets, > 10 ms of rank
  // reads -- against 54 ms with the sort)
  const bool rank_by_sort = G > ((int64_t)1 << 22) && G * 8 > n;
  if (rank_by_sort) {
    uint32_t* k0 = s.get<uint32_t>((size_t)G);
    uint32_t* v0 = s.get<uint32_t>((size_t)G);
    uint32_t* k1 = s.get<uint32_t>((size_t)G);
    uint32_t* v1 = s.get<uint32_t>((size_t)G);
    PDX_SCRATCH_CHECK(s);
    const uint32_t *ks = nullptr, *vs = nullptr;  // sort (first_row -> slot); first rows are distinct so any order of ties is moot
    rc = radix_sort_pairs<uint32_t>(occ_first_tmp, occ_slot_tmp, k0, v0, k1, v1, G, ilog2((uint64_t)n + 1) < 31 ? ilog2((uint64_t)n + 1) : 31, &ks, &vs,
                                    true, s, st);
    if (rc != PDX_OK) return rc;
    const int g = grid_for(G, 256);
    hipLaunchKernelGGL(k_assign_gids, dim3(g), dim3(256), 0, st, table, dense_min, dense_mask, gb->gid_of_slot, ks, vs, G, null_slot, gb->uniques,
                       gb->unique_ok, gb->first_rows, region);
    hipLaunchKernelGGL(k_gid_of_occ, dim3(g), dim3(256), 0, st, gb->gid_of_slot, gb->occ_slot, G, gb->gid_of_occ);
  } else {
    const int64_t nwords = (n + 63) >> 6, nrb = ceil_div(nwords, (int64_t)kRankWords);
    unsigned long long* bits = s.get<unsigned long long>((size_t)nwords);
    int64_t* block_pre = s.get<int64_t>((size_t)nrb);
    PDX_SCRATCH_CHECK(s);
    PDX_HIP(hipMemsetAsync(bits, 0, (size_t)nwords * sizeof(unsigned long long), st));
    const int g = grid_for(G, 256);
    hipLaunchKernelGGL(k_mark_first_rows, dim3(g), dim3(256), 0, st, occ_first_tmp, G, bits);
    hipLaunchKernelGGL(k_rank_block_counts, dim3(grid_for(nrb * kRankWords, 256)), dim3(256), 0, st, bits, nwords, nrb, block_pre);
    PDX_TRY((device_exclusive_scan<int64_t, SumOp>(block_pre, block_pre, nrb, (int64_t*)nullptr, s, st)));
    hipLaunchKernelGGL(k_assign_gids_ranked, dim3(g), dim3(256), 0, st, table, dense_min, dense_mask, gb->gid_of_slot, occ_first_tmp, occ_slot_tmp, G, bits,
                       block_pre, null_slot, gb->uniques, gb->unique_ok, gb->first_rows, region, gb->gid_of_occ);
  }
  hipError_t e = hipGetLastError();
  // no drain of the stream: everything the caller can read on the host (G) is known; later calls are ordered by the stream / the event
  if (e == hipSuccess) e = hipEventCreateWithFlags(&gb->ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventRecord(gb->ready, st);
  if (e != hipSuccess) return hip_fail(e, "pdx_groupby_create");
  gb->create_stream = st;
  *out = owner.release();
  return PDX_OK;
}

int pdx_groupby_destroy(pdx_groupby* gb) {
  delete gb;
  return PDX_OK;
}
int64_t pdx_groupby_num_groups(const pdx_groupby* gb) { return gb ? gb->G : -1; }
int64_t pdx_groupby_num_rows(const pdx_groupby* gb) { return gb ? gb->n : -1; }

int pdx_groupby_unique_keys(const pdx_groupby* gb, pdx_mut_column* out, void* stream) {
  if (!gb || !out) return fail(PDX_INVALID, "pdx_groupby_unique_keys: null argument");
  if (out->length < gb->G) return fail(PDX_INVALID, "pdx_groupby_unique_keys: output too small");
  hipStream_t st = as_stream(stream);
  gb->use_on(st);  // ordered behind the handle's creation; frees of its blocks are ordered behind this stream
  out->length = gb->G;
  out->null_count = -1;
  if (gb->G == 0) return PDX_OK;
  PDX_HIP(hipMemcpyAsync(out->values, gb->uniques, (size_t)gb->G * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
  if (out->validity) {
    hipLaunchKernelGGL(k_pack_bytes, dim3(grid_for((gb->G + 7) / 8, 256)), dim3(256), 0, st, gb->unique_ok, gb->G,
                       static_cast<uint8_t*>(out->validity));
    PDX_LAUNCH_CHECK();
  }
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}

int pdx_groupby_first_rows(const pdx_groupby* gb, int64_t* out_rows, void* stream) {
  if (!gb || !out_rows) return fail(PDX_INVALID, "pdx_groupby_first_rows: null argument");
  hipStream_t st = as_stream(stream);
  gb->use_on(st);  // ordered behind the handle's creation; frees of its blocks are ordered behind this stream
  if (gb->G) PDX_HIP(hipMemcpyAsync(out_rows, gb->first_rows, (size_t)gb->G * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}

int pdx_groupby_group_ids(pdx_groupby* gb, uint32_t* out_ids, void* stream) {
  if (!gb || !out_ids) return fail(PDX_INVALID, "pdx_groupby_group_ids: null argument");
  hipStream_t st = as_stream(stream);
  gb->use_on(st);  // ordered behind the handle's creation; frees of its blocks are ordered behind this stream
  if (gb->n == 0) return PDX_OK;
  if (gb->mode == 0 && gb->slot_part) PDX_TRY(ensure_slot_part(gb, st));
  if (gb->mode == 0 && gb->slot_part) PDX_TRY(ensure_rows_part(gb, st));
  if (gb->mode == 0 && gb->slot_part)
    hipLaunchKernelGGL(k_part_row_gids, dim3(grid_for(gb->n, 256, 4)), dim3(256), 0, st, gb->gid_of_slot, gb->slot_part, gb->rows_part, gb->n,
                       (const int64_t*)nullptr, out_ids, (int64_t*)nullptr);
  else if (gb->mode == 0)
    hipLaunchKernelGGL(k_row_gids, dim3(grid_for(gb->n, 256, 4)), dim3(256), 0, st, gb->gid_of_slot, gb->slot_of_row, gb->n, out_ids);
  else
    hipLaunchKernelGGL(k_seg_row_ids, dim3(grid_for(gb->n, 256, 4)), dim3(256), 0, st, gb->seg_start, gb->G, gb->n, out_ids);
  PDX_LAUNCH_CHECK();
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}

int pdx_groupby_map_ids(pdx_groupby* gb, const int64_t* map, int64_t* out, void* stream) {
  if (!gb || !map || !out) return fail(PDX_INVALID, "pdx_groupby_map_ids: null argument");
  hipStream_t st = as_stream(stream);
  gb->use_on(st);  // ordered behind the handle's creation; frees of its blocks are ordered behind this stream
  if (gb->n == 0) return PDX_OK;
  if (gb->mode == 0 && gb->slot_part) PDX_TRY(ensure_slot_part(gb, st));
  if (gb->mode == 0 && gb->slot_part) PDX_TRY(ensure_rows_part(gb, st));
  if (gb->mode == 0 && gb->slot_part)
    hipLaunchKernelGGL(k_part_row_gids, dim3(grid_for(gb->n, 256, 4)), dim3(256), 0, st, gb->gid_of_slot, gb->slot_part, gb->rows_part, gb->n, map,
                       (uint32_t*)nullptr, out);
  else
    hipLaunchKernelGGL(k_map_ids, dim3(grid_for(gb->n, 256, 4)), dim3(256), 0, st, gb->mode == 0 ? gb->gid_of_slot : nullptr, gb->slot_of_row,
                       gb->seg_start, gb->G, gb->n, map, out);
  PDX_LAUNCH_CHECK();
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}

}  // extern "C"

namespace pdx {
// offsets[g] = first position of group id g in the sorted ids (g = G: n); rows[i] widened to int64
__global__ void k_groupings_finish(const uint32_t* __restrict__ sorted_ids, const uint32_t* __restrict__ rows32, int64_t n, int64_t G,
                                   int64_t* __restrict__ out_rows, int64_t* __restrict__ out_offsets) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n + G + 1; i += stride) {
    if (i < n) {
      out_rows[i] = (int64_t)rows32[i];
      continue;
    }
    const int64_t g = i - n;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)sorted_ids[mid] < g) lo = mid + 1;
      else hi = mid;
    }
    out_offsets[g] = lo;
  }
}
__global__ void k_iota_rows(uint32_t* __restrict__ p, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = (uint32_t)i;
}
}  // namespace pdx

extern "C" {
// Grouper::MakeGroupings (src/dataframe.cpp:1546, 1562): the rows of every group, ascending, groups in group-id order
int pdx_groupby_groupings(pdx_groupby* gb, int64_t* out_rows, int64_t* out_offsets, void* stream) {
  if (!gb || !out_rows || !out_offsets) return fail(PDX_INVALID, "pdx_groupby_groupings: null argument");
  hipStream_t st = as_stream(stream);
  gb->use_on(st);
  const int64_t n = gb->n, G = gb->G;
  if (n == 0) {
    PDX_HIP(hipMemsetAsync(out_offsets, 0, sizeof(int64_t) * (size_t)(G + 1), st));
    PDX_HIP(hipStreamSynchronize(st));
    return PDX_OK;
  }
  Scratch s;
  uint32_t* ids = s.get<uint32_t>((size_t)n);
  uint32_t* rows = s.get<uint32_t>((size_t)n);
  uint32_t* k0 = s.get<uint32_t>((size_t)n);
  uint32_t* k1 = s.get<uint32_t>((size_t)n);
  uint32_t* v0 = s.get<uint32_t>((size_t)n);
  uint32_t* v1 = s.get<uint32_t>((size_t)n);
  PDX_SCRATCH_CHECK(s);
  PDX_TRY(pdx_groupby_group_ids(gb, ids, stream));
  hipLaunchKernelGGL(k_iota_rows, dim3(grid_for(n, 256, 4)), dim3(256), 0, st, rows, n);
  PDX_LAUNCH_CHECK();
  const uint32_t* ks = ids;
  const uint32_t* vs = rows;
  if (gb->mode == 0 && G > 1) {  // (segments mode: the rows are grouped as they stand)
    const int bits = std::max(1, ilog2((uint64_t)G));
    PDX_TRY((radix_sort_pairs<uint32_t>(ids, rows, k0, v0, k1, v1, n, bits, &ks, &vs, true, s, st)));
  }
  hipLaunchKernelGGL(k_groupings_finish, dim3(grid_for(n + G + 1, 256, 4)), dim3(256), 0, st, ks, vs, n, G, out_rows, out_offsets);
  PDX_LAUNCH_CHECK();
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}

// ---------------------------------------------------------------- stage 2 of pdx_groupby_agg: reducers over a GroupedLayout
}  // extern "C"

namespace pdx {
struct AggRequest {
  SegOut o{};
  bool want_pw = false, want_mm = false, want_is = false, want_std5 = false;
  // the "next" kinds (variance, stddev, product, first, last) run after the five standard ones on the same grouped values
  double *var_out = nullptr, *std_out = nullptr;
  void *prod_out = nullptr, *first_out = nullptr, *last_out = nullptr;
  bool is_f = true;
  bool std_only() const { return (want_std5 || var_out || std_out) && !prod_out && !first_out && !last_out; }  // (variance: two more fused passes)
};

// one segmented reduce of `vals` (float64 or int64 per f64) over `G` segments of `nrows` grouped rows into `oo`; nullable: the null flag of
// every grouped row is bit 31 of flag_keys (or read in place from row_valid, segments mode); ok_bytes: 1 = the group has a valid value
static int reduce_segments(bool nullable, const uint32_t* flag_keys, const uint8_t* row_valid, int64_t valid_off, const uint32_t* seg_start, int64_t G,
                           int64_t nrows, const void* vals, bool f64, const SegOut& oo, bool pw, bool mm, bool is, const uint32_t* oidx, uint8_t* ok_bytes,
                           Scratch& s, hipStream_t st) {
  PDX_PROFILE("seg_reduce", st);
  const int64_t n = nrows;
  if (!nullable) {
    if (f64) return launch_seg_reduce_dense<double>(static_cast<const double*>(vals), seg_start, G, oidx, oo, pw, mm, is, n, s, st);
    return launch_seg_reduce_dense<long long>(static_cast<const long long*>(vals), seg_start, G, oidx, oo, pw, mm, is, n, s, st);
  }
  const int grid = (int)std::min<int64_t>(ceil_div(G, kSegWaves), (int64_t)kCUs * 8);
  if (f64)
    hipLaunchKernelGGL((k_seg_reduce_nullable<double>), dim3(grid), dim3(kSegWaves * 64), 0, st, static_cast<const double*>(vals), flag_keys, row_valid,
                       valid_off, seg_start, G, oidx, oo, ok_bytes);
  else
    hipLaunchKernelGGL((k_seg_reduce_nullable<long long>), dim3(grid), dim3(kSegWaves * 64), 0, st, static_cast<const long long*>(vals), flag_keys,
                       row_valid, valid_off, seg_start, G, oidx, oo, ok_bytes);
  PDX_LAUNCH_CHECK();
  return reduce_huge_nullable_groups(vals, f64 ? PDX_FLOAT64 : PDX_INT64, flag_keys, row_valid, valid_off, seg_start, G, oidx, n, oo, ok_bytes, s, st);
}
// ... over a full layout; vals == the layout's own values or an array in the same order
static int reduce_full(const pdx_groupby* gb, const GroupedLayout& L, const void* vals, bool f64, const SegOut& oo, bool pw, bool mm, bool is,
                       const uint32_t* oidx, uint8_t* ok_bytes, Scratch& s, hipStream_t st) {
  return reduce_segments(L.validity != nullptr, L.flag_keys, L.row_valid, L.offset, L.seg_start, gb->G, gb->n, vals, f64, oo, pw, mm, is, oidx, ok_bytes, s, st);
}
// ... over the side form of a fused layout (the rows of its long runs): EVERY group's outputs are written -- zeros / nulls for the groups
// outside the long runs -- so it runs before the fused kernel, which then writes the groups of the runs it walks
static int reduce_side(const pdx_groupby* gb, const GroupedLayout& L, const SegOut& oo, bool pw, bool mm, bool is, uint8_t* ok_bytes, const double* sqmean,
                       Scratch& s, hipStream_t st) {
  const bool is_f = L.dtype == PDX_FLOAT64, nullable = L.validity != nullptr;
  const void* vals = L.side_vals;
  bool f64 = is_f;
  if (sqmean) {  // second pass of variance: (x - mean of x's group)^2, summed over the same valid runs
    double* d = s.get<double>((size_t)L.side_rows);
    PDX_SCRATCH_CHECK(s);
    PDX_PROFILE("seg_sqdev", st);
    const int grid = (int)std::min<int64_t>(ceil_div(gb->G, 4), (int64_t)kCUs * 16);
    if (is_f) hipLaunchKernelGGL((k_seg_sqdev<double>), dim3(grid), dim3(256), 0, st, reinterpret_cast<const double*>(L.side_vals), L.side_seg, gb->G, sqmean, d, gb->gid_of_occ);
    else hipLaunchKernelGGL((k_seg_sqdev<long long>), dim3(grid), dim3(256), 0, st, reinterpret_cast<const long long*>(L.side_vals), L.side_seg, gb->G, sqmean, d, gb->gid_of_occ);
    PDX_LAUNCH_CHECK();
    vals = d;
    f64 = true;
  }
  return reduce_segments(nullable, L.side_keys, nullptr, 0, L.side_seg, gb->G, L.side_rows, vals, f64, oo, pw, mm, is, gb->gid_of_occ, ok_bytes, s, st);
}

// The fused last digit: rank by the top 6 slot bits + Arrow's leaf / counter recurrence in one pass over a fused layout.
// ok_bytes (nullable values): 1 = the group has a valid value; sqmean: second pass of variance.
static int reduce_fused(const pdx_groupby* gb, const GroupedLayout& L, const AggTuning& t, const SegOut& oo, bool pw, bool mm, bool is, uint8_t* okbytes,
                        const double* sqmean, Scratch& s, hipStream_t st, std::string* reducer) {
  if (L.side_rows) PDX_TRY(reduce_side(gb, L, oo, pw, mm, is, okbytes, sqmean, s, st));
  const unsigned int max_run = L.max_run ? L.max_run : 0xFFFFFFFFu;
  PDX_PROFILE("fused_last_digit_reduce", st);
  const bool nullable = L.validity != nullptr, is_f = L.dtype == PDX_FLOAT64;
  const int64_t nruns = L.nruns, n = gb->n;
  const int low_bits = L.low_bits;
  const uint8_t* keys8 = L.keys8;
  const uint32_t* keys_sorted = L.fkeys;
  const uint64_t* vs = L.fvals;
  const uint32_t* run_start = L.run_start;
  const int grid = (int)std::min<int64_t>(nruns, (int64_t)kCUs * t.flr_wgs_per_cu);
  const bool dense = pw && !mm && !is && !nullable;
  // (values with nulls, sum / mean / count: the thread-per-leaf form is opt-in, PDX_FLR_NULL_PW=1 -- measured 11.3 ms against
  //  10.7 ms of the literal per-lane replay at 5 % nulls: its per-group walk over the leaf markers is as serial as the replay)
  const bool nullpw = pw && !mm && !is && nullable && t.null_pw;
#define FLR_LAUNCH(TT, DD)                                                                                                                       \
  if (keys8 && nullpw)                                                                                                                           \
    hipLaunchKernelGGL((k_flr_reduce<TT, false, uint8_t, true>), dim3(grid), dim3(kSortBlock), 0, st, keys8, reinterpret_cast<const TT*>(vs), run_start, \
                       nruns, low_bits, gb->gid_of_slot, oo, okbytes, (int)pw, (int)mm, (int)is, 1, sqmean, max_run);                              \
  else if (nullpw)                                                                                                                               \
    hipLaunchKernelGGL((k_flr_reduce<TT, false, uint32_t, true>), dim3(grid), dim3(kSortBlock), 0, st, keys_sorted, reinterpret_cast<const TT*>(vs),   \
                       run_start, nruns, low_bits, gb->gid_of_slot, oo, okbytes, (int)pw, (int)mm, (int)is, 1, sqmean, max_run);                   \
  else if (keys8)                                                                                                                                \
    hipLaunchKernelGGL((k_flr_reduce<TT, DD, uint8_t>), dim3(grid), dim3(kSortBlock), 0, st, keys8, reinterpret_cast<const TT*>(vs), run_start, nruns, \
                       low_bits, gb->gid_of_slot, oo, okbytes, (int)pw, (int)mm, (int)is, nullable ? 1 : 0, sqmean, max_run);                      \
  else                                                                                                                                           \
    hipLaunchKernelGGL((k_flr_reduce<TT, DD>), dim3(grid), dim3(kSortBlock), 0, st, keys_sorted, reinterpret_cast<const TT*>(vs), run_start, nruns, \
                       low_bits, gb->gid_of_slot, oo, okbytes, (int)pw, (int)mm, (int)is, nullable ? 1 : 0, sqmean, max_run)
  // one WAVE per run (flr_wave.hpp) for everything but the dense sum / mean / count: nullable values, min / max and int64 sums
  // were a per-lane replay by wave 0 alone in the workgroup-per-run kernel (5 % nulls: 10.7 -> 4.9 ms per 1e9 rows); the dense
  // fast path of k_flr_reduce (thread per leaf) is still ahead of the wave form (3.0 vs 3.3 ms).  PDX_FLR_WAVE=1 / 0 force either.
  const bool wave_form = !nullpw && (t.wave_force >= 0 ? t.wave_force != 0 : !dense);
  if (reducer) *reducer = wave_form ? "flr_wave" : (nullpw ? "flr_reduce_nullpw" : (dense ? "flr_reduce_dense" : "flr_reduce"));
  if (wave_form) {
    const int wgrid = (int)std::min<int64_t>(nruns, (int64_t)kCUs * t.fw_wgs_per_cu);
    const bool pw_only = pw && !mm && !is;
    // counter levels: a group cannot outgrow its run, and a leaf holds 16 rows unless nulls cut it short
    const uint64_t max_leaves = nullable ? (uint64_t)L.hmax + 1 : (uint64_t)L.hmax / 16 + 2;
    const int fw_levels = std::max(2, ilog2(max_leaves + 1));  // 2^levels > leaves: level index <= levels - 1
    const size_t fw_lds = (size_t)fw_lds_bytes(nullable, fw_levels);
#define FW_LAUNCH(TT, KK, KPTR, NN, PP)                                                                                                  \
  hipLaunchKernelGGL((k_flr_wave<TT, KK, NN, PP>), dim3(wgrid), dim3(64), fw_lds, st, KPTR, reinterpret_cast<const TT*>(vs), n, run_start, nruns, low_bits, \
                     gb->gid_of_slot, oo, okbytes, (int)pw, (int)mm, (int)is, sqmean, fw_levels, max_run)
#define FW_DISPATCH(TT)                                                                        \
  if (keys8) {                                                                                 \
    if (nullable) { if (pw_only) FW_LAUNCH(TT, uint8_t, keys8, true, true); else FW_LAUNCH(TT, uint8_t, keys8, true, false); }          \
    else { if (pw_only) FW_LAUNCH(TT, uint8_t, keys8, false, true); else FW_LAUNCH(TT, uint8_t, keys8, false, false); }              \
  } else {                                                                                     \
    if (nullable) { if (pw_only) FW_LAUNCH(TT, uint32_t, keys_sorted, true, true); else FW_LAUNCH(TT, uint32_t, keys_sorted, true, false); } \
    else { if (pw_only) FW_LAUNCH(TT, uint32_t, keys_sorted, false, true); else FW_LAUNCH(TT, uint32_t, keys_sorted, false, false); } \
  }
    if (is_f) { FW_DISPATCH(double) } else { FW_DISPATCH(long long) }
#undef FW_DISPATCH
#undef FW_LAUNCH
  } else if (is_f) {
    if (dense) { FLR_LAUNCH(double, true); }
    else { FLR_LAUNCH(double, false); }
  } else {
    if (dense) { FLR_LAUNCH(long long, true); }
    else { FLR_LAUNCH(long long, false); }
  }
#undef FLR_LAUNCH
  PDX_LAUNCH_CHECK();
  return PDX_OK;
}

__global__ void k_mean_from_cache(const double* __restrict__ sum, const long long* __restrict__ cnt, int64_t G, double* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < G; i += stride) out[i] = cnt[i] ? pw_mean(sum[i], (double)cnt[i]) : 0.0;
}

static GroupedLayout* find_bound(pdx_groupby* gb, const pdx_column* values) {
  const void* vv = validity_or_null(values);
  for (auto& b : gb->bound)
    if (b->values == values->values && b->offset == values->offset && b->dtype == values->dtype && b->validity == vv) {
      b->last_use = ++gb->use_clock;
      return b.get();
    }
  return nullptr;
}
static size_t bind_limit_bytes(const pdx_groupby* gb) {
  if (gb->bind_limit) return gb->bind_limit;
  if (const char* e = getenv("PDX_BIND_MAX_BYTES")) return (size_t)atoll(e);
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    return (size_t)64 << 30;
  }
  return total_b / 4;
}
// least recently used layouts go first until the bound set fits the limit (the newest one always stays)
static void enforce_bind_limit(pdx_groupby* gb, const GroupedLayout* keep) {
  const size_t limit = bind_limit_bytes(gb);
  for (;;) {
    size_t total = 0;
    for (auto& b : gb->bound) total += b->bytes;
    if (total <= limit || gb->bound.size() <= 1) return;
    size_t victim = gb->bound.size();
    for (size_t i = 0; i < gb->bound.size(); ++i)
      if (gb->bound[i].get() != keep && (victim == gb->bound.size() || gb->bound[i]->last_use < gb->bound[victim]->last_use)) victim = i;
    if (victim == gb->bound.size()) return;
    gb->bound[victim]->stream = gb->stream;
    gb->bound.erase(gb->bound.begin() + (long)victim);
  }
}
}  // namespace pdx

extern "C" {

int pdx_groupby_bind(pdx_groupby* gb, const pdx_column* values, void* stream) {
  if (!gb) return fail(PDX_INVALID, "pdx_groupby_bind: null handle");
  PDX_TRY(check_column(values, "pdx_groupby_bind"));
  if (values->length != gb->n) return fail(PDX_INVALID, "pdx_groupby_bind: values length differs from the grouped key length");
  if (values->dtype != PDX_FLOAT64 && values->dtype != PDX_INT64)
    return fail(PDX_NOT_IMPLEMENTED, "pdx_groupby_bind: values must be int64 or float64 (boolean columns are order-free: nothing to keep)");
  hipStream_t st = as_stream(stream);
  gb->use_on(st);
  if (find_bound(gb, values)) return PDX_OK;
  std::unique_ptr<GroupedLayout> L(new GroupedLayout());
  L->bound = true;
  L->last_use = ++gb->use_clock;
  // registration only: the first pdx_groupby_agg of the column builds the layout ITS kinds need (the fused form for the standard
  // kinds, the fully sorted form for product / first / last), later calls reuse it
  L->values = values->values;
  L->validity = validity_or_null(values);
  L->offset = values->offset;
  L->dtype = values->dtype;
  L->stream = st;
  gb->bound.push_back(std::move(L));
  enforce_bind_limit(gb, gb->bound.back().get());
  return PDX_OK;
}
int pdx_groupby_unbind(pdx_groupby* gb, const pdx_column* values) {
  if (!gb) return fail(PDX_INVALID, "pdx_groupby_unbind: null handle");
  for (auto& b : gb->bound) b->stream = gb->stream;
  if (!values) {
    gb->bound.clear();
    return PDX_OK;
  }
  const void* vv = validity_or_null(values);
  for (size_t i = 0; i < gb->bound.size(); ++i)
    if (gb->bound[i]->values == values->values && gb->bound[i]->offset == values->offset && gb->bound[i]->dtype == values->dtype &&
        gb->bound[i]->validity == vv) {
      gb->bound.erase(gb->bound.begin() + (long)i);
      break;
    }
  return PDX_OK;
}
int pdx_groupby_bind_limit(pdx_groupby* gb, size_t max_bytes) {
  if (!gb) return fail(PDX_INVALID, "pdx_groupby_bind_limit: null handle");
  gb->bind_limit = max_bytes;
  enforce_bind_limit(gb, nullptr);
  return PDX_OK;
}
int64_t pdx_groupby_bound_bytes(const pdx_groupby* gb) {
  if (!gb) return -1;
  size_t total = 0;
  for (auto& b : gb->bound) total += b->bytes;
  return (int64_t)total;
}
int pdx_groupby_last_plan(const pdx_groupby* gb, char* buf, size_t buf_len) {
  if (!gb || !buf || buf_len == 0) return fail(PDX_INVALID, "pdx_groupby_last_plan: null argument");
  if (gb->last_plan.size() + 1 > buf_len) return fail(PDX_INVALID, "pdx_groupby_last_plan: buffer too small");
  memcpy(buf, gb->last_plan.c_str(), gb->last_plan.size() + 1);
  return PDX_OK;
}

int pdx_groupby_agg(pdx_groupby* gb, const pdx_column* values, const int* kinds, int nk, pdx_mut_column* outs, void* stream) {
  if (!gb || !kinds || !outs || nk <= 0) return fail(PDX_INVALID, "pdx_groupby_agg: null argument");
  PDX_TRY(check_column(values, "pdx_groupby_agg"));
  if (values->length != gb->n) return fail(PDX_INVALID, "pdx_groupby_agg: values length differs from the grouped key length");
  // all / any / count_distinct (and boolean values) are order-free: they live in groupby_extra.hip and come back here for the
  // standard kinds named in the same request
  bool extra = values->dtype == PDX_BOOL;
  for (int k = 0; k < nk; ++k) extra = extra || kinds[k] == PDX_AGG_ALL || kinds[k] == PDX_AGG_ANY || kinds[k] == PDX_AGG_COUNT_DISTINCT;
  if (extra) return groupby_agg_extra(gb, values, kinds, nk, outs, stream);
  const bool is_f = values->dtype == PDX_FLOAT64;
  if (!is_f && values->dtype != PDX_INT64) return fail(PDX_NOT_IMPLEMENTED, "pdx_groupby_agg: values must be int64 or float64");
  hipStream_t st = as_stream(stream);
  gb->use_on(st);  // ordered behind the handle's creation; frees of its blocks are ordered behind this stream
  const int64_t n = gb->n, G = gb->G;
  const uint8_t* vvalid = validity_or_null(values);
  AggRequest rq;
  rq.is_f = is_f;
  SegOut& o = rq.o;
  for (int k = 0; k < nk; ++k) {
    pdx_mut_column* oc = &outs[k];
    if (oc->length < G) return fail(PDX_INVALID, "pdx_groupby_agg: output too small");
    if (G && !oc->values) return fail(PDX_INVALID, "pdx_groupby_agg: null output buffer");
    int want_dt;
    bool needs_validity = vvalid != nullptr;
    switch (kinds[k]) {
      case PDX_AGG_SUM:
        want_dt = is_f ? PDX_FLOAT64 : PDX_INT64;
        if (is_f) { o.sum_f = static_cast<double*>(oc->values); rq.want_pw = true; }
        else { o.sum_i = static_cast<long long*>(oc->values); rq.want_is = true; }
        rq.want_std5 = true;
        break;
      case PDX_AGG_MEAN: want_dt = PDX_FLOAT64; o.mean = static_cast<double*>(oc->values); rq.want_pw = true; rq.want_std5 = true; break;
      case PDX_AGG_MIN: want_dt = values->dtype; o.vmin = oc->values; rq.want_mm = true; rq.want_std5 = true; break;
      case PDX_AGG_MAX: want_dt = values->dtype; o.vmax = oc->values; rq.want_mm = true; rq.want_std5 = true; break;
      case PDX_AGG_COUNT: want_dt = PDX_INT64; o.count = static_cast<long long*>(oc->values); needs_validity = false; rq.want_std5 = true; break;
      case PDX_AGG_VARIANCE: want_dt = PDX_FLOAT64; rq.var_out = static_cast<double*>(oc->values); break;
      case PDX_AGG_STDDEV: want_dt = PDX_FLOAT64; rq.std_out = static_cast<double*>(oc->values); break;
      case PDX_AGG_PRODUCT: want_dt = values->dtype; rq.prod_out = oc->values; break;
      case PDX_AGG_FIRST: want_dt = values->dtype; rq.first_out = oc->values; break;
      case PDX_AGG_LAST: want_dt = values->dtype; rq.last_out = oc->values; break;
      default: return fail(PDX_INVALID, "pdx_groupby_agg: unknown aggregate kind");
    }
    if (oc->dtype != want_dt) return fail(PDX_INVALID, "pdx_groupby_agg: output dtype does not match the aggregate's result type");
    if (needs_validity && !oc->validity)
      return fail(PDX_INVALID, "pdx_groupby_agg: values carry nulls but an output has no validity buffer");
    oc->length = G;
    oc->null_count = needs_validity ? -1 : 0;
  }
  if (G == 0) return PDX_OK;
  const AggTuning t = AggTuning::read();
  // ---- stage 1: the grouped layout (a bound column brings its own)
  GroupedLayout local;
  GroupedLayout* Lp = find_bound(gb, values);
  const bool bound = Lp != nullptr;
  if (!Lp) Lp = &local;
  GroupedLayout& L = *Lp;
  const bool std_only = rq.std_only();
  const size_t bytes_before = L.bytes;
  // count / min / max / int64 sum do not depend on the order of a group's rows: when nothing else is asked for and no sorted layout of
  // the column exists yet, they skip the value sort (gb_acc.hpp: one partition pass + accumulators in LDS)
  const AccTuning at = AccTuning::read();
  const bool order_free = rq.want_std5 && !rq.want_pw && !rq.var_out && !rq.std_out && !rq.prod_out && !rq.first_out && !rq.last_out;
  AccGeom ag;
  // (a BOUND column's int64 sum keeps the sorted layout: gb.sum(c) is usually followed by gb.mean(c), which needs it -- Arrow's int64 mean
  //  is the pairwise sum of the doubles -- and the layout then serves both; min / max / count of a bound column do skip the sort)
  if (order_free && !(bound && rq.want_is) && !L.fused && !L.full)
    ag = acc_geometry(gb, vvalid != nullptr, (o.vmin ? kAccMin : 0u) | (o.vmax ? kAccMax : 0u) | (o.sum_i ? kAccSum : 0u) | (o.count ? kAccCnt : 0u), is_f, at);
  const bool use_acc = ag.ok;
  if (!use_acc && ((!L.fused && !L.full) || (!std_only && !L.full))) PDX_TRY(build_layout(gb, values, std_only, t, L, st));
  const bool use_fused = std_only && L.fused;
  Scratch s;
  std::string reducer, acc_plan;
  auto pack_validity = [&](uint8_t* bits, const uint8_t* ok_bytes) {
    if (!bits) return;
    if (!ok_bytes) hipMemsetAsync(bits, 0xFF, (size_t)((G + 7) / 8), st);
    else hipLaunchKernelGGL(k_pack_bytes, dim3(grid_for((G + 7) / 8, 256)), dim3(256), 0, st, ok_bytes, G, bits);
  };
  // the five standard kinds from one reduce into `oo`
  auto reduce_std = [&](const SegOut& oo, bool pw, bool mm, bool is, uint8_t* okb) -> int {
    if (use_acc) {
      reducer = "lds_acc";
      return reduce_acc(gb, ag, values, oo, okb, at, s, st, &acc_plan);
    }
    if (use_fused) return reduce_fused(gb, L, t, oo, pw, mm, is, okb, nullptr, s, st, &reducer);
    reducer = vvalid ? "seg_reduce_nullable" : "seg_reduce";
    return reduce_full(gb, L, L.vals_sorted, is_f, oo, pw, mm, is, L.out_index, okb, s, st);
  };
  std::string cache_note;
  if (rq.want_std5 && bound) {
    // a bound column keeps its per-group sum / count (/ min / max): sum(); mean(); count() as three calls cost one reduce
    const bool have_cnt = L.have_pw || L.have_count, have_isum = L.have_pw || L.have_is;
    const bool need_sumfam = (rq.want_pw && !L.have_pw) || (rq.want_is && !have_isum) || (o.count && !have_cnt), need_mm = rq.want_mm && !L.have_mm;
    if (use_acc && (need_sumfam || need_mm)) {
      // order-free kinds of a bound column: only what is asked for is computed and kept
      const bool need_is = rq.want_is && !have_isum, need_cnt = o.count && !have_cnt;
      if (need_cnt && !L.c_count) L.c_count = L.own<long long>((size_t)G);
      if (vvalid && !L.c_ok) L.c_ok = L.own<uint8_t>((size_t)G);
      if (need_is && !L.c_isum) L.c_isum = L.own<long long>((size_t)G);
      if (need_mm && !L.c_min) {
        L.c_min = L.own<uint64_t>((size_t)G);
        L.c_max = L.own<uint64_t>((size_t)G);
      }
      if ((need_cnt && !L.c_count) || (vvalid && !L.c_ok) || (need_is && !L.c_isum) || (need_mm && (!L.c_min || !L.c_max))) return PDX_OOM;
      SegOut c{};
      if (need_cnt) c.count = L.c_count;
      if (need_is) c.sum_i = L.c_isum;
      if (need_mm) {
        c.vmin = L.c_min;
        c.vmax = L.c_max;
      }
      PDX_TRY(reduce_std(c, false, need_mm, need_is, L.c_ok));
      L.have_count = L.have_count || need_cnt;
      L.have_is = L.have_is || need_is;
      L.have_mm = L.have_mm || need_mm;
      cache_note = " cache=fill";
    } else if (need_sumfam || need_mm) {
      if (!L.c_count) L.c_count = L.own<long long>((size_t)G);
      if (vvalid && !L.c_ok) L.c_ok = L.own<uint8_t>((size_t)G);
      if (need_sumfam && !L.c_sum) L.c_sum = L.own<double>((size_t)G);
      if (need_sumfam && !is_f && !L.c_isum) L.c_isum = L.own<long long>((size_t)G);
      if (need_mm && !L.c_min) {
        L.c_min = L.own<uint64_t>((size_t)G);
        L.c_max = L.own<uint64_t>((size_t)G);
      }
      if (!L.c_count || (vvalid && !L.c_ok) || (need_sumfam && (!L.c_sum || (!is_f && !L.c_isum))) || (need_mm && (!L.c_min || !L.c_max))) return PDX_OOM;
      SegOut c{};
      c.count = L.c_count;
      if (need_sumfam) {
        c.sum_f = L.c_sum;
        if (!is_f) c.sum_i = L.c_isum;
      }
      if (need_mm) {
        c.vmin = L.c_min;
        c.vmax = L.c_max;
      }
      PDX_TRY(reduce_std(c, need_sumfam, need_mm, need_sumfam && !is_f, L.c_ok));
      L.have_pw = L.have_pw || need_sumfam;
      L.have_mm = L.have_mm || need_mm;
      cache_note = " cache=fill";
    } else {
      cache_note = " cache=hit";
      reducer = "none";
    }
    const size_t gb8 = (size_t)G * 8;
    if (o.sum_f) PDX_HIP(hipMemcpyAsync(o.sum_f, L.c_sum, gb8, hipMemcpyDeviceToDevice, st));
    if (o.sum_i) PDX_HIP(hipMemcpyAsync(o.sum_i, L.c_isum, gb8, hipMemcpyDeviceToDevice, st));
    if (o.count) PDX_HIP(hipMemcpyAsync(o.count, L.c_count, gb8, hipMemcpyDeviceToDevice, st));
    if (o.vmin) PDX_HIP(hipMemcpyAsync(o.vmin, L.c_min, gb8, hipMemcpyDeviceToDevice, st));
    if (o.vmax) PDX_HIP(hipMemcpyAsync(o.vmax, L.c_max, gb8, hipMemcpyDeviceToDevice, st));
    if (o.mean) hipLaunchKernelGGL(k_mean_from_cache, dim3(grid_for(G, 256)), dim3(256), 0, st, L.c_sum, L.c_count, G, o.mean);
    for (int k = 0; k < nk; ++k)
      if (kinds[k] <= PDX_AGG_COUNT) pack_validity(static_cast<uint8_t*>(outs[k].validity), kinds[k] == PDX_AGG_COUNT ? nullptr : L.c_ok);
    PDX_LAUNCH_CHECK();
  } else if (rq.want_std5) {
    uint8_t* okb = vvalid ? s.get<uint8_t>((size_t)G) : nullptr;
    PDX_SCRATCH_CHECK(s);
    PDX_TRY(reduce_std(o, rq.want_pw, rq.want_mm, rq.want_is, okb));
    for (int k = 0; k < nk; ++k)
      if (kinds[k] <= PDX_AGG_COUNT) pack_validity(static_cast<uint8_t*>(outs[k].validity), kinds[k] == PDX_AGG_COUNT ? nullptr : okb);
    PDX_LAUNCH_CHECK();
  }
  if (rq.var_out || rq.std_out) {
    // Arrow's two passes: mean = pairwise sum / count, then the pairwise sum of (x - mean)^2 over the same valid runs
    double* m2 = s.get<double>((size_t)G);
    long long* cnt_g = s.get<long long>((size_t)G);
    uint8_t* ok2 = vvalid ? s.get<uint8_t>((size_t)G) : nullptr;
    uint8_t* ok1 = vvalid ? s.get<uint8_t>((size_t)G) : nullptr;
    double* mean_g = s.get<double>((size_t)G);
    PDX_SCRATCH_CHECK(s);
    SegOut o1{};
    o1.mean = mean_g;
    SegOut o2{};
    o2.sum_f = m2;
    o2.count = cnt_g;
    if (use_fused) {
      // on the same partially sorted rows: per-group mean (group-id order), then the squared deviations formed inside the kernel
      PDX_TRY(reduce_fused(gb, L, t, o1, true, false, false, ok1, nullptr, s, st, &reducer));
      PDX_TRY(reduce_fused(gb, L, t, o2, true, false, false, ok2, mean_g, s, st, nullptr));
    } else {
      if (reducer.empty()) reducer = vvalid ? "seg_reduce_nullable" : "seg_reduce";
      double* d = s.get<double>((size_t)n);
      PDX_SCRATCH_CHECK(s);
      PDX_TRY(reduce_full(gb, L, L.vals_sorted, is_f, o1, true, false, false, nullptr, ok1, s, st));  // segment order
      {
        PDX_PROFILE("seg_sqdev", st);
        const int grid = (int)std::min<int64_t>(ceil_div(G, 4), (int64_t)kCUs * 16);
        if (is_f) hipLaunchKernelGGL((k_seg_sqdev<double>), dim3(grid), dim3(256), 0, st, static_cast<const double*>(L.vals_sorted), L.seg_start, G, mean_g, d);
        else hipLaunchKernelGGL((k_seg_sqdev<long long>), dim3(grid), dim3(256), 0, st, static_cast<const long long*>(L.vals_sorted), L.seg_start, G, mean_g, d);
        PDX_LAUNCH_CHECK();
      }
      PDX_TRY(reduce_full(gb, L, d, true, o2, true, false, false, L.out_index, ok2, s, st));
    }
    hipLaunchKernelGGL(k_var_finish, dim3(grid_for(G, 256)), dim3(256), 0, st, m2, cnt_g, G, rq.var_out, rq.std_out);
    for (int k = 0; k < nk; ++k)
      if (kinds[k] == PDX_AGG_VARIANCE || kinds[k] == PDX_AGG_STDDEV) pack_validity(static_cast<uint8_t*>(outs[k].validity), ok2);
    PDX_LAUNCH_CHECK();
  }
  if (rq.prod_out || rq.first_out || rq.last_out) {
    uint8_t* pok = (vvalid && rq.prod_out) ? s.get<uint8_t>((size_t)G) : nullptr;
    uint8_t* fok = (vvalid && rq.first_out) ? s.get<uint8_t>((size_t)G) : nullptr;
    uint8_t* lok = (vvalid && rq.last_out) ? s.get<uint8_t>((size_t)G) : nullptr;
    PDX_SCRATCH_CHECK(s);
    {
      PDX_PROFILE("seg_product_first_last", st);
      const int pf_grid = (int)std::min<int64_t>(ceil_div(G, 4), (int64_t)kCUs * 16);
      if (is_f)
        hipLaunchKernelGGL((k_seg_product_first_last<double>), dim3(pf_grid), dim3(256), 0, st, static_cast<const double*>(L.vals_sorted), L.flag_keys,
                           L.row_valid, values->offset, L.seg_start, G, L.out_index, static_cast<double*>(rq.prod_out), pok, static_cast<double*>(rq.first_out),
                           fok, static_cast<double*>(rq.last_out), lok);
      else
        hipLaunchKernelGGL((k_seg_product_first_last<long long>), dim3(pf_grid), dim3(256), 0, st, static_cast<const long long*>(L.vals_sorted),
                           L.flag_keys, L.row_valid, values->offset, L.seg_start, G, L.out_index, static_cast<long long*>(rq.prod_out), pok,
                           static_cast<long long*>(rq.first_out), fok, static_cast<long long*>(rq.last_out), lok);
    }
    for (int k = 0; k < nk; ++k) {
      if (kinds[k] == PDX_AGG_PRODUCT) pack_validity(static_cast<uint8_t*>(outs[k].validity), pok);
      if (kinds[k] == PDX_AGG_FIRST) pack_validity(static_cast<uint8_t*>(outs[k].validity), fok);
      if (kinds[k] == PDX_AGG_LAST) pack_validity(static_cast<uint8_t*>(outs[k].validity), lok);
    }
    PDX_LAUNCH_CHECK();
    if (reducer.empty() || reducer == "none") reducer = "seg_product_first_last";
  }
  if (use_acc)
    gb->last_plan = (acc_plan.empty() ? std::string("slots=") + slots_name(gb) + " sort=none layout=none reducer=none" : acc_plan) + (bound ? " bound=1" : " bound=0") + cache_note;
  else
    gb->last_plan = (use_fused ? L.plan_fused : L.plan_full) + " reducer=" + reducer + (bound ? " bound=1" : " bound=0") + cache_note;
  if (bound && L.bytes != bytes_before) enforce_bind_limit(gb, &L);
  PDX_HIP(hipStreamSynchronize(st));  // outputs are valid on return; scratch and a local layout go back to the pool on exit
  return PDX_OK;
}


// adjustDatesAnchored + date_range (src/resample.cpp:85-178, src/core.cpp:308-331), tz == "": the first bin edge and the number of bins
// of an axis whose smallest / largest timestamps are tmin / tmax.  Host arithmetic only (also used by the sharded resample, where
// every rank must bin on the WHOLE axis' grid).
int pdx_resample_grid(int64_t tmin, int64_t tmax, int64_t freq_ns, int closed_right, int origin_type, int64_t origin_custom_ns, int64_t offset_ns,
                      int64_t* first_edge, int64_t* num_bins) {
  if (!first_edge || !num_bins) return fail(PDX_INVALID, "pdx_resample_grid: null output");
  if (freq_ns <= 0) return fail(PDX_INVALID, "FREQ must be positive");
  auto floor_div = [](long long a, long long b) { long long q = a / b, r = a % b; return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q; };
  const long long day = 86400000000000LL;
  long long first = tmin, last = tmax, origin = 0;
  switch (origin_type & ~PDX_ORIGIN_SHARD) {
    case PDX_ORIGIN_EPOCH: origin = 0; break;
    case PDX_ORIGIN_START_DAY: origin = floor_div(first, day) * day; break;
    case PDX_ORIGIN_START: origin = first; break;
    case PDX_ORIGIN_END: origin = last; break;
    case PDX_ORIGIN_END_DAY: origin = floor_div(last, day) * day; break;
    default: origin = origin_custom_ns; break;
  }
  origin += offset_ns;
  long long foffset = (first - origin) % freq_ns, loffset = (last - origin) % freq_ns;
  if (closed_right) {
    if (foffset > 0) first -= foffset; else first -= freq_ns;
    if (loffset > 0) last += freq_ns - loffset;
  } else {
    if (foffset > 0) first -= foffset;
    if (loffset > 0) last += freq_ns - loffset; else last += freq_ns;
  }
  if (first >= last) return fail(PDX_INVALID, "start date has to be less than end date");
  long long nedges = (last - first) / freq_ns + 1;  // date_range: first + k*freq <= last (src/core.cpp:308-331)
  long long last_edge = first + (nedges - 1) * freq_ns;
  if (tmin < first) return fail(PDX_INVALID, "Values falls before first bin");
  if (tmax > last_edge) return fail(PDX_INVALID, "Values falls after last bin");
  *first_edge = first;
  *num_bins = nedges - 1;
  return PDX_OK;
}

int pdx_resample_create(const pdx_column* ts, int64_t freq_ns, int closed_right, int label_right, int origin_type,
                        int64_t origin_custom_ns, int64_t offset_ns, void* stream, pdx_groupby** out) {
  PDX_TRY(check_column(ts, "pdx_resample_create"));
  if (!out) return fail(PDX_INVALID, "pdx_resample_create: null output");
  if (ts->dtype != PDX_TIMESTAMP_NS && ts->dtype != PDX_INT64) return fail(PDX_INVALID, "axis must be a TimestampArray");
  if (validity_or_null(ts)) return fail(PDX_NOT_IMPLEMENTED, "pdx_resample_create: null timestamps are not supported");
  if (freq_ns <= 0) return fail(PDX_INVALID, "FREQ must be positive");
  const int64_t n = ts->length;
  if (n > 0x7FFFFFFFll) return fail(PDX_NOT_IMPLEMENTED, "pdx_resample_create: more than 2^31-1 rows per call is not supported yet");
  hipStream_t st = as_stream(stream);
  std::unique_ptr<pdx_groupby> owner(new pdx_groupby());  // released into *out on success
  owner->stream = st;
  pdx_groupby* gb = owner.get();
  gb->mode = 1;
  gb->resample = true;
  gb->n = n;
  gb->key_dtype = PDX_TIMESTAMP_NS;
  *out = nullptr;
  if (n == 0) {
    *out = owner.release();
    return PDX_OK;
  }
  Scratch s;
  const long long* t = static_cast<const long long*>(ts->values) + ts->offset;
  unsigned int* bad = s.get<unsigned int>(1);
  if (s.failed) return PDX_OOM;
  hipMemsetAsync(bad, 0, sizeof(unsigned int), st);
  {
    PDX_PROFILE("resample_check_sorted", st);
    hipLaunchKernelGGL(k_check_sorted, dim3(grid_for(n, 256, 4)), dim3(256), 0, st, t, n, bad);
  }
  // sorted input: MinMax (src/resample.cpp:223) is the first and the last timestamp
  long long mn = 0, mx = 0;
  unsigned int hbad = 0;
  int rc = PDX_OK;
  {
    hipError_t e = hipMemcpyAsync(&hbad, bad, sizeof(hbad), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&mn, t, sizeof(mn), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&mx, t + (n - 1), sizeof(mx), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) rc = hip_fail(e, "pdx_resample_create");
  }
  if (rc != PDX_OK) return rc;
  if (hbad) return fail(PDX_INVALID, "pdx_resample_create: timestamps must be sorted ascending");
  const bool is_shard = (origin_type & PDX_ORIGIN_SHARD) != 0;
  origin_type &= ~PDX_ORIGIN_SHARD;
  int64_t first_edge = 0, num_bins = 0;
  PDX_TRY(pdx_resample_grid(mn, mx, freq_ns, closed_right, origin_type, origin_custom_ns, offset_ns, &first_edge, &num_bins));
  const long long first = first_edge, nbins = num_bins;
  if (n < nbins && !is_shard) return fail(PDX_INVALID, "upSampling is not implemented.");  // GroupInfo::upsampling, src/resample.h:14-17
  gb->bin = BinParams{t, first, freq_ns, 1.0 / (double)freq_ns, closed_right};
  gb->label_base = first + (label_right ? freq_ns : 0);
  // non-empty bins: boundaries where the bin index changes (timestamps are sorted)
  int64_t maxg = std::min<int64_t>(n, nbins);
  gb->seg_start = gb->own<uint32_t>((size_t)maxg + 1);
  gb->uniques = gb->own<int64_t>((size_t)maxg);
  gb->first_rows = gb->own<int64_t>((size_t)maxg);
  gb->unique_ok = gb->own<uint8_t>((size_t)maxg);
  if (!gb->seg_start || !gb->uniques || !gb->first_rows || !gb->unique_ok) return PDX_OOM;
  int64_t G = 0;
  {
    PDX_PROFILE("resample_bins", st);
    if (nbins * 16 <= n) {
      // many rows per bin: binary-search every edge (nbins * log n reads) instead of evaluating the bin of every row
      uint32_t* lb = s.get<uint32_t>((size_t)nbins + 1);
      if (s.failed) return PDX_OOM;
      hipLaunchKernelGGL(k_bin_lower_bounds, dim3(grid_for(nbins + 1, 256)), dim3(256), 0, st, gb->bin, n, (int64_t)nbins, mn, mx, lb);
      rc = compact_indices((int64_t)nbins, NonEmptyBinPred{lb}, NonEmptyBinEmit{lb, gb->label_base, freq_ns, gb->seg_start, gb->uniques, gb->first_rows},
                           &G, s, st);
    } else {
      rc = compact_indices(n, BinStartPred{gb->bin}, BinStartEmit{gb->bin, gb->label_base, gb->seg_start, gb->uniques, gb->first_rows}, &G, s, st);
    }
  }
  if (rc != PDX_OK) return rc;
  gb->G = G;
  hipLaunchKernelGGL(k_set_last, dim3(1), dim3(64), 0, st, gb->seg_start, G, (uint32_t)n);
  hipMemsetAsync(gb->unique_ok, 1, (size_t)G, st);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return hip_fail(e, "pdx_resample_create");
  *out = owner.release();
  return PDX_OK;
}

// DataFrame::downsample (reference src/dataframe.cpp:1265-1290): Ceil/FloorTemporal of the index, optionally one day less, then a
// GroupBy keyed on the rounded index.  On a sorted axis the rounded labels are (almost always) non-decreasing, so the groups are
// runs of equal labels: ONE pass over the timestamps rounds each row in registers, marks the run starts and proves the order --
// the rounded column is never written or hashed (8 B/row instead of 16 + the dictionary build).  Anything else (nulls, an unsorted
// axis, a calendar-origin ceil that steps back at an origin) goes through pdx_round_temporal + pdx_groupby_create as before.
int pdx_downsample_create(const pdx_column* ts, int64_t multiple, int unit, int ceil_mode, int week_starts_monday, int calendar_based_origin,
                          int64_t label_shift_ns, void* stream, pdx_groupby** out) {
  PDX_TRY(check_column(ts, "pdx_downsample_create"));
  if (!out) return fail(PDX_INVALID, "pdx_downsample_create: null output");
  if (ts->dtype != PDX_TIMESTAMP_NS) return fail(PDX_INVALID, "pdx_downsample_create: index must be PDX_TIMESTAMP_NS");
  RoundParams q{};
  int mode = 0;
  PDX_TRY(make_round_params(multiple, unit, week_starts_monday, calendar_based_origin, &q, &mode, "pdx_downsample_create"));
  const int64_t n = ts->length;
  if (n > 0x7FFFFFFFll) return fail(PDX_NOT_IMPLEMENTED, "pdx_downsample_create: more than 2^31-1 rows per call is not supported yet");
  hipStream_t st = as_stream(stream);
  *out = nullptr;
  const bool runs_env = [] { const char* e = getenv("PDX_GROUPBY_SORTED"); return !(e && e[0] == '0'); }();
  if (runs_env && n >= 1 && !validity_or_null(ts)) {
    std::unique_ptr<pdx_groupby> owner(new pdx_groupby());
    owner->stream = st;
    pdx_groupby* gb = owner.get();
    gb->n = n;
    gb->key_dtype = PDX_TIMESTAMP_NS;
    Scratch s;
    const long long* t = static_cast<const long long*>(ts->values) + ts->offset;
    bool done = false;
    int rc = PDX_OK;
#define DS_RUNS(M)                                                                                      \
  rc = ceil_mode ? build_label_runs(gb, n, RoundLabel<M, true>{t, q}, label_shift_ns, s, st, &done)     \
                 : build_label_runs(gb, n, RoundLabel<M, false>{t, q}, label_shift_ns, s, st, &done)
    switch (mode) {
      case 0: DS_RUNS(0); break;
      case 1: DS_RUNS(1); break;
      case 2: DS_RUNS(2); break;
      case 3: DS_RUNS(3); break;
      case 4: DS_RUNS(4); break;
      case 5: DS_RUNS(5); break;
      case 6: DS_RUNS(6); break;
      case 7: DS_RUNS(7); break;
      case 8: DS_RUNS(8); break;
      default: DS_RUNS(9); break;
    }
#undef DS_RUNS
    if (rc != PDX_OK) return rc;
    if (done) {
      *out = owner.release();
      return PDX_OK;
    }
  }
  // the general way: the rounded column, then a dictionary of it (first-occurrence order; nulls form their own group)
  const bool has_nulls = validity_or_null(ts) != nullptr;
  void* binned_vals = pool_alloc((size_t)(n ? n : 1) * sizeof(int64_t));
  void* binned_valid = has_nulls ? pool_alloc((size_t)(n + 7) / 8 + 8) : nullptr;
  if (!binned_vals || (has_nulls && !binned_valid)) {
    if (binned_vals) pool_free(binned_vals);
    if (binned_valid) pool_free(binned_valid);
    return PDX_OOM;
  }
  pdx_mut_column mb{};
  mb.dtype = PDX_TIMESTAMP_NS;
  mb.length = n;
  mb.values = binned_vals;
  mb.validity = binned_valid;
  int rc = pdx_round_temporal(ceil_mode, ts, multiple, unit, week_starts_monday, calendar_based_origin, &mb, stream);
  if (rc == PDX_OK) {
    pdx_column cb{};
    cb.dtype = PDX_TIMESTAMP_NS;
    cb.length = n;
    cb.values = binned_vals;
    cb.validity = binned_valid;
    cb.offset = 0;
    cb.null_count = has_nulls ? -1 : 0;
    rc = pdx_groupby_create(&cb, stream, out);
  }
  if (rc == PDX_OK && *out && label_shift_ns != 0 && (*out)->G > 0) {
    hipLaunchKernelGGL(k_shift_labels, dim3(grid_for((*out)->G, 256)), dim3(256), 0, st, (*out)->uniques, (*out)->G, (long long)label_shift_ns);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = fail(PDX_DEVICE, "pdx_downsample_create: label shift failed");
  }
  {
    StreamNote note(st);
    pool_free(binned_vals);
    if (binned_valid) pool_free(binned_valid);
  }
  if (rc != PDX_OK && *out) {
    pdx_groupby_destroy(*out);
    *out = nullptr;
  }
  return rc;
}

int pdx_resample_row_labels(pdx_groupby* gb, int64_t* out_labels, void* stream) {
  if (!gb || !out_labels) return fail(PDX_INVALID, "pdx_resample_row_labels: null argument");
  if (!gb->resample) return fail(PDX_INVALID, "pdx_resample_row_labels: handle was not created by pdx_resample_create");
  hipStream_t st = as_stream(stream);
  gb->use_on(st);  // ordered behind the handle's creation; frees of its blocks are ordered behind this stream
  if (gb->n) hipLaunchKernelGGL(k_row_labels, dim3(grid_for(gb->n, 256, 4)), dim3(256), 0, st, gb->bin, gb->label_base, gb->n, out_labels);
  PDX_LAUNCH_CHECK();
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}

}  // extern "C"

#include "gb_partial_tree.hpp"

#ifdef PDX_FLR_TIMING
// diagnostic build only: read (and clear) the dense fused kernel's per-phase cycle sums
extern "C" int pdx_debug_flr_cycles(unsigned long long* out24, int reset) {
  if (hipMemcpyFromSymbol(out24, HIP_SYMBOL(pdx::g_flr_cycles), sizeof(unsigned long long) * 24) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[24] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(pdx::g_flr_cycles), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif
