// groupby.hip -- hash group-by with sum/mean/min/max/count, and the resample front-end, for gfx950.
//
// Replaces (reference file:line)
//   GroupBy::makeGroups  src/dataframe.cpp:1571-1600   Grouper::Make/Consume/GetUniques  -> pdx_groupby_create
//   processIndex/processEach (MakeGroupings + ApplyGroupings of every column, 1539-1569)  -> deferred, per aggregated
//                                                                                            column, to pdx_groupby_agg
//   GROUPBY_AGG / GROUPBY_NUMERIC_AGG  src/pd_core_macros.h:5-147 (one CallFunction per group)  -> pdx_groupby_agg
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
//      ranks each 2560-row tile by them in LDS and replays Arrow's leaf / binary-counter recurrence with one lane per group.
//      Dense slots + values without nulls: narrowing sort (the key shrinks 4 -> 2 -> 1 byte as digits are consumed; run and
//      group starts come from the scatter offsets, k_level_starts).
//   4. classic reducers on fully sorted values (skewed keys, small inputs, resample, product/first/last): k_seg_reduce (one wave
//      per group, 16-value leaves + shuffle tree + counter), k_seg_reduce_mid (batches of short groups per wave),
//      k_seg_reduce_sub + k_seg_combine_big (many waves per long group), k_seg_reduce_nullable.
// All fp64 sums reproduce Arrow's pairwise summation bit for bit.  Algorithmic bytes: 16 B/row (8 key + 8 value).
#include <stdlib.h>
#include <algorithm>
#include <memory>
#include <vector>
#include "compact.hpp"
#include "minmax.hpp"
#include "pairwise.hpp"
#include "radix_sort.hpp"
#include "scan.hpp"

namespace pdx {

struct Slot {
  long long key;
  unsigned int first;  // first row with this key (0xFFFFFFFF = slot never used)
  unsigned int gid;    // dense group id in first-occurrence order
};
static_assert(sizeof(Slot) == 16, "slot layout");
constexpr long long kEmptyKey = (long long)0x8000000000000000ull;  // INT64_MIN is routed to a dedicated slot
constexpr unsigned int kNoRow = 0xFFFFFFFFu;

struct HashCtl {
  unsigned int inserted;
  unsigned int overflow;
  unsigned long long rows_seen;     // LDS build only: rows consumed before the buckets finished / gave up ...
  unsigned long long est_distinct;  // ... and the distinct keys among exactly those rows (cardinality estimate)
};

__global__ void k_table_init(Slot* __restrict__ table, int64_t nslots) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += stride) {
    table[i].key = kEmptyKey;
    table[i].first = kNoRow;
    table[i].gid = kNoRow;
  }
}

// Lock-free insert-or-find.  A stale (cached) read of an EMPTY key only costs a CAS: the CAS result is authoritative.
__global__ void __launch_bounds__(256) k_hash_insert(const long long* __restrict__ keys, const uint8_t* __restrict__ valid,
                                                     int64_t off, int64_t n, Slot* table, unsigned int cap, unsigned int limit,
                                                     uint32_t* __restrict__ slot_of_row, HashCtl* ctl) {
  const unsigned int mask = cap - 1;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    // (the overflow flag is polled only on the insert / long-probe path: a per-row poll of one address serialises on one L2 channel)
    unsigned int s;
    long long key = keys[i];
    if (valid && !bit_get(valid, off + i)) {
      s = cap;  // the null key is its own group
    } else if (key == kEmptyKey) {
      s = cap + 1;
    } else {
      unsigned int h = (unsigned int)splitmix64((uint64_t)key) & mask;
      unsigned int probes = 0;
      for (;;) {
        long long cur = table[h].key;
        if (cur == key) { s = h; break; }
        if (cur == kEmptyKey) {
          if (__hip_atomic_load(&ctl->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
          unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&table[h].key), (unsigned long long)kEmptyKey,
                                             (unsigned long long)key);
          if (old == (unsigned long long)kEmptyKey) {
            unsigned int c = atomicAdd(&ctl->inserted, 1u);
            if (c >= limit) atomicExch(&ctl->overflow, 1u);
            s = h;
            break;
          }
          if (old == (unsigned long long)key) { s = h; break; }
        }
        h = (h + 1) & mask;
        ++probes;
        if ((probes & 63) == 0 && __hip_atomic_load(&ctl->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        if (probes > cap) {
          atomicExch(&ctl->overflow, 1u);
          s = cap;
          break;
        }
      }
    }
    if ((unsigned int)i < table[s].first) atomicMin(&table[s].first, (unsigned int)i);
    slot_of_row[i] = s;
  }
}

// ---- partitioned hash build (general keys).  Rows are first partitioned (stably) by the low kPartBits of a 32-bit key hash:
// that pass IS the first LSD pass of the later sort by slot, because the logical slot id is (index inside the bucket's table
// region << kPartBits) | bucket.  All rows of a bucket probe one contiguous 1/256 region of the table, and tiles are processed in
// bucket order, so the active part of the table (a few hundred KB) stays in every XCD's L2 instead of costing one random
// 128-byte line from the Infinity Cache per row.
constexpr int kPartBits = 8;
// 32-bit hash of a key as the partitioned build sees it; the two keys with dedicated slots get fixed hashes whose low bits
// equal the low bits of those slots' logical ids (cap -> 0, cap + 1 -> 1)
__device__ __forceinline__ uint32_t key_hash32(long long k, bool is_null) {
  uint32_t h = (uint32_t)(splitmix64((uint64_t)k) >> 32);
  if (k == kEmptyKey) h = 1;
  if (is_null) h = 0;
  return h;
}
// bucket (low kPartBits of the hash) of every row, one byte per row, + the per-tile bucket histogram of the partition pass
__global__ void __launch_bounds__(kSortBlock) k_hash_bucket_hist(const long long* __restrict__ keys, const uint8_t* __restrict__ valid, int64_t off,
                                                                 int64_t n, uint8_t* __restrict__ bucket, uint32_t* __restrict__ hist) {
  constexpr int R = 1 << kPartBits;
  __shared__ uint32_t h[R];
  for (int d = threadIdx.x; d < R; d += kSortBlock) h[d] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortTile;
  long long k[kSortItems];
#pragma unroll
  for (int u = 0; u < kSortItems; ++u) {
    int64_t i = base + u * kSortBlock + threadIdx.x;
    k[u] = i < n ? keys[i] : 0;
  }
#pragma unroll
  for (int u = 0; u < kSortItems; ++u) {
    int64_t i = base + u * kSortBlock + threadIdx.x;
    if (i >= n) continue;
    const uint32_t b = key_hash32(k[u], valid && !bit_get(valid, off + i)) & (R - 1);
    bucket[i] = (uint8_t)b;
    atomicAdd(&h[b], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < R; d += kSortBlock) hist[(int64_t)blockIdx.x * R + d] = h[d];
}
// second partition level (very many groups: the buckets are split until a bucket's table fits in LDS): digit = hash bits
// [shift, shift + BITS) of the rows in their CURRENT (first-level) order, + the per-tile histogram of that digit
template <int BITS>
__global__ void __launch_bounds__(kSortBlock) k_hash_digit_hist(const long long* __restrict__ keys_cur, const uint32_t* __restrict__ rows_cur, int64_t n,
                                                                int shift, uint8_t* __restrict__ digit, uint32_t* __restrict__ hist) {
  constexpr int R = 1 << BITS;
  __shared__ uint32_t h[R];
  for (int d = threadIdx.x; d < R; d += kSortBlock) h[d] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortTile;
#pragma unroll
  for (int u = 0; u < kSortItems; ++u) {
    int64_t i = base + u * kSortBlock + threadIdx.x;
    if (i >= n) continue;
    const uint32_t d = (key_hash32(keys_cur[i], rows_cur[i] >> 31) >> shift) & (R - 1);
    digit[i] = (uint8_t)d;
    atomicAdd(&h[d], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < R; d += kSortBlock) hist[(int64_t)blockIdx.x * R + d] = h[d];
}
// start of every bucket in the final partitioned order (ascending low `pb` hash bits): lower bounds by binary search
__global__ void k_bucket_starts(const long long* __restrict__ keys_part, const uint32_t* __restrict__ rows_part, int64_t n, unsigned int pb,
                                uint32_t* __restrict__ starts) {
  const int64_t nb = (int64_t)1 << pb, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < nb; g += stride) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      const unsigned int b = key_hash32(keys_part[mid], rows_part[mid] >> 31) & (unsigned int)(nb - 1);
      if ((int64_t)b < g) lo = mid + 1;
      else hi = mid;
    }
    starts[g] = (uint32_t)lo;
  }
}
constexpr int kProbeTiles = 4;
// inputs in partitioned order; rows carry the null flag in bit 31.  U rows per thread are kept in flight: the stream loads and
// the first table probe of all U rows are issued before any of them is consumed.
template <int U>
__global__ void __launch_bounds__(256) k_hash_probe_part(const long long* __restrict__ keys_part, const uint32_t* __restrict__ rows_part,
                                                         int64_t n, Slot* table, unsigned int cap,
                                                         unsigned int region, unsigned int limit, uint32_t* __restrict__ slot_part,
                                                         HashCtl* ctl, unsigned int sweep_shift, unsigned int sweep, unsigned int pb) {
  // One contiguous run of kProbeTiles*U*256 partition-ordered rows per workgroup, runs dispatched in order: the workgroups
  // resident at any moment work on one or two neighbouring buckets, so a table far larger than the L2 is probed a few MB at a time.
  // Regions beyond ~2 MB fall out of the 4 MB L2 of an XCD and the build collapses (measured: 26 ms at 2 MB regions, 1.4 s at
  // 4 MB -- every probe and atomic goes to memory), so such tables are built in SWEEPS: sweep j handles only the rows whose home
  // slot lies in window j (2^sweep_shift slots) of their region; every sweep re-streams the rows but probes a 1 MB window.
  // Insertions are counted per thread and flushed once per wave: with tens of millions of groups a per-insert atomic on the one
  // counter word serialises the whole build (measured 0.9 s for 1e8 groups).
  const unsigned int rmask = region - 1;
  const int64_t stride = blockDim.x;
  unsigned int my_inserts = 0;
  bool dead = false;
  for (int t = 0; t < kProbeTiles && !dead; ++t) {
    const int64_t p0 = ((int64_t)blockIdx.x * kProbeTiles + t) * blockDim.x * U + threadIdx.x;
    if (p0 - threadIdx.x >= n) break;
    // a failed attempt must end quickly: once the load limit is passed (or a chain got too long) nobody starts another tile
    if (__hip_atomic_load(&ctl->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    unsigned int row[U], h[U], phys[U], idx[U];
    long long key[U], cur[U];
    bool act[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t p = p0 + u * stride;
      act[u] = p < n;
      row[u] = act[u] ? rows_part[p] : 0u;
      key[u] = act[u] ? keys_part[p] : 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      h[u] = key_hash32(key[u], row[u] >> 31);
      const unsigned int b = h[u] & ((1u << pb) - 1);
      idx[u] = (h[u] >> pb) & rmask;
      phys[u] = b * region + idx[u];
      unsigned int win = idx[u] >> sweep_shift;
      if (row[u] >> 31) { phys[u] = cap; win = 0; }
      else if (key[u] == kEmptyKey) { phys[u] = cap + 1; win = 0; }
      act[u] = act[u] && win == sweep;
      cur[u] = act[u] ? table[phys[u]].key : 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!act[u] || dead) continue;
      const bool special = (row[u] >> 31) || key[u] == kEmptyKey;
      const unsigned int r = row[u] & 0x7FFFFFFFu;
      unsigned int logical;
      if (special) {
        logical = phys[u];
      } else {
        const unsigned int b = h[u] & ((1u << pb) - 1), base = b * region;
        unsigned int probes = 0;
        long long c = cur[u];
        for (;;) {
          if (c == key[u]) break;
          if (c == kEmptyKey) {
            unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&table[base + idx[u]].key), (unsigned long long)kEmptyKey,
                                               (unsigned long long)key[u]);
            if (old == (unsigned long long)kEmptyKey) {
              ++my_inserts;
              break;
            }
            if (old == (unsigned long long)key[u]) break;
          }
          idx[u] = (idx[u] + 1) & rmask;
          if ((++probes & 63) == 0) {  // long chain: this bucket's region is (nearly) full, or another wave already gave up
            if (probes > region || probes >= 4096 || __hip_atomic_load(&ctl->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
              atomicExch(&ctl->overflow, 1u);
              dead = true;
              break;
            }
          }
          c = table[base + idx[u]].key;
        }
        if (dead) continue;
        phys[u] = base + idx[u];
        logical = (idx[u] << pb) | b;
      }
      if (r < table[phys[u]].first) atomicMin(&table[phys[u]].first, r);
      slot_part[p0 + u * stride] = logical;
    }
  }
  // one counter update per wave (divergent exits above are re-converged here)
  unsigned int wave_inserts = my_inserts;
  for (int d = 32; d >= 1; d >>= 1) wave_inserts += __shfl_xor(wave_inserts, d, 64);
  if ((threadIdx.x & 63) == 0 && wave_inserts) {
    unsigned int before = atomicAdd(&ctl->inserted, wave_inserts);
    if (before + wave_inserts > limit) atomicExch(&ctl->overflow, 1u);
  }
}
// LDS-resident build: one workgroup per bucket keeps the bucket's whole table region (<= 8192 keys + first rows = 96 KB) in LDS,
// streams the bucket's rows once and writes the region back.  Random probes hit LDS banks instead of the L2/TA path, which
// tops out near 70 G random accesses/s chip-wide however local the table is (measured: profiles/ notes in DESIGN.md).
constexpr int kLdsRegionMax = 8192;
constexpr int kProbeBlock = 1024;
__global__ void __launch_bounds__(kProbeBlock) k_hash_probe_lds(const long long* __restrict__ keys_part, const uint32_t* __restrict__ rows_part,
                                                                const uint32_t* __restrict__ bucket_off,
                                                                int64_t n, Slot* table, unsigned int cap, unsigned int region,
                                                                uint32_t* __restrict__ slot_part, HashCtl* ctl, unsigned int pb, int64_t head_rows) {
  __shared__ unsigned long long lkeys[kLdsRegionMax];
  __shared__ unsigned int lfirst[kLdsRegionMax];
  __shared__ unsigned int linserted;
  __shared__ unsigned int lspecial[2];
  const int tid = threadIdx.x;
  const unsigned int b = blockIdx.x;
  const unsigned int rmask = region - 1;
  const int64_t start = bucket_off[b];
  int64_t end = (b + 1 < (1u << pb)) ? (int64_t)bucket_off[b + 1] : n;
  if (end - start > head_rows) end = start + head_rows;  // an overlong (skewed) bucket: the rest goes to k_hash_probe_lds_tail
  for (int i = tid; i < (int)region; i += kProbeBlock) {
    lkeys[i] = (unsigned long long)kEmptyKey;
    lfirst[i] = kNoRow;
  }
  if (tid == 0) {
    linserted = 0;
    lspecial[0] = lspecial[1] = kNoRow;
  }
  __syncthreads();
  constexpr int U = 4;
  const unsigned int dense_limit = region - (region >> 2);  // 75 % full: give up early, the host retries with a larger table
  bool sampled = false;
  // (the trip count is uniform over the workgroup -- rows are masked by act[] -- so the barrier after the first trip is safe)
  // The NEXT trip's rows are requested before this trip's probes: the probe chain of a row is a string of dependent LDS round
  // trips with little to issue in between, so with load -> wait -> probe per trip the waves spent 70 % of their cycles waiting
  // (SQ_WAIT_ANY) with the memory pipe idle half of the time.
  unsigned int nrow[U];
  long long nkey[U];
  auto request = [&](int64_t base0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = base0 + tid + (int64_t)u * kProbeBlock;
      nrow[u] = p < end ? rows_part[p] : 0u;
      nkey[u] = p < end ? keys_part[p] : 0;
    }
  };
  request(start);
  for (int64_t base0 = start; base0 < end; base0 += (int64_t)U * kProbeBlock) {
    const int64_t p0 = base0 + tid;
    if (linserted > dense_limit) {  // (LDS word, read by every thread each iteration: a handful of cycles)
      if (tid == 0) atomicExch(&ctl->overflow, 1u);
      break;
    }
    unsigned int row[U], h[U];
    long long key[U];
    bool act[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      act[u] = p0 + (int64_t)u * kProbeBlock < end;
      row[u] = nrow[u];
      key[u] = nkey[u];
    }
    if (base0 + (int64_t)U * kProbeBlock < end) request(base0 + (int64_t)U * kProbeBlock);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!act[u]) continue;
      h[u] = key_hash32(key[u], row[u] >> 31);
      const unsigned int r = row[u] & 0x7FFFFFFFu;
      unsigned int logical;
      if ((row[u] >> 31) || key[u] == kEmptyKey) {  // null key / INT64_MIN key: dedicated global slots
        // (their first row is tracked in LDS and flushed once: a column that is half null would otherwise send 5e8 atomics to
        //  one address -- measured 7.5 s)
        const unsigned int sp = (row[u] >> 31) ? cap : cap + 1;
        if (r < lspecial[sp - cap]) atomicMin(&lspecial[sp - cap], r);
        logical = sp;
      } else {
        unsigned int idx = (h[u] >> pb) & rmask, probes = 0;
        for (;;) {
          unsigned long long cur = lkeys[idx];
          if (cur == (unsigned long long)key[u]) break;
          if (cur == (unsigned long long)kEmptyKey) {
            unsigned long long old = atomicCAS(&lkeys[idx], (unsigned long long)kEmptyKey, (unsigned long long)key[u]);
            if (old == (unsigned long long)kEmptyKey) {
              atomicAdd(&linserted, 1u);
              break;
            }
            if (old == (unsigned long long)key[u]) break;
          }
          idx = (idx + 1) & rmask;
          if (++probes > 512) {  // pathologically long probe chain: the host retries with a larger table (L2 path)
            atomicExch(&ctl->overflow, 1u);
            break;
          }
        }
        if (r < lfirst[idx]) atomicMin(&lfirst[idx], r);
        logical = (idx << pb) | b;
      }
      slot_part[p0 + (int64_t)u * kProbeBlock] = logical;
    }
    if (!sampled) {
      // cardinality sample: after the bucket's first U*kProbeBlock rows every thread has inserted its rows, so (rows, distinct)
      // is an exact pair (the table is at most half full: no saturation)
      sampled = true;
      __syncthreads();
      if (tid == 0) {
        const int64_t seen = end - start < (int64_t)U * kProbeBlock ? end - start : (int64_t)U * kProbeBlock;
        atomicAdd(&ctl->rows_seen, (unsigned long long)seen);
        atomicAdd(&ctl->est_distinct, (unsigned long long)linserted);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < (int)region; i += kProbeBlock) {
    Slot sl;
    sl.key = (long long)lkeys[i];
    sl.first = lfirst[i];
    sl.gid = kNoRow;
    table[(int64_t)b * region + i] = sl;
  }
  if (tid == 0 && linserted) atomicAdd(&ctl->inserted, linserted);
  if (tid < 2 && lspecial[tid] != kNoRow) atomicMin(&table[cap + tid].first, lspecial[tid]);
}

// Skewed buckets (a hot key, or half of the keys null: all of those rows share one bucket): the workgroup above only builds the
// region from the bucket's first `head_rows` rows; the rest of the bucket is cut into chunks, one workgroup each.  A chunk's
// workgroup copies the region's keys into LDS (read-only snapshot) and resolves its rows there; a key the snapshot does not hold
// continues its probe chain in the memory-side region (CAS insert, the snapshot is a subset of it and keys never move), where
// its first row is also kept.  Keys found in the snapshot were inserted by the head rows, which precede every tail row of the
// bucket (the partition is stable), so their first row is already final.
struct TailChunk {
  uint32_t bucket, begin, end;
};
constexpr int kTailChunkRows = 1 << 17;
__global__ void __launch_bounds__(kProbeBlock) k_hash_probe_lds_tail(const long long* __restrict__ keys_part, const uint32_t* __restrict__ rows_part,
                                                                     const TailChunk* __restrict__ chunks, Slot* table, unsigned int cap,
                                                                     unsigned int region, uint32_t* __restrict__ slot_part, HashCtl* ctl, unsigned int pb) {
  __shared__ unsigned long long lkeys[kLdsRegionMax];
  __shared__ unsigned int linserted;
  __shared__ unsigned int lspecial[2];
  const int tid = threadIdx.x;
  const TailChunk ch = chunks[blockIdx.x];
  const unsigned int b = ch.bucket, rmask = region - 1;
  // the head attempt already failed: leave (ONE thread reads the flag -- it can change under us, and a workgroup that splits over
  // it would leave some waves at the barriers below forever)
  __shared__ unsigned int gave_up;
  if (tid == 0) gave_up = __hip_atomic_load(&ctl->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (gave_up) return;
  Slot* reg = table + (int64_t)b * region;
  for (int i = tid; i < (int)region; i += kProbeBlock) lkeys[i] = (unsigned long long)reg[i].key;
  if (tid == 0) {
    linserted = 0;
    lspecial[0] = lspecial[1] = kNoRow;
  }
  __syncthreads();
  constexpr int U = 4;
  for (int64_t base0 = ch.begin; base0 < (int64_t)ch.end; base0 += (int64_t)U * kProbeBlock) {
    const int64_t p0 = base0 + tid;
    unsigned int row[U];
    long long key[U];
    bool act[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t p = p0 + (int64_t)u * kProbeBlock;
      act[u] = p < (int64_t)ch.end;
      row[u] = act[u] ? rows_part[p] : 0u;
      key[u] = act[u] ? keys_part[p] : 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!act[u]) continue;
      const unsigned int r = row[u] & 0x7FFFFFFFu;
      unsigned int logical;
      if ((row[u] >> 31) || key[u] == kEmptyKey) {
        const unsigned int sp = (row[u] >> 31) ? cap : cap + 1;
        if (r < lspecial[sp - cap]) atomicMin(&lspecial[sp - cap], r);
        logical = sp;
      } else {
        unsigned int idx = (key_hash32(key[u], false) >> pb) & rmask, probes = 0;
        bool found = false, dead = false;
        for (;;) {  // the snapshot
          unsigned long long cur = lkeys[idx];
          if (cur == (unsigned long long)key[u]) { found = true; break; }
          if (cur == (unsigned long long)kEmptyKey) break;
          idx = (idx + 1) & rmask;
          if (++probes > region) { dead = true; break; }
        }
        if (!found && !dead) {  // memory side, from the slot the snapshot had empty
          for (;;) {
            unsigned long long cur = __hip_atomic_load(reinterpret_cast<unsigned long long*>(&reg[idx].key), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == (unsigned long long)key[u]) break;
            if (cur == (unsigned long long)kEmptyKey) {
              unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&reg[idx].key), (unsigned long long)kEmptyKey, (unsigned long long)key[u]);
              if (old == (unsigned long long)kEmptyKey) {
                atomicAdd(&linserted, 1u);
                break;
              }
              if (old == (unsigned long long)key[u]) break;
            }
            idx = (idx + 1) & rmask;
            if (++probes > region) { dead = true; break; }
          }
          if (!dead && r < __hip_atomic_load(&reg[idx].first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&reg[idx].first, r);
        }
        if (dead) {  // the region is full: the host retries with a larger table
          atomicExch(&ctl->overflow, 1u);
          continue;
        }
        logical = (idx << pb) | b;
      }
      slot_part[p0 + (int64_t)u * kProbeBlock] = logical;
    }
  }
  __syncthreads();
  if (tid == 0 && linserted) atomicAdd(&ctl->inserted, linserted);
  if (tid < 2 && lspecial[tid] != kNoRow) atomicMin(&table[cap + tid].first, lspecial[tid]);
}

__device__ __forceinline__ int64_t phys_slot(int64_t logical, unsigned int region, unsigned int cap) {
  if (region == 0 || logical >= (int64_t)cap) return logical;
  const int pb = (__ffs((int)cap) - 1) - (__ffs((int)region) - 1);  // cap = region << pb, both powers of two
  return (logical & ((1 << pb) - 1)) * (int64_t)region + (logical >> pb);
}
// row-order views from the partitioned arrays (on demand: group ids / mapped ids)
__global__ void k_part_row_gids(const uint32_t* __restrict__ gid_of_slot, const uint32_t* __restrict__ slot_part,
                                const uint32_t* __restrict__ rows_part, int64_t n, const int64_t* __restrict__ map, uint32_t* __restrict__ out32,
                                int64_t* __restrict__ out64) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
    uint32_t g = gid_of_slot[slot_part[p]];
    uint32_t row = rows_part[p] & 0x7FFFFFFFu;
    if (out32) out32[row] = g;
    if (out64) out64[row] = map[g];
  }
}
__global__ void k_flag_keys_part(const uint32_t* __restrict__ slot_part, const uint32_t* __restrict__ rows_part, const uint8_t* __restrict__ valid,
                                 int64_t off, int64_t n, uint32_t* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride)
    out[p] = slot_part[p] | (bit_get(valid, off + (int64_t)(rows_part[p] & 0x7FFFFFFFu)) ? 0u : 0x80000000u);
}

// Dense-domain fast path: when the valid keys span a small integer range the slot is key - min (no table, no probing):
// the only per-row memory access besides the streams is first[slot] (4 B, range-sized table that stays cache resident).
// Residue form (mask != 0): slot = key & mask.  Any window of <= mask + 1 consecutive integers has distinct residues, so this
// is the same perfect hash up to a rotation -- and it needs no minimum, which lets the build run in the SAME pass that
// computes the exact min/max (speculating on the width of the window; verified afterwards).
__device__ __forceinline__ unsigned int dense_slot_of(long long k, long long mn, unsigned int mask) {
  return mask ? ((unsigned int)(unsigned long long)k & mask) : (unsigned int)((unsigned long long)k - (unsigned long long)mn);
}
struct KeyRange {
  long long vmin, vmax;
  int any, pad;
};
// min/max of <= 65536 evenly spaced valid keys (64 workgroups, one sample per thread, one KeyRange per workgroup): the guess for
// the width of the key window
__global__ void __launch_bounds__(1024) k_sample_key_range(const long long* __restrict__ keys, const uint8_t* __restrict__ valid, int64_t off,
                                                           int64_t n, KeyRange* __restrict__ out) {
  __shared__ long long smn[16], smx[16];
  __shared__ int sany[16];
  const int64_t nsamp = n < 65536 ? n : 65536;
  long long mn = 0x7FFFFFFFFFFFFFFFll, mx = (long long)0x8000000000000000ull;
  int any = 0;
  {
    const int64_t j = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    int64_t i = j < nsamp ? (int64_t)((unsigned __int128)j * (unsigned __int128)n / (unsigned __int128)nsamp) : n;
    if (i < n && (!valid || bit_get(valid, off + i))) {
      long long k = keys[i];
      mn = k < mn ? k : mn;
      mx = k > mx ? k : mx;
      any = 1;
    }
  }
  for (int d = 32; d >= 1; d >>= 1) {
    long long a = __shfl_xor(mn, d, 64), b = __shfl_xor(mx, d, 64);
    int c = __shfl_xor(any, d, 64);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
    any |= c;
  }
  if ((threadIdx.x & 63) == 0) {
    smn[threadIdx.x >> 6] = mn;
    smx[threadIdx.x >> 6] = mx;
    sany[threadIdx.x >> 6] = any;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) {
      mn = smn[w] < mn ? smn[w] : mn;
      mx = smx[w] > mx ? smx[w] : mx;
      any |= sany[w];
    }
    out[blockIdx.x].vmin = mn;
    out[blockIdx.x].vmax = mx;
    out[blockIdx.x].any = any;
  }
}
__global__ void __launch_bounds__(256) k_dense_slots(const long long* __restrict__ keys, const uint8_t* __restrict__ valid, int64_t off,
                                                     int64_t n, long long mn, unsigned int mask, unsigned int range, unsigned int* first,
                                                     uint32_t* __restrict__ slot_of_row) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    unsigned int s = range;  // the null key's slot
    if (!valid || bit_get(valid, off + i)) s = dense_slot_of(keys[i], mn, mask);
    slot_of_row[i] = s;
    // only the first lane of every slot present in the wave needs the atomic (lanes hold ascending rows): with a handful of
    // distinct keys tens of thousands of in-flight atomicMin's on one word otherwise serialise in the L2 (measured 5.8 ms)
    const bool want = (unsigned int)i < first[s];
    const int lane = threadIdx.x & 63;
    uint64_t rem = __ballot(want);
    for (int rounds = 0; rem && rounds < 4; ++rounds) {
      const int leader = __ffsll((unsigned long long)rem) - 1;
      const unsigned int sl = (unsigned int)__shfl((int)s, leader, 64);
      const uint64_t grp = __ballot(want && s == sl) & rem;
      if (lane == leader) atomicMin(&first[s], (unsigned int)i);
      rem &= ~grp;
    }
    if ((rem >> lane) & 1) atomicMin(&first[s], (unsigned int)i);
  }
}

// Rows [row0, n) of the dense path once a prefix has been processed by k_dense_slots: `seen` has one bit per slot that already
// has a first row in the prefix.  Every row here is later than every prefix row, so a set bit means "not a first occurrence":
// the common case touches only the (cache-resident) bitmap instead of first[].
__global__ void k_seen_bitmap(const unsigned int* __restrict__ first, int64_t nslots, uint32_t* __restrict__ seen) {
  int64_t nwords = (nslots + 31) >> 5;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += stride) {
    uint32_t bits = 0;
    for (int k = 0; k < 32; ++k) {
      int64_t sl = (w << 5) + k;
      if (sl < nslots && first[sl] != kNoRow) bits |= 1u << k;
    }
    seen[w] = bits;
  }
}
__global__ void __launch_bounds__(256) k_dense_slots_tail(const long long* __restrict__ keys, const uint8_t* __restrict__ valid, int64_t off,
                                                          int64_t row0, int64_t n, long long mn, unsigned int mask, unsigned int range,
                                                          const uint32_t* __restrict__ seen, unsigned int* first,
                                                          uint32_t* __restrict__ slot_of_row) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = row0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // 8 independent key loads in flight per thread: the loop is otherwise latency bound (one 8-byte load per iteration)
  for (; i + 7 * stride < n; i += 8 * stride) {
    long long k[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) k[u] = keys[i + u * stride];
    unsigned int sl[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      sl[u] = range;
      if (!valid || bit_get(valid, off + i + u * stride)) sl[u] = dense_slot_of(k[u], mn, mask);
    }
    uint32_t w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = seen[sl[u] >> 5];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      slot_of_row[i + u * stride] = sl[u];
      if (!((w[u] >> (sl[u] & 31)) & 1u)) {
        unsigned int r = (unsigned int)(i + u * stride);
        if (r < first[sl[u]]) atomicMin(&first[sl[u]], r);
      }
    }
  }
  for (; i < n; i += stride) {
    unsigned int s = range;
    if (!valid || bit_get(valid, off + i)) s = dense_slot_of(keys[i], mn, mask);
    slot_of_row[i] = s;
    if (!((seen[s >> 5] >> (s & 31)) & 1u)) {
      if ((unsigned int)i < first[s]) atomicMin(&first[s], (unsigned int)i);
    }
  }
}

// Tile-shaped variant of the tail (one block = one 4096-row sort tile) that also produces the tile's pass-0 digit histogram, so
// the first pass of every later sort by slot needs no histogram read of slot_of_row.
template <int BITS>
__global__ void __launch_bounds__(kSortBlock) k_dense_slots_tail_hist(const long long* __restrict__ keys, const uint8_t* __restrict__ valid,
                                                                      int64_t off, int64_t tile0, int64_t n, long long mn, unsigned int mask,
                                                                      unsigned int range, const uint32_t* __restrict__ seen, int64_t track_from,
                                                                      unsigned int* first, uint32_t* __restrict__ slot_of_row,
                                                                      uint32_t* __restrict__ hist, long long* __restrict__ tile_min,
                                                                      long long* __restrict__ tile_max) {
  // rows < track_from went through the full first-row protocol already; tile_min / tile_max (optional): key range of the tile
  // (MAX / MIN sentinels when it has no valid key), reduced afterwards -- a shared accumulator would serialise 1e7 atomics
  constexpr int R = 1 << BITS;
  __shared__ uint32_t h[R];
  for (int d = threadIdx.x; d < R; d += kSortBlock) h[d] = 0;
  __syncthreads();
  const int64_t tile = tile0 + blockIdx.x;
  const int64_t base = tile * kSortTile;
  long long k[kSortItems];
#pragma unroll
  for (int u = 0; u < kSortItems; ++u) {
    int64_t i = base + u * kSortBlock + threadIdx.x;
    k[u] = i < n ? keys[i] : 0;
  }
  unsigned int sl[kSortItems];
  uint32_t w[kSortItems];
  long long kmn = 0x7FFFFFFFFFFFFFFFll, kmx = (long long)0x8000000000000000ull;
#pragma unroll
  for (int u = 0; u < kSortItems; ++u) {
    int64_t i = base + u * kSortBlock + threadIdx.x;
    sl[u] = range;
    if (i < n && (!valid || bit_get(valid, off + i))) {
      sl[u] = dense_slot_of(k[u], mn, mask);
      kmn = k[u] < kmn ? k[u] : kmn;
      kmx = k[u] > kmx ? k[u] : kmx;
    }
    w[u] = i < n ? seen[sl[u] >> 5] : ~0u;
  }
#pragma unroll
  for (int u = 0; u < kSortItems; ++u) {
    int64_t i = base + u * kSortBlock + threadIdx.x;
    if (i >= n) continue;
    slot_of_row[i] = sl[u];
    atomicAdd(&h[sl[u] & (R - 1)], 1u);
    if (!((w[u] >> (sl[u] & 31)) & 1u) && i >= track_from) {
      if ((unsigned int)i < first[sl[u]]) atomicMin(&first[sl[u]], (unsigned int)i);
    }
  }
  __shared__ long long smn[kSortWaves], smx[kSortWaves];
  if (tile_min) {
    for (int d = 32; d >= 1; d >>= 1) {
      long long a = __shfl_xor(kmn, d, 64), b = __shfl_xor(kmx, d, 64);
      kmn = a < kmn ? a : kmn;
      kmx = b > kmx ? b : kmx;
    }
    if ((threadIdx.x & 63) == 0) {
      smn[threadIdx.x >> 6] = kmn;
      smx[threadIdx.x >> 6] = kmx;
    }
  }
  __syncthreads();
  if (tile_min && threadIdx.x == 0) {
    for (int w = 1; w < kSortWaves; ++w) {
      kmn = smn[w] < kmn ? smn[w] : kmn;
      kmx = smx[w] > kmx ? smx[w] : kmx;
    }
    tile_min[tile] = kmn;
    tile_max[tile] = kmx;
  }
  for (int d = threadIdx.x; d < R; d += kSortBlock) hist[tile * R + d] = h[d];
}

// Same, for domains of <= 2^20 slots: persistent workgroups (one per CU) keep the whole `seen` bitmap in LDS (128 KB), so the
// per-row bitmap lookup is an LDS read instead of a random TA/L1 access; one wave owns one tile at a time (wave-private histogram).
constexpr int kDenseLdsWords = 32768;
constexpr int kDenseLdsBlock = 1024;
template <int BITS>
__global__ void __launch_bounds__(kDenseLdsBlock) k_dense_slots_tail_hist_lds(const long long* __restrict__ keys, const uint8_t* __restrict__ valid,
                                                                              int64_t off, int64_t tile0, int64_t ntiles, int64_t n, long long mn,
                                                                              unsigned int mask, unsigned int range,
                                                                              const uint32_t* __restrict__ seen, int nwords, int64_t track_from,
                                                                              unsigned int* first, uint32_t* __restrict__ slot_of_row,
                                                                              uint32_t* __restrict__ hist, KeyRange* __restrict__ range_out) {
  // rows < track_from already went through the full first-row protocol (k_dense_slots): here they only get their slot, their
  // histogram count and their share of the min/max.  range_out (optional): one exact KeyRange per workgroup.
  constexpr int R = 1 << BITS;
  constexpr int W = kDenseLdsBlock / 64;
  __shared__ uint32_t lseen[kDenseLdsWords];
  __shared__ uint32_t lh[W][R];
  __shared__ long long smn[W], smx[W];
  __shared__ int sany[W];
  for (int i = threadIdx.x; i < nwords; i += kDenseLdsBlock) lseen[i] = seen[i];
  for (int i = threadIdx.x; i < W * R; i += kDenseLdsBlock) (&lh[0][0])[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t nw = (int64_t)gridDim.x * W;
  long long kmn = 0x7FFFFFFFFFFFFFFFll, kmx = (long long)0x8000000000000000ull;
  int any = 0;
  bool few = true;  // wave-uniform
  for (int64_t tile = tile0 + (int64_t)blockIdx.x * W + wave; tile < ntiles; tile += nw) {
    const int64_t base = tile * kSortTile;
#pragma unroll 1
    for (int c = 0; c < kSortTile / 1024; ++c) {
      long long k[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        int64_t i = base + c * 1024 + u * 64 + lane;
        k[u] = i < n ? keys[i] : 0;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        int64_t i = base + c * 1024 + u * 64 + lane;
        const bool in = i < n;
        unsigned int sl = range;
        if (in && (!valid || bit_get(valid, off + i))) {
          sl = dense_slot_of(k[u], mn, mask);
          kmn = k[u] < kmn ? k[u] : kmn;
          kmx = k[u] > kmx ? k[u] : kmx;
          any = 1;
        }
        const uint32_t w = lseen[sl >> 5];
        if (in) slot_of_row[i] = sl;
        // few distinct digits (few distinct keys): lanes adding to the same LDS word serialise, so peel the rows off digit by
        // digit and add each digit's count once.  `few` is dropped for good the first time a step needs more than 4 rounds (a round costs about as much as a 4-way conflict).
        const unsigned int d = sl & (R - 1);
        if (few) {
          uint64_t rem = __ballot(in);
          int rounds = 0;
          while (rem && rounds < 4) {
            const int leader = __ffsll((unsigned long long)rem) - 1;
            const unsigned int dl = (unsigned int)__shfl((int)d, leader, 64);
            const uint64_t grp = __ballot(in && d == dl) & rem;
            if (lane == leader) atomicAdd(&lh[wave][dl], (uint32_t)__popcll(grp));
            rem &= ~grp;
            ++rounds;
          }
          if (rem) {
            few = false;
            if ((rem >> lane) & 1) atomicAdd(&lh[wave][d], 1u);
          }
        } else if (in) {
          atomicAdd(&lh[wave][d], 1u);
        }
        if (in && !((w >> (sl & 31)) & 1u) && i >= track_from) {
          if ((unsigned int)i < first[sl]) atomicMin(&first[sl], (unsigned int)i);
        }
      }
    }
    for (int d = lane; d < R; d += 64) {
      hist[tile * R + d] = lh[wave][d];
      lh[wave][d] = 0;
    }
  }
  if (range_out) {
    for (int d = 32; d >= 1; d >>= 1) {
      long long a = __shfl_xor(kmn, d, 64), b = __shfl_xor(kmx, d, 64);
      int c = __shfl_xor(any, d, 64);
      kmn = a < kmn ? a : kmn;
      kmx = b > kmx ? b : kmx;
      any |= c;
    }
    if (lane == 0) {
      smn[wave] = kmn;
      smx[wave] = kmx;
      sany[wave] = any;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < W; ++w) {
        kmn = smn[w] < kmn ? smn[w] : kmn;
        kmx = smx[w] > kmx ? smx[w] : kmx;
        any |= sany[w];
      }
      range_out[blockIdx.x].vmin = kmn;
      range_out[blockIdx.x].vmax = kmx;
      range_out[blockIdx.x].any = any;
    }
  }
}

// first-row of every slot: from the hash table (table != nullptr) or the dense first[] array
struct OccPred {
  const Slot* table;
  const unsigned int* first;
  unsigned int region, cap;  // region != 0: i is a LOGICAL slot of the partitioned table
  __device__ bool operator()(int64_t i) const { return (table ? table[phys_slot(i, region, cap)].first : first[i]) != kNoRow; }
};
struct OccEmit {
  const Slot* table;
  const unsigned int* first;
  unsigned int region, cap;
  uint32_t* occ_slot;
  uint32_t* occ_first;
  __device__ void operator()(int64_t pos, int64_t i) const {
    occ_slot[pos] = (uint32_t)i;
    occ_first[pos] = table ? table[phys_slot(i, region, cap)].first : first[i];
  }
};

// sorted_slot[r] = slot of the r-th group in first-occurrence order
// null_slot: the slot of the null key; table == nullptr => dense mode (key = dense_min + slot)
__global__ void k_assign_gids(const Slot* __restrict__ table, long long dense_min, unsigned int dense_mask, uint32_t* __restrict__ gid_of_slot,
                              const uint32_t* __restrict__ sorted_first, const uint32_t* __restrict__ sorted_slot, int64_t G,
                              unsigned int null_slot, int64_t* __restrict__ uniques, uint8_t* __restrict__ unique_ok,
                              int64_t* __restrict__ first_rows, unsigned int region) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < G; r += stride) {
    unsigned int s = sorted_slot[r];
    gid_of_slot[s] = (unsigned int)r;
    long long k;
    if (table) {
      k = table[phys_slot(s, region, null_slot)].key;
      if (s == null_slot + 1) k = kEmptyKey;
    } else {
      k = dense_mask ? (long long)((unsigned long long)dense_min + (((unsigned long long)s - (unsigned long long)dense_min) & dense_mask))
                     : (long long)((unsigned long long)dense_min + (unsigned long long)s);
    }
    if (s == null_slot) k = 0;
    uniques[r] = k;
    unique_ok[r] = s != null_slot;
    first_rows[r] = (int64_t)sorted_first[r];
  }
}
__global__ void k_gid_of_occ(const uint32_t* __restrict__ gid_of_slot, const uint32_t* __restrict__ occ_slot, int64_t G, uint32_t* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < G; k += stride) out[k] = gid_of_slot[occ_slot[k]];
}
__global__ void k_row_gids(const uint32_t* __restrict__ gid_of_slot, const uint32_t* __restrict__ slot_of_row, int64_t n, uint32_t* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = gid_of_slot[slot_of_row[i]];
}

__global__ void k_map_ids(const uint32_t* __restrict__ gid_of_slot, const uint32_t* __restrict__ slot_of_row, const uint32_t* __restrict__ seg_start,
                          int64_t G, int64_t n, const int64_t* __restrict__ map, int64_t* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint32_t g;
    if (gid_of_slot) g = gid_of_slot[slot_of_row[i]];
    else {
      int64_t lo = 0, hi = G;
      while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (seg_start[mid] <= (uint32_t)i) lo = mid;
        else hi = mid;
      }
      g = (uint32_t)lo;
    }
    out[i] = map[g];
  }
}

// keys for the value sort when the value column has nulls: bit 31 = row is null
__global__ void k_flag_keys(const uint32_t* __restrict__ slot_of_row, const uint8_t* __restrict__ valid, int64_t off, int64_t n,
                            uint32_t* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = slot_of_row[i] | (bit_get(valid, off + i) ? 0u : 0x80000000u);
}

// seg_start[k] = first position in sorted keys whose (masked) key >= occ_slot[k]; seg_start[G] = n
__global__ void k_seg_starts(const uint32_t* __restrict__ sorted_keys, int64_t n, const uint32_t* __restrict__ occ_slot, int64_t G,
                             uint32_t* __restrict__ seg_start) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= G; k += stride) {
    if (k == G) {
      seg_start[k] = (uint32_t)n;
      continue;
    }
    uint32_t target = occ_slot[k];
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      if ((sorted_keys[mid] & kSortKeyMask) < target) lo = mid + 1;
      else hi = mid;
    }
    seg_start[k] = (uint32_t)lo;
  }
}

// seg_start[k] = slot_start[occ_slot[k]] (starts of every slot's rows, from k_level_starts); seg_start[G] = n
__global__ void k_seg_starts_from_slots(const uint32_t* __restrict__ slot_start, int64_t n, const uint32_t* __restrict__ occ_slot, int64_t G,
                                        uint32_t* __restrict__ seg_start) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= G; k += stride) seg_start[k] = k == G ? (uint32_t)n : slot_start[occ_slot[k]];
}

// k_seg_starts on keys that carry extra bits above `mask`
__global__ void k_seg_starts_masked(const uint32_t* __restrict__ sorted_keys, int64_t n, uint32_t mask, int64_t G, uint32_t* __restrict__ seg_start) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= G; k += stride) {
    int64_t lo = 0, hi = n;
    if (k == G) lo = n;
    else
      while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((sorted_keys[mid] & mask) < (uint32_t)k) lo = mid + 1;
        else hi = mid;
      }
    seg_start[k] = (uint32_t)lo;
  }
}

// ---------------------------------------------------------------- segmented reduce (dense values: no nulls)
struct SegOut {
  double* sum_f;     // SUM of float64 values, or nullptr
  long long* sum_i;  // SUM of int64 values (wrapping)
  double* mean;
  void* vmin;        // T*
  void* vmax;        // T*
  long long* count;
};

constexpr int kSegWaves = 4;
constexpr int64_t kBigSeg = 65536;  // rows per sub-segment of a long group = 2^12 sixteen-value leaves: a full one is ONE level-12 node
constexpr int kBigLevels = 16;      // counter levels kept per sub-segment (0..12 are used)

template <typename T>
__device__ __forceinline__ double seg_to_f64(T x) { return (double)x; }

// uniform (whole-wave) replay of Arrow's counter with the level sums in LDS; lane 0 stores, every lane reads
__device__ __forceinline__ void lds_counter_push(double* csum, uint64_t& mask, int& root, double x, int level, int lane) {
  int cur = level;
  uint64_t mb = 1ull << level;
  double v = csum[cur] + x;
  mask ^= mb;
  while ((mask & mb) == 0) {
    if (lane == 0) csum[cur] = 0.0;
    ++cur;
    mb <<= 1;
    v = csum[cur] + v;
    mask ^= mb;
  }
  if (lane == 0) csum[cur] = v;
  if (cur > root) root = cur;
}

// The wave walks its (group, chunk) sequence with the NEXT chunk's 16 loads per lane already in flight while the current chunk
// is staged and reduced, and the bounds of the next group loaded one group ahead: without this every group pays a full
// dependent seg_start -> values memory round trip with nothing else to do (measured 3.0 -> see DESIGN.md).
template <typename T, bool WANT_PAIRWISE, bool WANT_MINMAX, bool WANT_ISUM>
__global__ void __launch_bounds__(kSegWaves * 64) k_seg_reduce(const T* __restrict__ vals, const uint32_t* __restrict__ seg_start,
                                                               int64_t nseg, const uint32_t* __restrict__ out_index, SegOut out, int64_t min_len) {
  // groups of <= min_len rows belong to k_seg_reduce_mid (batches of short groups per wave) and are skipped here like the long ones
  constexpr int LEAF = 16;              // Arrow's kBlockSize
  constexpr int kSegChunk = 64 * LEAF;  // values per wave-chunk = 64 leaves
  __shared__ double stage[kSegWaves][64 * 17];
  __shared__ double csum_all[kSegWaves][48];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* lds = stage[wave];
  double* csum = csum_all[wave];
  const int64_t nw = (int64_t)gridDim.x * kSegWaves;
  int64_t k = (int64_t)blockIdx.x * kSegWaves + wave;
  if (k >= nseg) return;
  // groups longer than kBigSeg rows are reduced by k_seg_reduce_sub / k_seg_combine_big (many waves per group): here they
  // are walked as empty segments whose result is not written
  int64_t s = seg_start[k], e = seg_start[k + 1];
  bool big = e - s > kBigSeg || e - s <= min_len;
  if (big) e = s;
  int64_t s_next = 0, e_next = 0;  // bounds of group k + nw
  bool big_next = false;
  if (k + nw < nseg) {
    s_next = seg_start[k + nw];
    e_next = seg_start[k + nw + 1];
    big_next = e_next - s_next > kBigSeg || e_next - s_next <= min_len;
    if (big_next) e_next = s_next;
  }
  int64_t c0 = 0;
  T cur[LEAF];
  {
    const int cl = (int)((e - s) < kSegChunk ? (e - s) : kSegChunk);
#pragma unroll
    for (int q = 0; q < LEAF; ++q) {
      int idx = q * 64 + lane;
      cur[q] = idx < cl ? vals[s + idx] : T(0);
    }
  }
  Extreme<T> ext;
  ext.init();
  unsigned long long isum = 0;
  uint64_t mask = 0;
  int root = 0;
  double single = 0.0;  // result when the group fits one chunk
  if (WANT_PAIRWISE && (e - s) > kSegChunk) {
    if (lane < 48) csum[lane] = 0.0;
  }
  for (;;) {
    const int64_t len = e - s;
    const bool multi = len > kSegChunk;
    const int cl = (int)((len - c0) < kSegChunk ? (len - c0) : kSegChunk);
    const bool last_chunk = c0 + kSegChunk >= len;
    // ---- issue the next chunk's loads
    const int64_t nk = last_chunk ? k + nw : k;
    const bool have_next = nk < nseg;
    const int64_t ns = last_chunk ? s_next : s, ne = last_chunk ? e_next : e, nc0 = last_chunk ? 0 : c0 + kSegChunk;
    T nxt[LEAF];
    if (have_next) {
      const int ncl = (int)((ne - ns - nc0) < kSegChunk ? (ne - ns - nc0) : kSegChunk);
#pragma unroll
      for (int q = 0; q < LEAF; ++q) {
        int idx = q * 64 + lane;
        nxt[q] = idx < ncl ? vals[ns + nc0 + idx] : T(0);
      }
    }
    int64_t s_nn = 0, e_nn = 0;
    bool big_nn = false;
    if (last_chunk && nk + nw < nseg) {  // bounds two groups ahead, consumed when the next group finishes
      s_nn = seg_start[nk + nw];
      e_nn = seg_start[nk + nw + 1];
      big_nn = e_nn - s_nn > kBigSeg || e_nn - s_nn <= min_len;
      if (big_nn) e_nn = s_nn;
    }
    // ---- current chunk
#pragma unroll
    for (int q = 0; q < LEAF; ++q) {
      int idx = q * 64 + lane;
      if (idx < cl) {
        T x = cur[q];
        if (WANT_PAIRWISE) lds[idx + (idx >> 4)] = seg_to_f64(x);
        if (WANT_MINMAX) {
          if (x == x) ext.add(x, (long long)(c0 + idx));
        }
        if (WANT_ISUM) isum += (unsigned long long)x;
      }
    }
    __builtin_amdgcn_wave_barrier();  // the LDS image is wave-private: in-order LDS issue makes it visible to all lanes
    if (WANT_PAIRWISE) {
      const int m = (cl + LEAF - 1) / LEAF;  // leaves in this chunk (wave-uniform)
      double x = 0.0;
      const int first = lane * LEAF;
      if (first < cl) {
        int cnt = cl - first < 16 ? cl - first : 16;
        x = leaf_sum(&lds[lane * 17], cnt);
      }
      // butterfly; pick the perfect subtrees that tile [0, m)
      double node[7];
#pragma unroll
      for (int sft = 0; sft < 6; ++sft) {
        node[sft] = 0.0;
        if ((m >> sft) & 1) node[sft] = __shfl(x, m & ~((2 << sft) - 1), 64);
        double y = __shfl_down(x, 1 << sft, 64);
        x = x + y;
      }
      node[6] = __shfl(x, 0, 64);
      if (!multi) {
        // fold ascending: acc = lowest node; acc = higher + acc
        bool have = false;
        double acc = 0.0;
#pragma unroll
        for (int sft = 0; sft <= 6; ++sft) {
          if ((m >> sft) & 1) {
            acc = have ? node[sft] + acc : node[sft];
            have = true;
          }
        }
        single = acc;
      } else {
#pragma unroll
        for (int sft = 6; sft >= 0; --sft)
          if ((m >> sft) & 1) lds_counter_push(csum, mask, root, node[sft], sft, lane);
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (last_chunk) {
      // ---- group k is complete
      const uint32_t oi = out_index ? out_index[k] : (uint32_t)k;
      double total = single;
      if (WANT_PAIRWISE && multi) {
        double acc = csum[0];
        for (int i = 1; i <= root; ++i) acc = csum[i] + acc;
        total = acc;
      }
      if (WANT_MINMAX) {
        for (int d = 32; d > 0; d >>= 1) {
          T omin = __shfl_down(ext.vmin, d, 64), omax = __shfl_down(ext.vmax, d, 64);
          long long ormin = __shfl_down(ext.rmin, d, 64), ormax = __shfl_down(ext.rmax, d, 64);
          ext.merge(omin, ormin, omax, ormax);
        }
      }
      if (WANT_ISUM) {
        for (int d = 32; d > 0; d >>= 1) isum += __shfl_down(isum, d, 64);
      }
      if (lane == 0 && !big) {
        if (WANT_PAIRWISE) {
          if (out.sum_f) out.sum_f[oi] = total;
          if (out.mean) out.mean[oi] = total / (double)len;
        }
        if (WANT_ISUM && out.sum_i) out.sum_i[oi] = (long long)isum;
        if (WANT_MINMAX) {
          T nanv = T(0);
          if constexpr (__is_same(T, double)) nanv = __builtin_nan("");
          if (out.vmin) static_cast<T*>(out.vmin)[oi] = ext.rmin < 0 ? nanv : ext.vmin;
          if (out.vmax) static_cast<T*>(out.vmax)[oi] = ext.rmax < 0 ? nanv : ext.vmax;
        }
        if (out.count) out.count[oi] = (long long)len;
      }
      if (!have_next) break;
      // ---- reset the per-group state
      ext.init();
      isum = 0;
      mask = 0;
      root = 0;
      single = 0.0;
      big = big_next;
      s_next = s_nn;
      e_next = e_nn;
      big_next = big_nn;
      if (WANT_PAIRWISE && (ne - ns) > kSegChunk) {
        __builtin_amdgcn_wave_barrier();
        if (lane < 48) csum[lane] = 0.0;
      }
    }
    k = nk;
    s = ns;
    e = ne;
    c0 = nc0;
#pragma unroll
    for (int q = 0; q < LEAF; ++q) cur[q] = nxt[q];
  }
}

// ---------------------------------------------------------------- one wave reduces one contiguous segment (any length).
// Chunks of 1024 values (64 leaves), the next chunk's loads in flight while the current one is staged and reduced; the chunk's
// perfect subtrees go through the LDS-resident counter (csum/mask/root: Arrow's state after the segment; the caller folds it or
// stores it).  ext (wave-reduced, valid in lane 0) and isum (wave-reduced) cover the whole segment; rows are numbered from row_base.
template <typename T, bool WANT_PAIRWISE, bool WANT_MINMAX, bool WANT_ISUM, bool PREFETCH = true>
__device__ __forceinline__ void seg_chunked(const T* __restrict__ vals, int64_t s, int64_t len, long long row_base, int lane, double* lds /* 64*17 */,
                                            double* csum /* 48 */, Extreme<T>& ext, unsigned long long& isum, uint64_t& mask, int& root) {
  constexpr int LEAF = 16;
  constexpr int kSegChunk = 64 * LEAF;
  __builtin_amdgcn_wave_barrier();
  if (lane < 48) csum[lane] = 0.0;
  // PREFETCH = false (callers with many live registers of their own): plain load-then-reduce per chunk, half the registers
  T cur[LEAF];
  if (PREFETCH) {
    const int cl = (int)(len < kSegChunk ? len : kSegChunk);
#pragma unroll
    for (int q = 0; q < LEAF; ++q) {
      int idx = q * 64 + lane;
      cur[q] = idx < cl ? vals[s + idx] : T(0);
    }
  }
  for (int64_t c0 = 0; c0 < len; c0 += kSegChunk) {
    const int cl = (int)((len - c0) < kSegChunk ? (len - c0) : kSegChunk);
    T nxt[PREFETCH ? LEAF : 1];
    if (PREFETCH) {
      const int64_t n0 = c0 + kSegChunk;
      const int ncl = n0 < len ? (int)((len - n0) < kSegChunk ? (len - n0) : kSegChunk) : 0;
#pragma unroll
      for (int q = 0; q < LEAF; ++q) {
        int idx = q * 64 + lane;
        nxt[PREFETCH ? q : 0] = idx < ncl ? vals[s + n0 + idx] : T(0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < LEAF; ++q) {
        int idx = q * 64 + lane;
        cur[q] = idx < cl ? vals[s + c0 + idx] : T(0);
      }
    }
#pragma unroll
    for (int q = 0; q < LEAF; ++q) {
      int idx = q * 64 + lane;
      if (idx < cl) {
        T x = cur[q];
        if (WANT_PAIRWISE) lds[idx + (idx >> 4)] = seg_to_f64(x);
        if (WANT_MINMAX) {
          if (x == x) ext.add(x, row_base + (long long)(c0 + idx));
        }
        if (WANT_ISUM) isum += (unsigned long long)x;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (WANT_PAIRWISE) {
      const int m = (cl + LEAF - 1) / LEAF;
      double x = 0.0;
      const int first = lane * LEAF;
      if (first < cl) {
        int cnt = cl - first < 16 ? cl - first : 16;
        x = leaf_sum(&lds[lane * 17], cnt);
      }
      double node[7];
#pragma unroll
      for (int sft = 0; sft < 6; ++sft) {
        node[sft] = 0.0;
        if ((m >> sft) & 1) node[sft] = __shfl(x, m & ~((2 << sft) - 1), 64);
        double y = __shfl_down(x, 1 << sft, 64);
        x = x + y;
      }
      node[6] = __shfl(x, 0, 64);
#pragma unroll
      for (int sft = 6; sft >= 0; --sft)
        if ((m >> sft) & 1) lds_counter_push(csum, mask, root, node[sft], sft, lane);
    }
    __builtin_amdgcn_wave_barrier();
    if (PREFETCH) {
#pragma unroll
      for (int q = 0; q < LEAF; ++q) cur[q] = nxt[PREFETCH ? q : 0];
    }
  }
  if (WANT_MINMAX) {
    for (int d = 32; d > 0; d >>= 1) {
      T omin = __shfl_down(ext.vmin, d, 64), omax = __shfl_down(ext.vmax, d, 64);
      long long ormin = __shfl_down(ext.rmin, d, 64), ormax = __shfl_down(ext.rmax, d, 64);
      ext.merge(omin, ormin, omax, ormax);
    }
  }
  if (WANT_ISUM) {
    for (int d = 32; d > 0; d >>= 1) isum += __shfl_down(isum, d, 64);
  }
}

// ---------------------------------------------------------------- segmented reduce, short groups: batches of groups per wave.
// Groups of a few to a few hundred rows leave most of a wave idle in k_seg_reduce (and a thread per group thrashes the L1).  Here a wave takes a
// run of consecutive short groups (<= 64 groups, <= 1024 rows: the grouped values are contiguous), loads the whole run coalesced
// into LDS, sums the 16-value leaves with one lane per leaf and then combines every group's leaves with one lane per group
// (in-place perfect subtrees + ascending fold == Arrow's counter).  Groups longer than kMidLen are left to k_seg_reduce.
constexpr int kMidLen = 256;
constexpr int kMidRows = 1024;
template <typename T, bool WANT_PAIRWISE, bool WANT_MINMAX, bool WANT_ISUM>
__global__ void __launch_bounds__(kSegWaves * 64) k_seg_reduce_mid(const T* __restrict__ vals, const uint32_t* __restrict__ seg_start, int64_t nseg,
                                                                   const uint32_t* __restrict__ out_index, SegOut out, int64_t groups_per_wave) {
  __shared__ T stage_all[kSegWaves][64 * 17];  // >= kMidRows values; the padded 64 x 17 image when a longer group is chunked
  __shared__ double csum_all[kSegWaves][48];
  __shared__ double leaf_all[kSegWaves][kMidRows / 16 + 64];
  __shared__ int lp_all[kSegWaves][65];
  __shared__ int goff_all[kSegWaves][64];
  __shared__ int glen_all[kSegWaves][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* stage = stage_all[wave];
  double* leaf = leaf_all[wave];
  int* lp = lp_all[wave];
  int* goff = goff_all[wave];
  int* glen = glen_all[wave];
  const int64_t gw = (int64_t)blockIdx.x * kSegWaves + wave;
  int64_t k0 = gw * groups_per_wave;
  const int64_t kend = k0 + groups_per_wave < nseg ? k0 + groups_per_wave : nseg;
  while (k0 < kend) {
    const int64_t kk = k0 + lane;
    const int64_t b0 = seg_start[kk < kend ? kk : kend], b1 = seg_start[kk + 1 < kend ? kk + 1 : kend];
    const int64_t S = __shfl(b0, 0, 64);
    const int len = (int)(b1 - b0);
    const bool ok = kk < kend && len <= kMidLen && (b1 - S) <= kMidRows;
    const uint64_t okm = __ballot(ok);
    const int g = ~okm ? __ffsll((unsigned long long)~okm) - 1 : 64;  // leading run of short groups that fits
    if (g == 0) {
      // a longer group: the whole wave chunks through it (groups beyond kBigSeg belong to the many-waves path)
      const int64_t glen0 = __shfl(b1, 0, 64) - S;
      if (glen0 <= kBigSeg) {
        Extreme<T> ext;
        ext.init();
        unsigned long long isum = 0;
        uint64_t mask = 0;
        int root = 0;
        double* csum = csum_all[wave];
        seg_chunked<T, WANT_PAIRWISE, WANT_MINMAX, WANT_ISUM, false>(vals, S, glen0, 0ll, lane, reinterpret_cast<double*>(stage), csum, ext, isum, mask,
                                                                     root);
        double total = 0.0;
        if (WANT_PAIRWISE) {
          double acc = csum[0];
          for (int i = 1; i <= root; ++i) acc = csum[i] + acc;
          total = acc;
        }
        if (lane == 0) {
          const uint32_t oi = out_index ? out_index[k0] : (uint32_t)k0;
          if (WANT_PAIRWISE) {
            if (out.sum_f) out.sum_f[oi] = total;
            if (out.mean) out.mean[oi] = total / (double)glen0;
          }
          if (WANT_ISUM && out.sum_i) out.sum_i[oi] = (long long)isum;
          if (WANT_MINMAX) {
            T nanv = T(0);
            if constexpr (__is_same(T, double)) nanv = __builtin_nan("");
            if (out.vmin) static_cast<T*>(out.vmin)[oi] = ext.rmin < 0 ? nanv : ext.vmin;
            if (out.vmax) static_cast<T*>(out.vmax)[oi] = ext.rmax < 0 ? nanv : ext.vmax;
          }
          if (out.count) out.count[oi] = (long long)glen0;
        }
        __builtin_amdgcn_wave_barrier();
      }
      k0 += 1;
      continue;
    }
    const int R = (int)(__shfl(b1, g - 1, 64) - S);
#pragma unroll
    for (int q = 0; q < kMidRows / 64; ++q) {
      int idx = q * 64 + lane;
      if (idx < R) stage[idx] = vals[S + idx];
    }
    const int nl = lane < g ? (len + 15) >> 4 : 0;
    const int incl = wave_inclusive_scan(nl, SumOp());
    const int excl = incl - nl;
    const int NL = __shfl(incl, 63, 64);
    lp[lane] = excl;
    if (lane == 63) lp[64] = NL;
    goff[lane] = (int)(b0 - S);
    glen[lane] = len;
    __builtin_amdgcn_wave_barrier();
    if (WANT_PAIRWISE) {
      for (int L = lane; L < NL; L += 64) {
        int lo = 0, hi = g - 1;
        while (lo < hi) {
          int mid = (lo + hi + 1) >> 1;
          if (lp[mid] <= L) lo = mid;
          else hi = mid - 1;
        }
        const int j = L - lp[lo];
        const int off = goff[lo] + 16 * j;
        int cnt = glen[lo] - 16 * j;
        cnt = cnt < 16 ? cnt : 16;
        double acc = 0.0;
        if (cnt == 16) {
#pragma unroll
          for (int q = 0; q < 16; ++q) acc += seg_to_f64(stage[off + q]);
        } else {
          for (int q = 0; q < cnt; ++q) acc += seg_to_f64(stage[off + q]);
        }
        leaf[L] = acc;
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (lane < g) {
      const uint32_t oi = out_index ? out_index[kk] : (uint32_t)kk;
      if (WANT_PAIRWISE) {
        double* x = leaf + excl;
        const int m = nl;
        for (int stride = 1; stride < m; stride <<= 1)
          for (int i = 0; i + 2 * stride <= m; i += 2 * stride) x[i] = x[i] + x[i + stride];
        double acc = 0.0;
        bool have = false;
        int pos = m;
        for (int jb = 0; jb < 7; ++jb)
          if ((m >> jb) & 1) {
            pos -= 1 << jb;
            acc = have ? x[pos] + acc : x[pos];
            have = true;
          }
        if (out.sum_f) out.sum_f[oi] = acc;
        if (out.mean) out.mean[oi] = acc / (double)len;
      }
      if (WANT_MINMAX || WANT_ISUM) {
        const T* v = stage + (int)(b0 - S);
        unsigned long long isum = 0;
        T vmn = T(0), vmx = T(0);
        bool has = false;
        for (int r = 0; r < len; ++r) {
          T xv = v[r];
          if (WANT_ISUM) isum += (unsigned long long)xv;
          if (WANT_MINMAX && xv == xv) {
            if (!has) { vmn = vmx = xv; has = true; }
            else {
              if (xv < vmn) vmn = xv;
              if (xv > vmx) vmx = xv;
            }
          }
        }
        if (WANT_ISUM && out.sum_i) out.sum_i[oi] = (long long)isum;
        if (WANT_MINMAX) {
          T nanv = T(0);
          if constexpr (__is_same(T, double)) nanv = __builtin_nan("");
          if (out.vmin) static_cast<T*>(out.vmin)[oi] = has ? vmn : nanv;
          if (out.vmax) static_cast<T*>(out.vmax)[oi] = has ? vmx : nanv;
        }
      }
      if (out.count) out.count[oi] = (long long)len;
    }
    __builtin_amdgcn_wave_barrier();
    k0 += g;
  }
}

// ---------------------------------------------------------------- long groups: many waves per group.
// A group of more than kBigSeg rows is cut into sub-segments of kBigSeg rows (aligned to the group start, so every full
// sub-segment is a perfect subtree of 2^12 leaves = one level-12 node of Arrow's counter).  One wave reduces one sub-segment to
// its counter state; one thread per long group then replays the states in order (full ones are a single level-12 push, the last
// one pushes its <= 13 nodes from the highest level down, which is legal because everything before it is 2^12-aligned).
template <typename T>
struct SubState {
  double csum[kBigLevels];
  unsigned long long mask;
  unsigned long long isum;
  T vmin, vmax;
  long long rmin, rmax;
};
struct BigPred {
  const uint32_t* seg_start;
  __device__ bool operator()(int64_t k) const { return (int64_t)seg_start[k + 1] - (int64_t)seg_start[k] > kBigSeg; }
};
struct BigEmit {
  uint32_t* big_idx;
  __device__ void operator()(int64_t pos, int64_t k) const { big_idx[pos] = (uint32_t)k; }
};
// item_off[b] = first work item (sub-segment) of long group b; item_off[B] = number of items.  One workgroup.
__global__ void __launch_bounds__(256) k_big_offsets(const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ big_idx, int64_t B,
                                                     int64_t* __restrict__ item_off) {
  __shared__ int64_t smem[8];
  int64_t carry = 0;
  for (int64_t b0 = 0; b0 < B; b0 += 256) {
    int64_t b = b0 + threadIdx.x;
    int64_t nsub = 0;
    if (b < B) {
      const uint32_t k = big_idx[b];
      nsub = ((int64_t)seg_start[k + 1] - (int64_t)seg_start[k] + kBigSeg - 1) / kBigSeg;
    }
    int64_t total;
    int64_t pre = block_exclusive_scan(nsub, SumOp(), &total, smem);
    if (b < B) item_off[b] = carry + pre;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) item_off[B] = carry;
}
template <typename T, bool WANT_PAIRWISE, bool WANT_MINMAX, bool WANT_ISUM>
__global__ void __launch_bounds__(kSegWaves * 64) k_seg_reduce_sub(const T* __restrict__ vals, const uint32_t* __restrict__ seg_start,
                                                                   const uint32_t* __restrict__ big_idx, const int64_t* __restrict__ item_off,
                                                                   int64_t B, SubState<T>* __restrict__ state) {
  __shared__ double stage[kSegWaves][64 * 17];
  __shared__ double csum_all[kSegWaves][48];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* lds = stage[wave];
  double* csum = csum_all[wave];
  const int64_t nitems = item_off[B];
  const int64_t nw = (int64_t)gridDim.x * kSegWaves;
  for (int64_t t = (int64_t)blockIdx.x * kSegWaves + wave; t < nitems; t += nw) {
    // long group of item t: last b with item_off[b] <= t
    int64_t lo = 0, hi = B - 1;
    while (lo < hi) {
      int64_t mid = (lo + hi + 1) >> 1;
      if (item_off[mid] <= t) lo = mid;
      else hi = mid - 1;
    }
    const uint32_t k = big_idx[lo];
    const int64_t j = t - item_off[lo];
    const int64_t s = (int64_t)seg_start[k] + j * kBigSeg;
    const int64_t gend = seg_start[k + 1];
    const int64_t e = s + kBigSeg < gend ? s + kBigSeg : gend;
    Extreme<T> ext;
    ext.init();
    unsigned long long isum = 0;
    uint64_t mask = 0;
    int root = 0;
    seg_chunked<T, WANT_PAIRWISE, WANT_MINMAX, WANT_ISUM>(vals, s, e - s, (long long)(j * kBigSeg), lane, lds, csum, ext, isum, mask, root);
    if (lane < kBigLevels) state[t].csum[lane] = csum[lane];
    if (lane == 0) {
      state[t].mask = mask;
      state[t].isum = isum;
      state[t].vmin = ext.vmin;
      state[t].vmax = ext.vmax;
      state[t].rmin = ext.rmin;
      state[t].rmax = ext.rmax;
    }
  }
}
template <typename T, bool WANT_PAIRWISE, bool WANT_MINMAX, bool WANT_ISUM>
__global__ void __launch_bounds__(64) k_seg_combine_big(const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ big_idx,
                                                        const int64_t* __restrict__ item_off, int64_t B, const SubState<T>* __restrict__ state,
                                                        const uint32_t* __restrict__ out_index, SegOut out) {
  // one wave per long group.  64 consecutive FULL sub-segments (64-aligned within the group) are a perfect subtree of level-12
  // nodes: one butterfly makes their level-18 node; everything else is replayed by lane 0.
  const int64_t b = blockIdx.x;
  const int lane = threadIdx.x;
  const uint32_t k = big_idx[b];
  const uint32_t oi = out_index ? out_index[k] : k;
  const long long len = (long long)seg_start[k + 1] - (long long)seg_start[k];
  PairwiseCounter c;  // used by lane 0 only
  if (lane == 0) c.init();
  Extreme<T> ext;
  ext.init();
  unsigned long long isum = 0;
  const int64_t begin = item_off[b], end = item_off[b + 1];
  for (int64_t t0 = begin; t0 < end; t0 += 64) {
    const int64_t t = t0 + lane;
    const bool have = t < end;
    unsigned long long m = 0;
    double v = 0.0;
    if (have) {
      m = state[t].mask;
      v = state[t].csum[12];
      if (WANT_MINMAX) ext.merge(state[t].vmin, state[t].rmin, state[t].vmax, state[t].rmax);
      if (WANT_ISUM) isum += state[t].isum;
    }
    if (WANT_PAIRWISE) {
      const bool all_full = (end - t0 >= 64) && __all(m == (1ull << 12));
      if (all_full) {
        const double node = wave_tree64(v);
        if (lane == 0) c.push(node, 18);
      } else if (lane == 0) {
        const int64_t cnt = end - t0 < 64 ? end - t0 : 64;
        for (int64_t i = 0; i < cnt; ++i) {
          const SubState<T>& st = state[t0 + i];
          for (int lvl = kBigLevels - 1; lvl >= 0; --lvl)
            if ((st.mask >> lvl) & 1) c.push(st.csum[lvl], lvl);
        }
      }
    }
  }
  if (WANT_MINMAX) {
    for (int d = 32; d > 0; d >>= 1) {
      T omin = __shfl_down(ext.vmin, d, 64), omax = __shfl_down(ext.vmax, d, 64);
      long long ormin = __shfl_down(ext.rmin, d, 64), ormax = __shfl_down(ext.rmax, d, 64);
      ext.merge(omin, ormin, omax, ormax);
    }
  }
  if (WANT_ISUM) {
    for (int d = 32; d > 0; d >>= 1) isum += __shfl_down(isum, d, 64);
  }
  if (lane != 0) return;
  if (WANT_PAIRWISE) {
    const double total = c.finish();
    if (out.sum_f) out.sum_f[oi] = total;
    if (out.mean) out.mean[oi] = total / (double)len;
  }
  if (WANT_ISUM && out.sum_i) out.sum_i[oi] = (long long)isum;
  if (WANT_MINMAX) {
    T nanv = T(0);
    if constexpr (__is_same(T, double)) nanv = __builtin_nan("");
    if (out.vmin) static_cast<T*>(out.vmin)[oi] = ext.rmin < 0 ? nanv : ext.vmin;
    if (out.vmax) static_cast<T*>(out.vmax)[oi] = ext.rmax < 0 ? nanv : ext.vmax;
  }
  if (out.count) out.count[oi] = len;
}

// ---------------------------------------------------------------- segmented reduce (nullable values): one wave per group.
// Arrow restarts the 16-value leaves at every run of valid rows, so leaf boundaries are data dependent.  Per 1024-row chunk a
// lane owns a 16-row window: the number of rows already in the leaf that is open at the window start comes from a "latest"
// scan across the lanes (a window is as long as a leaf, so a full window passes the count through unchanged), the open leaf's
// partial sum is the sequential sum of the previous window's last rows (one shuffle), and each lane walks its 16 validity bits
// emitting finished leaves in order.  The emitted leaf sums are merged with a butterfly whose lanes are aligned to the GLOBAL leaf
// index, so every perfect subtree it extracts is exactly a run of carries of Arrow's binary counter.
constexpr int64_t kHugeNullable = (int64_t)1 << 22;  // rows: beyond this a nullable group is not left to one wave
constexpr int kNullLeafCap = 64 * 9 + 8 + 64;  // a 16-row window emits at most 9 leaves (8 isolated values + the carried one); + the queue's tail

template <typename T>
__global__ void __launch_bounds__(kSegWaves * 64) k_seg_reduce_nullable(const T* __restrict__ vals, const uint32_t* __restrict__ sorted_keys,
                                                                        const uint8_t* __restrict__ row_valid, int64_t valid_off,
                                                                        const uint32_t* __restrict__ seg_start, int64_t nseg,
                                                                        const uint32_t* __restrict__ out_index, SegOut out,
                                                                        uint8_t* __restrict__ ok) {
  __shared__ double stage[kSegWaves][64 * 17];
  __shared__ double leafbuf[kSegWaves][kNullLeafCap];
  __shared__ double csum_all[kSegWaves][48];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* lds = stage[wave];
  double* leaves = leafbuf[wave];
  double* csum = csum_all[wave];
  int64_t gw = (int64_t)blockIdx.x * kSegWaves + wave;
  int64_t nw = (int64_t)gridDim.x * kSegWaves;
  for (int64_t k = gw; k < nseg; k += nw) {
    const int64_t s = seg_start[k], e = seg_start[k + 1];
    const int64_t len = e - s;
    if (len > kHugeNullable) continue;  // reduced slice by slice with the whole-column kernels (reduce_huge_nullable_groups)
    const uint32_t oi = out_index ? out_index[k] : (uint32_t)k;
    Extreme<T> ext;
    ext.init();
    long long zlast = -1;     // last valid zero-valued row of the group and its sign: the max tie rule of a group WITH nulls (minmax.hpp)
    unsigned long long isum = 0;
    long long nvalid = 0;
    uint64_t cmask = 0;       // binary counter occupancy (wave-uniform)
    int croot = 0;
    int pend = 0;             // finished leaves waiting in the queue for their block of 64 (wave-uniform, < 64 between chunks)
    int carry_pos = 0;        // rows already in the leaf that is open at the chunk start
    double carry_acc = 0.0;   // ... and their sequential sum
    if (lane < 48) csum[lane] = 0.0;
    for (int64_t c0 = 0; c0 < len; c0 += 1024) {
      const int cl = (int)((len - c0) < 1024 ? (len - c0) : 1024);
      // coalesced loads + validity words; lane l keeps the word that holds its window [16l, 16l+16)
      uint64_t myword = 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int idx = q * 64 + lane;
        bool v = false;
        if (idx < cl) {
          const int64_t i = s + c0 + idx;
          v = sorted_keys ? !(sorted_keys[i] >> 31) : (!row_valid || bit_get(row_valid, valid_off + i));
          T x = vals[i];
          lds[idx + (idx >> 4)] = (double)x;
          if (v) {
            isum += (unsigned long long)x;
            if (x == x) ext.add(x, (long long)(c0 + idx));
            if constexpr (__is_same(T, double)) {
              if (x == 0.0) {
                const long long zm = zero_mark(x, (long long)(c0 + idx));
                zlast = zm > zlast ? zm : zlast;
              }
            }
          }
        }
        const uint64_t bal = __ballot(v);
        nvalid += __popcll(bal);
        if ((lane >> 2) == q) myword = bal;
      }
      __builtin_amdgcn_wave_barrier();
      const unsigned m = (unsigned)(myword >> ((lane & 3) * 16)) & 0xFFFFu;
      const bool full = m == 0xFFFFu;
      const int t = full ? 16 : __builtin_clz(~(m << 16));  // valid rows at the END of the window (leading ones of m << 16)
      // rows in the open leaf at the start of every window ("latest" scan; a full window passes its own start value on)
      const int z = full ? -1 : t;
      const int inc = wave_inclusive_scan(z, LatestOp());
      const int exc = __shfl_up(inc, 1, 64);
      const int pos = (lane != 0 && exc >= 0) ? exc : carry_pos;  // exc < 0: every earlier window of this chunk is full
      // sequential sum of this window's last rows that stay in an open leaf (handed to the next window)
      const int cnt_tail = full ? pos : t;
      double tail = 0.0;
      for (int q = 16 - cnt_tail; q < 16; ++q) tail += lds[lane * 17 + q];
      double acc = __shfl_up(tail, 1, 64);
      if (lane == 0) acc = carry_acc;
      // pass 1: number of leaves this window finishes
      int nfin = 0;
      {
        int p = pos;
        for (int q = 0; q < 16; ++q) {
          if ((m >> q) & 1u) {
            if (++p == 16) { ++nfin; p = 0; }
          } else if (p > 0) { ++nfin; p = 0; }
        }
      }
      int inc_n = wave_inclusive_scan(nfin, SumOp());
      const int base = inc_n - nfin;
      const int total_new = __shfl(inc_n, 63, 64);
      // pass 2: emit the finished leaves in order
      {
        int p = pos, w = pend + base;
        double a = pos > 0 ? acc : 0.0;
        for (int q = 0; q < 16; ++q) {
          if ((m >> q) & 1u) {
            a = (p == 0 ? 0.0 : a) + lds[lane * 17 + q];
            if (++p == 16) { leaves[w++] = a; p = 0; }
          } else if (p > 0) { leaves[w++] = a; p = 0; }
        }
      }
      // state handed to the next chunk
      const int last_inc = __shfl(inc, 63, 64);
      const double last_tail = __shfl(tail, 63, 64);
      carry_pos = last_inc < 0 ? carry_pos : last_inc;
      carry_acc = last_tail;
      __builtin_amdgcn_wave_barrier();
      // merge: the finished leaves queue up behind `pend` leaves left over from earlier chunks (the queue always starts at a
      // multiple of 64 of the group's leaf sequence); every full block of 64 is one perfect subtree = ONE level-6 push
      {
        const int total = pend + total_new;
        int b = 0;
        for (; b + 64 <= total; b += 64) {
          const double node = wave_tree64(leaves[b + lane]);
          lds_counter_push(csum, cmask, croot, __shfl(node, 0, 64), 6, lane);
        }
        const int rem = total - b;
        double keep = 0.0;
        if (b > 0 && lane < rem) keep = leaves[b + lane];
        __builtin_amdgcn_wave_barrier();
        if (b > 0 && lane < rem) leaves[lane] = keep;
        pend = rem;
      }
      __builtin_amdgcn_wave_barrier();
    }
    // the queue's tail (< 64 leaves, aligned to a multiple of 64): its perfect subtrees, highest first
    if (pend > 0) {
      double x0 = lane < pend ? leaves[lane] : 0.0;
      double x1 = x0 + __shfl_down(x0, 1, 64);
      double x2 = x1 + __shfl_down(x1, 2, 64);
      double x3 = x2 + __shfl_down(x2, 4, 64);
      double x4 = x3 + __shfl_down(x3, 8, 64);
      double x5 = x4 + __shfl_down(x4, 16, 64);
      for (int sidx = 0; sidx < pend;) {
        int lg = 31 - __builtin_clz((unsigned)(pend - sidx));
        int tz = sidx == 0 ? 6 : __builtin_ctz((unsigned)sidx);
        const int j = tz < lg ? tz : lg;
        double xs = j == 0 ? x0 : j == 1 ? x1 : j == 2 ? x2 : j == 3 ? x3 : j == 4 ? x4 : x5;
        lds_counter_push(csum, cmask, croot, __shfl(xs, sidx, 64), j, lane);
        sidx += 1 << j;
      }
    }
    // the leaf still open at the end of the group
    if (carry_pos > 0) lds_counter_push(csum, cmask, croot, carry_acc, 0, lane);
    double total = 0.0;
    if (nvalid > 0) {
      double a = csum[0];
      for (int i = 1; i <= croot; ++i) a = csum[i] + a;
      total = a;
    }
    for (int d = 32; d > 0; d >>= 1) {
      T omin = __shfl_down(ext.vmin, d, 64), omax = __shfl_down(ext.vmax, d, 64);
      long long ormin = __shfl_down(ext.rmin, d, 64), ormax = __shfl_down(ext.rmax, d, 64);
      ext.merge(omin, ormin, omax, ormax);
      isum += __shfl_down(isum, d, 64);
      const long long oz = __shfl_down(zlast, d, 64);
      zlast = oz > zlast ? oz : zlast;
    }
    if (lane == 0) {
      if (out.sum_f) out.sum_f[oi] = total;
      if (out.mean) out.mean[oi] = nvalid ? total / (double)nvalid : 0.0;
      if (out.sum_i) out.sum_i[oi] = (long long)isum;
      T nanv = T(0);
      if constexpr (__is_same(T, double)) nanv = __builtin_nan("");
      if (out.vmin) static_cast<T*>(out.vmin)[oi] = ext.rmin < 0 ? nanv : ext.vmin;
      if constexpr (__is_same(T, double)) ext.vmax = zero_tie_fix(ext.vmax, zlast, nvalid < len);
      if (out.vmax) static_cast<T*>(out.vmax)[oi] = ext.rmax < 0 ? nanv : ext.vmax;
      if (out.count) out.count[oi] = nvalid;
      ok[oi] = nvalid > 0;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------- resample helpers
struct BinParams {
  const long long* ts;
  long long first, freq;
  double inv_freq;  // 1.0 / freq: quotient estimate, corrected exactly below (int64 division is ~100 instructions on CDNA)
  int closed_right;
  __device__ long long bin(int64_t i) const {
    long long x = ts[i] - first - (closed_right ? 1 : 0);  // >= 0: every timestamp is >= first (checked on the host)
    long long q = (long long)((double)x * inv_freq);
    long long r = x - q * freq;
    while (r < 0) { --q; r += freq; }
    while (r >= freq) { ++q; r -= freq; }
    return q;
  }
};
// sparse bins (many rows per bin): bin b starts at the first row whose timestamp is >= (closed-left) / > (closed-right) edge b
// (the search starts from the position a uniformly spaced axis would give and brackets the answer with growing steps before it
//  bisects: a few probes in neighbouring cache lines instead of ~30 scattered ones per edge on regular timestamps)
__global__ void k_bin_lower_bounds(BinParams p, int64_t n, int64_t nbins, long long tmin, long long tmax, uint32_t* __restrict__ lb /* nbins + 1 */) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const double scale = tmax > tmin ? (double)(n - 1) / (double)(tmax - tmin) : 0.0;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= nbins; b += stride) {
    if (b == nbins) {
      lb[b] = (uint32_t)n;
      continue;
    }
    long long edge = p.first + b * p.freq;
    auto before_at = [&](int64_t i) {
      const long long v = p.ts[i];
      return p.closed_right ? (v <= edge) : (v < edge);
    };
    int64_t g = (int64_t)((double)(edge - tmin) * scale);
    g = g < 0 ? 0 : (g > n - 1 ? n - 1 : g);
    int64_t lo = 0, hi = n;
    if (before_at(g)) {  // the answer lies behind g: step forward until a row is not before the edge
      lo = g + 1;
      for (int64_t st = 64; lo + st < n; st <<= 2) {
        if (!before_at(lo + st)) {
          hi = lo + st;
          break;
        }
        lo = lo + st + 1;
      }
    } else {  // the answer is g or in front of it
      hi = g;
      for (int64_t st = 64; hi - st > 0; st <<= 2) {
        if (before_at(hi - st)) {
          lo = hi - st + 1;
          break;
        }
        hi = hi - st;
      }
    }
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      long long v = p.ts[mid];
      bool before = p.closed_right ? (v <= edge) : (v < edge);
      if (before) lo = mid + 1;
      else hi = mid;
    }
    lb[b] = (uint32_t)lo;
  }
}
struct NonEmptyBinPred {
  const uint32_t* lb;
  __device__ bool operator()(int64_t b) const { return lb[b] < lb[b + 1]; }
};
struct NonEmptyBinEmit {
  const uint32_t* lb;
  long long label_base, freq;
  uint32_t* seg_start;
  int64_t* labels;
  int64_t* first_rows;
  __device__ void operator()(int64_t pos, int64_t b) const {
    seg_start[pos] = lb[b];
    labels[pos] = label_base + b * freq;
    first_rows[pos] = (int64_t)lb[b];
  }
};
struct BinStartPred {
  BinParams p;
  __device__ bool operator()(int64_t i) const { return i == 0 || p.bin(i) != p.bin(i - 1); }
};
struct BinStartEmit {
  BinParams p;
  long long label_base;  // first + label_right * freq
  uint32_t* seg_start;
  int64_t* labels;
  int64_t* first_rows;
  __device__ void operator()(int64_t pos, int64_t i) const {
    seg_start[pos] = (uint32_t)i;
    labels[pos] = label_base + p.bin(i) * p.freq;
    first_rows[pos] = i;
  }
};
__global__ void k_check_sorted(const long long* __restrict__ ts, int64_t n, unsigned int* __restrict__ bad) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; i < n; i += stride)
    if (ts[i] < ts[i - 1]) atomicExch(bad, 1u);
}
__global__ void k_row_labels(BinParams p, long long label_base, int64_t n, int64_t* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = label_base + p.bin(i) * p.freq;
}
__global__ void k_seg_row_ids(const uint32_t* __restrict__ seg_start, int64_t G, int64_t n, uint32_t* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int64_t lo = 0, hi = G;  // last segment with start <= i
    while (hi - lo > 1) {
      int64_t mid = (lo + hi) >> 1;
      if (seg_start[mid] <= (uint32_t)i) lo = mid;
      else hi = mid;
    }
    out[i] = (uint32_t)lo;
  }
}
__global__ void k_set_last(uint32_t* p, int64_t idx, uint32_t v) {
  if (threadIdx.x == 0 && blockIdx.x == 0) p[idx] = v;
}

}  // namespace pdx

using namespace pdx;

struct pdx_groupby {
  int mode = 0;  // 0 = hash group-by, 1 = contiguous segments (resample)
  int64_t n = 0, G = 0;
  int key_dtype = PDX_INT64;
  // hash mode
  uint32_t* gid_of_slot = nullptr;  // nslots entries
  int64_t nslots = 0;
  int slot_bits = 0;
  int dense = 0;                    // 1: slots are key - min (dense integer key domain), 0: open-addressing hash table
  // partitioned hash build (slot_of_row == nullptr): rows live in hash-partition order
  int part_bits = 0;                   // hash bits the rows are partitioned by (8, or 8 + digit2_bits after a second level)
  int special_slots = 0;               // 1: the null key and/or the INT64_MIN key occur (their two slots lie past the table)
  uint8_t* digit2 = nullptr;           // second-level digit of every row in FIRST-LEVEL order (very many groups only)
  uint32_t* part_off2 = nullptr;       // its scatter offsets [tiles][1 << digit2_bits]
  int digit2_bits = 0;
  uint8_t* bucket8 = nullptr;          // n, row order: low kPartBits = partition
  uint32_t* part_off = nullptr;     // [tiles][256] scatter offsets of the partition pass
  uint32_t* slot_part = nullptr;    // n, logical slot per partitioned position
  uint32_t* rows_part = nullptr;    // n, original row (bit 31: key is null)
  uint32_t* pass0_off = nullptr;    // row-order slots: scanned offsets of the first sort pass (fused into the slot kernel)
  uint32_t* slot_of_row = nullptr;  // n
  uint32_t* occ_slot = nullptr;     // G, slot order
  uint32_t* gid_of_occ = nullptr;   // G
  // both modes
  int64_t* uniques = nullptr;      // G (labels in resample mode)
  uint8_t* unique_ok = nullptr;    // G bytes
  int64_t* first_rows = nullptr;   // G
  // segments mode
  uint32_t* seg_start = nullptr;   // G + 1
  BinParams bin{};
  long long label_base = 0;
  mutable hipStream_t stream = nullptr;  // the stream of the last call that used the handle (pool frees are ordered behind it)
  std::vector<void*> owned;
  template <typename T>
  T* own(size_t count) {
    T* p = static_cast<T*>(pool_alloc((count ? count : 1) * sizeof(T)));
    if (p) owned.push_back(p);
    return p;
  }
  ~pdx_groupby() {
    StreamNote note(stream);
    pool_free_many(owned.data(), (int)owned.size());
  }
};

namespace pdx {

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
template <typename Alloc>
static int sort_values_narrow_full(pdx_groupby* gb, const uint64_t* vin, Alloc&& alloc, Scratch& s, hipStream_t st, const uint64_t** vals_sorted,
                                   uint32_t* seg_start_out, bool* done) {
  *done = false;
  const int64_t n = gb->n, G = gb->G;
  const SortPlan plan = make_sort_plan(gb->slot_bits, sort_max_bits());
  const bool env_ok = [] { const char* e = getenv("PDX_SORT_NARROW"); return !(e && e[0] == '0'); }();
  if (!env_ok || gb->slot_part || !gb->pass0_off || !gb->slot_of_row || plan.npasses != 3 || n < ((int64_t)1 << 22)) return PDX_OK;
  const int b0 = plan.bits[0], b1 = plan.bits[1], b2 = plan.bits[2];
  if (b0 > 8 || b1 > 8 || b2 > 8 || gb->slot_bits - b0 > 16 || b2 > 8 || gb->slot_bits != b0 + b1 + b2) return PDX_OK;
  const int64_t ntiles = ceil_div(n, kSortTile), nchunks = ceil_div(ntiles, kColChunk);
  uint16_t* k16 = s.get<uint16_t>((size_t)n);
  uint8_t* k8 = s.get<uint8_t>((size_t)n);
  uint32_t* hist = s.get<uint32_t>((size_t)ntiles << 8);
  uint32_t* chunk = s.get<uint32_t>((size_t)(nchunks + 1) << 8);
  uint32_t* starts1 = s.get<uint32_t>(((size_t)1 << (b0 + b1)) + 1);
  uint32_t* slot_start = s.get<uint32_t>(((size_t)1 << gb->slot_bits) + 1);
  PDX_SCRATCH_CHECK(s);
  uint64_t* v0 = static_cast<uint64_t*>(alloc((size_t)n * 8));
  uint64_t* v1 = static_cast<uint64_t*>(alloc((size_t)n * 8));
  if (!v0 || !v1) return PDX_OOM;
  PDX_TRY((narrow_pass<uint32_t, uint16_t>(b0, gb->slot_of_row, vin, k16, v0, n, gb->pass0_off, hist, chunk, st)));
  PDX_TRY((narrow_pass<uint16_t, uint8_t>(b1, k16, v0, k8, v1, n, nullptr, hist, chunk, st)));
  hipLaunchKernelGGL((k_level_starts<uint16_t>), dim3(1u << b0), dim3(256), 0, st, k16, n, gb->pass0_off, (int64_t)1 << b0, b0, b1, hist, starts1);
  PDX_TRY((narrow_pass<uint8_t, uint8_t>(b2, k8, v1, (uint8_t*)nullptr, v0, n, nullptr, hist, chunk, st)));
  hipLaunchKernelGGL((k_level_starts<uint8_t>), dim3((unsigned)std::min<int64_t>((int64_t)1 << (b0 + b1), 65536)), dim3(256), 0, st, k8, n, starts1,
                     (int64_t)1 << (b0 + b1), b0 + b1, b2, hist, slot_start);
  hipLaunchKernelGGL(k_seg_starts_from_slots, dim3(grid_for(G + 1, 256)), dim3(256), 0, st, slot_start, n, gb->occ_slot, G, seg_start_out);
  PDX_LAUNCH_CHECK();
  *vals_sorted = v0;
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
    const int64_t maxB = nrows / kBigSeg + 1;  // a long group has more than kBigSeg rows
    uint32_t* big_idx = s.get<uint32_t>((size_t)std::min<int64_t>(nseg, maxB));
    PDX_SCRATCH_CHECK(s);
    int64_t B = 0;
    PDX_TRY(compact_indices(nseg, BigPred{seg_start}, BigEmit{big_idx}, &B, s, st));
    if (B > 0) {
      const int64_t max_items = nrows / kBigSeg + B;
      int64_t* item_off = s.get<int64_t>((size_t)B + 1);
      SubState<T>* state = s.get<SubState<T>>((size_t)max_items);
      PDX_SCRATCH_CHECK(s);
      hipLaunchKernelGGL(k_big_offsets, dim3(1), dim3(256), 0, st, seg_start, big_idx, B, item_off);
      const int grid_sub = (int)std::min<int64_t>(ceil_div(max_items, kSegWaves), (int64_t)kCUs * 8);
#define SEG_SUB(PW, MM, IS)                                                                                                                   \
  hipLaunchKernelGGL((k_seg_reduce_sub<T, PW, MM, IS>), dim3(grid_sub), dim3(kSegWaves * 64), 0, st, vals, seg_start, big_idx, item_off, B, state); \
  hipLaunchKernelGGL((k_seg_combine_big<T, PW, MM, IS>), dim3((unsigned)B), dim3(64), 0, st, seg_start, big_idx, item_off, B, state, \
                     out_index, o)
      SEG_DISPATCH(SEG_SUB)
#undef SEG_SUB
      PDX_LAUNCH_CHECK();
    }
  }
  int grid = (int)std::min<int64_t>(ceil_div(nseg, kSegWaves), (int64_t)kCUs * 8);
  dim3 g(grid), b(kSegWaves * 64);
  // mostly short groups: one kernel that batches the groups of <= kMidLen rows per wave and chunks through the longer ones
  const int64_t mid_max = [] { const char* e = getenv("PDX_SEG_MID_MAX"); return e ? atoll(e) : 1100ll; }();
  const int64_t min_len = -1;
  if (nrows / nseg < mid_max) {
    const int64_t nwaves = (int64_t)kCUs * 8 * kSegWaves;
    const int64_t gpw = std::max<int64_t>(64, ceil_div(nseg, nwaves));
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

// ---------------------------------------------------------------- fused last digit: the final sort pass and the reduce in one kernel.
// After the LSD passes over the low L = B - 6 slot bits, the rows of one "run" (equal low bits) hold at most 64 groups -- the
// values of the top 6 bits -- interleaved in row order.  Instead of one more 24 B/row scatter pass followed by an 8 B/row reduce,
// one workgroup per run ranks every 4096-row tile stably by the top digit in LDS (the scatter kernel's ballot ranking) and wave 0
// replays Arrow's leaf / binary-counter recurrence LITERALLY with one lane per group (state in registers + one LDS column per
// lane); the other waves already hold the next tile's loads.  Reads 12 B/row once.  Value nulls are the key's bit 31: a null row
// closes the open leaf, exactly Arrow's restart rule -- no separate nullable kernel on this path.
constexpr int kFlrBits = 6;
constexpr int kFlrLevels = 20;  // a group lies inside one run, a run is <= 2^19 rows (checked by the host), and with nulls a leaf can
                                // be a single row: <= 2^19 leaves
constexpr int kFlrItems = 10;   // rows per thread and tile: 3072-row tiles (measured best: 4096 -> 5.6 ms, 3072 -> 4.2 ms, 2048 -> 4.5 ms per 1e9 rows)
constexpr int kFlrTile = kSortBlock * kFlrItems;
__global__ void k_run_starts(const uint32_t* __restrict__ sorted_keys, int64_t n, int low_bits, int64_t nruns, uint32_t* __restrict__ run_start,
                             unsigned int* __restrict__ max_len) {
  const uint32_t lmask = (1u << low_bits) - 1u;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= nruns; r += stride) {
    int64_t lo = 0, hi = n;
    if (r < nruns) {
      while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)(sorted_keys[mid] & lmask) < r) lo = mid + 1;
        else hi = mid;
      }
    } else {
      lo = n;
    }
    run_start[r] = (uint32_t)lo;
  }
  (void)max_len;
}
__global__ void k_run_max_len(const uint32_t* __restrict__ run_start, int64_t nruns, unsigned int* __restrict__ max_len) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned int m = 0;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nruns; r += stride) {
    unsigned int len = run_start[r + 1] - run_start[r];
    m = len > m ? len : m;
  }
  for (int d = 32; d >= 1; d >>= 1) {
    unsigned int o = __shfl_xor(m, d, 64);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(max_len, m);
}
// (x - mean)^2 with x86 NaN operand propagation (k_seg_sqdev)
__device__ __forceinline__ double flr_sqdev(double v, double mu) {
  const double x = v - mu;
  return v != v ? v : (mu != mu ? mu : x * x);
}
// Arrow's binary counter, one LDS column per lane: push a finished leaf sum
__device__ __forceinline__ void flr_counter_push(double (*csum)[1 << kFlrBits], int lane, unsigned long long& cmask, int& root, double leaf) {
  int cur = 0;
  unsigned long long m = 1;
  double v = csum[0][lane] + leaf;
  cmask ^= m;
  while ((cmask & m) == 0) {
    csum[cur][lane] = 0.0;
    ++cur;
    m <<= 1;
    v = csum[cur][lane] + v;
    cmask ^= m;
  }
  csum[cur][lane] = v;
  root = cur > root ? cur : root;
}
}  // namespace pdx
#include "flr_wave.hpp"
namespace pdx {
// Segmented "run of valid rows" state of a chunk of staged rows, packed in 32 bits, for the nullable leaf phase: bits 0-11 valid rows at
// the chunk's end since its last break, bit 12 the chunk holds a break (a null row or a group boundary), bits 13-14 the kind of its
// last break (1 null, 2 group boundary), bits 16-22 group boundaries in the chunk.  Associative, earlier operand first.
struct RunStateOp {
  template <typename U>
  __device__ static U identity() { return U(0); }
  __device__ uint32_t operator()(uint32_t a, uint32_t b) const {
    const uint32_t nh = ((a >> 16) + (b >> 16)) << 16;
    if (b & 0x1000u) return (b & 0xFFFFu) | nh;
    return (((a & 0xFFFu) + (b & 0xFFFu)) & 0xFFFu) | (a & 0x7000u) | nh;
  }
};
// KT: uint32 slots (top digit at bit low_bits, bit 31 = the value's null flag) or, after a narrowing sort, the top digit alone in a byte
// NULL_PW (host: nullable values, sum / mean / count only): leaves restart at every null, so they are data dependent; a segmented
// scan over the staged rows finds every leaf's first row, ONE THREAD PER LEAF sums it (<= 16 rows) and leaves the sum and a marker
// byte in place, then one lane per group walks its leaves in order for the counter pushes (instead of one lane per group adding
// up all of its rows one by one).  Bit-exact, but not faster yet (11.3 vs 10.7 ms per 1e9 rows at 5 % nulls): opt-in.
template <typename T, bool DENSE_PW, typename KT = uint32_t, bool NULL_PW = false>
__global__ void __launch_bounds__(kSortBlock) k_flr_reduce(const KT* __restrict__ keys, const T* __restrict__ vals,
                                                           const uint32_t* __restrict__ run_start, int64_t nruns, int low_bits,
                                                           const uint32_t* __restrict__ gid_of_slot, SegOut out, uint8_t* __restrict__ ok,
                                                           int want_pw, int want_mm, int want_is, int nullable,
                                                           const double* __restrict__ sqdev_mean) {
  // sqdev_mean != nullptr (second pass of variance): every value x of group g enters the sum as (x - sqdev_mean[g])^2, with the
  // reference's x86 NaN propagation (see k_seg_sqdev)
  constexpr int R = 1 << kFlrBits;
  // staged rows of digit d start at dstart[d] + d: the digits' regions are ~64 rows = 512 B apart, so without the skew the 64
  // lanes of the replay (one digit each) would hit the same LDS bank on every read (measured: 3x slower)
  __shared__ T svals[kFlrTile + R];
  __shared__ __attribute__((aligned(8))) uint8_t snull[kFlrTile + R];
  __shared__ uint32_t cnt[kSortWaves][R];
  __shared__ unsigned long long match[kSortWaves][R];  // match-any words of the ranking (wave_match_rank)
  __shared__ uint32_t dstart[R + 1];
  __shared__ double csum[kFlrLevels][R];
  // dense sum/mean/count fast path (no nulls, no min/max/int sum): one THREAD per 16-value leaf, then one lane per group for the
  // few counter pushes -- the open leaf of every group (rows so far + their sequential sum) lives in LDS between tiles
  __shared__ int open_pos[R];
  __shared__ double open_acc[R];
  __shared__ double mu_s[R];
  __shared__ int lp[R + 1];
  __shared__ uint32_t run_smem[8];
  double* leafsum = reinterpret_cast<double*>(snull);  // (the null flags are unused on this path: room for (tile + 64) / 8 leaf sums)
  constexpr bool dense_pw = DENSE_PW;  // host: want_pw && !want_mm && !want_is && !nullable (a separate instantiation: fewer live registers)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  // Barriers per tile: after the ranking, after the prefixes, after the staging and (dense path) after the leaf sums.  The digit
  // counters are re-zeroed right after the staging barrier, and wave 0's replay needs no closing barrier: the next tile's
  // staging lies behind two barriers that wave 0 itself has to reach.
  for (int d = tid; d < kSortWaves * R; d += kSortBlock) {
    (&cnt[0][0])[d] = 0;
    (&match[0][0])[d] = 0;
  }
  __syncthreads();
  for (int64_t run = blockIdx.x; run < nruns; run += gridDim.x) {
    const int64_t s = run_start[run], e = run_start[run + 1];
    if (s == e) continue;
    if (tid < R) {
      open_pos[tid] = 0;
      open_acc[tid] = 0.0;
      mu_s[tid] = 0.0;
    }
    double mu = 0.0;
    bool mu_known = false;
    // per-group state (wave 0, lane = top digit)
    double acc = 0.0;
    int pos = 0, root = 0;
    unsigned long long cmask = 0, isum = 0;
    long long nvalid = 0, nrows = 0;
    T vmn = T(0), vmx = T(0);
    int zneg = -1;  // sign of the last zero-valued valid row (-1: none)
    bool has = false;
    if (wave == 0)
      for (int l = 0; l < kFlrLevels; ++l) csum[l][lane] = 0.0;
    uint32_t key[kFlrItems];
    T val[kFlrItems];
    auto load_tile = [&](int64_t t0) {
      const int rows = (int)(e - t0 < kFlrTile ? e - t0 : kFlrTile);
#pragma unroll
      for (int q = 0; q < kFlrItems; ++q) {
        const int r = wave * (64 * kFlrItems) + q * 64 + lane;
        if (r < rows) {
          key[q] = keys[t0 + r];
          val[q] = vals[t0 + r];
        } else {
          key[q] = 0;
          val[q] = T(0);
        }
      }
    };
    load_tile(s);
    for (int64_t t0 = s; t0 < e; t0 += kFlrTile) {
      const int rows = (int)(e - t0 < kFlrTile ? e - t0 : kFlrTile);
      uint32_t rank[kFlrItems];
#pragma unroll
      for (int q = 0; q < kFlrItems; ++q) {
        const int r = wave * (64 * kFlrItems) + q * 64 + lane;
        const bool active = r < rows;
        const uint32_t d = sizeof(KT) == 4 ? ((key[q] & kSortKeyMask) >> low_bits) & (R - 1) : key[q] & (R - 1);
        rank[q] = wave_match_rank(match[wave], cnt[wave], d, active, lane, lt_mask);
      }
      __syncthreads();
      if (tid < R) {  // exclusive prefix over waves per digit, then over digits (64 values: one wave)
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
          const uint32_t c = cnt[w][tid];
          cnt[w][tid] = tot;
          tot += c;
        }
        const uint32_t inc = wave_inclusive_scan(tot, SumOp());
        const uint32_t ex = inc - tot;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) cnt[w][tid] += ex;
        dstart[tid] = ex;
        if (tid == R - 1) dstart[R] = inc;
        if (NULL_PW) {
          snull[inc + tid] = 2;  // the unused slot behind this group's staged rows: a group boundary for the scan below
          const int c = (int)tot;
          if (sqdev_mean && c > 0 && !mu_known) {
            mu = sqdev_mean[gid_of_slot[((uint32_t)lane << low_bits) | (uint32_t)run]];
            mu_s[lane] = mu;
            mu_known = true;
          }
          nrows += c;
        }
        if (dense_pw) {
          // leaves touched by this tile, per group: the first one may continue the open leaf, the last one may stay open
          const int c = (int)tot;
          if (sqdev_mean && c > 0 && !mu_known) {
            mu = sqdev_mean[gid_of_slot[((uint32_t)lane << low_bits) | (uint32_t)run]];
            mu_s[lane] = mu;
            mu_known = true;
          }
          const int nl = c > 0 ? (open_pos[lane] + c + 15) >> 4 : 0;
          const int incl = wave_inclusive_scan(nl, SumOp());
          lp[lane] = incl - nl;
          if (lane == R - 1) lp[R] = incl;
          nrows += c;
          nvalid += c;
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < kFlrItems; ++q) {
        const int r = wave * (64 * kFlrItems) + q * 64 + lane;
        if (r < rows) {
          const uint32_t d = sizeof(KT) == 4 ? ((key[q] & kSortKeyMask) >> low_bits) & (R - 1) : key[q] & (R - 1);
          const uint32_t p = cnt[wave][d] + rank[q] + d;
          svals[p] = val[q];
          if (nullable) snull[p] = (uint8_t)(key[q] >> (8 * (int)sizeof(KT) - 1));
        }
      }
      if (t0 + kFlrTile < e) load_tile(t0 + kFlrTile);  // in flight while wave 0 replays this tile
      __syncthreads();
      for (int d = tid; d < kSortWaves * R; d += kSortBlock) (&cnt[0][0])[d] = 0;  // (free again: the bases were consumed above)
      if (dense_pw) {
        const int NL = lp[R];
        for (int Lf = tid; Lf < NL; Lf += kSortBlock) {
          int lo = 0, hi = R - 1;
          while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (lp[mid] <= Lf) lo = mid;
            else hi = mid - 1;
          }
          const int d = lo, j = Lf - lp[d];
          const int p0 = open_pos[d], c = (int)(dstart[d + 1] - dstart[d]);
          const int r0 = j == 0 ? 0 : 16 * j - p0;
          int r1 = 16 * (j + 1) - p0;
          r1 = r1 < c ? r1 : c;
          double a = (j == 0 && p0 > 0) ? open_acc[d] : 0.0;
          const T* v = svals + dstart[d] + d;
          // all (<= 16) values of the leaf are requested before the first add: a loop that loads, waits and adds row by row pays
          // one LDS round trip per row on the critical path of the tile
          double xs[16];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int r = r0 + q < r1 ? r0 + q : r1 - 1;
            xs[q] = seg_to_f64(v[r]);
          }
          if (sqdev_mean) {
            const double m = mu_s[d];
#pragma unroll
            for (int q = 0; q < 16; ++q)
              if (r0 + q < r1) a += flr_sqdev(xs[q], m);
          } else {
#pragma unroll
            for (int q = 0; q < 16; ++q)
              if (r0 + q < r1) a += xs[q];
          }
          leafsum[Lf] = a;
        }
        __syncthreads();
        if (wave == 0) {
          const int c = (int)(dstart[lane + 1] - dstart[lane]);
          if (c > 0) {
            const int p0 = open_pos[lane];
            const int nl = (p0 + c + 15) >> 4, nfull = (p0 + c) >> 4;
            for (int j = 0; j < nfull; ++j) flr_counter_push(csum, lane, cmask, root, leafsum[lp[lane] + j]);
            const int rem = (p0 + c) & 15;
            open_pos[lane] = rem;
            if (rem) open_acc[lane] = leafsum[lp[lane] + nl - 1];
            pos = rem;
            acc = rem ? leafsum[lp[lane] + nl - 1] : 0.0;
          }
        }
      } else if (NULL_PW) {
        constexpr int CH = (kFlrTile + R + kSortBlock - 1) / kSortBlock;  // staged slots per thread
        const int L = rows + R;                                            // staged slots of this tile (rows + one boundary per group)
        uint8_t f[CH];
        uint32_t st = 0;
        {
          uint32_t tl = 0, hb = 0, lt = 0, nh = 0;
#pragma unroll
          for (int k = 0; k < CH; ++k) {
            const int pp = tid * CH + k;
            f[k] = pp < L ? snull[pp] : (uint8_t)2;
            if (f[k]) {
              tl = 0;
              hb = 1;
              lt = f[k];
              nh += f[k] == 2;
            } else {
              ++tl;
            }
          }
          st = tl | (hb << 12) | (lt << 13) | (nh << 16);
        }
        uint32_t tot_unused;
        const uint32_t ex = block_exclusive_scan(st, RunStateOp(), &tot_unused, run_smem);  // (two barriers: every flag byte has been read)
        {
          // first rows of leaves in this thread's chunk (bit k of `starts`) and group boundaries (bit k of `holes`): registers only.
          // The leaves themselves are summed in a second loop over the set bits, so a wave runs the 16-row loop once per leaf of its
          // busiest lane and not once per chunk slot
          int run_idx = (int)(ex & 0xFFFu);
          const int d0 = (int)(ex >> 16);
          int d = d0;
          const int type0 = (ex & 0x1000u) ? (int)((ex >> 13) & 3u) : 2;  // nothing in front: slot 0 starts group 0
          int q_cur = (type0 == 2 && d < R) ? open_pos[d] : 0;
          uint32_t starts = 0, holes = 0;
#pragma unroll
          for (int k = 0; k < CH; ++k) {
            if (f[k] == 0) {
              if (run_idx == 0 || ((q_cur + run_idx) & 15) == 0) starts |= 1u << k;
              ++run_idx;
            } else {
              run_idx = 0;
              if (f[k] == 2) {
                holes |= 1u << k;
                ++d;
                q_cur = d < R ? open_pos[d] : 0;
              } else {
                q_cur = 0;
              }
            }
          }
          while (starts) {
            const int k = __ffs((int)starts) - 1;
            starts &= starts - 1;
            const int pp = tid * CH + k;
            const int dd = d0 + __popc(holes & ((1u << k) - 1u));
            // the open leaf of the previous tile continues only on the group's first staged row
            const int q0 = (dd < R && pp == (int)dstart[dd] + dd) ? open_pos[dd] : 0;
            double a = q0 > 0 ? open_acc[dd] : 0.0;
            const double m = sqdev_mean ? mu_s[dd < R ? dd : 0] : 0.0;
            // all 16 flag bytes and values are requested before any is looked at (a loop that stops at the first null would pay
            // two dependent LDS round trips per row); slots behind the leaf's end may already hold another leaf's marker or sum:
            // they are never used (the first nonzero flag inside the leaf's 16 - q0 slots is an untouched null or boundary)
            uint8_t ffl[16];
            double xs[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              const int pj = pp + j < L ? pp + j : L - 1;
              ffl[j] = j == 0 ? (uint8_t)0 : (pp + j < L ? snull[pj] : (uint8_t)2);
              xs[j] = seg_to_f64(svals[pj]);
            }
            int count = 16 - q0;
            uint8_t term = 0;
#pragma unroll
            for (int j = 15; j >= 1; --j)
              if (j < 16 - q0 && ffl[j]) {
                count = j;
                term = ffl[j];
              }
#pragma unroll
            for (int j = 0; j < 16; ++j)
              if (j < count) a += sqdev_mean ? flr_sqdev(xs[j], m) : xs[j];
            const bool closed = q0 + count == 16 || term == 1;  // full, or cut by a null row
            reinterpret_cast<double*>(svals)[pp] = a;
            snull[pp] = (uint8_t)(0x80 | (closed ? 0x40 : 0) | (q0 + count - 1));
          }
        }
        __syncthreads();
        if (wave == 0) {
          const int i0 = (int)dstart[lane] + lane, i1 = (int)dstart[lane + 1] + lane;
          // Pass A: walk the group's leaves in row order; finished leaves are written back compactly over the slots already consumed
          // (every step consumes at least one slot and emits at most one leaf).  The pushes come afterwards, as in the literal replay.
          int nleaf = 0;
          for (int ip = i0; ip < i1;) {
            const uint8_t b = snull[ip];
            if (b & 0x80) {
              const int fill = (b & 15) + 1;
              const int cnt_rows = fill - (ip == i0 ? pos : 0);
              const double sum = reinterpret_cast<const double*>(svals)[ip];
              nvalid += cnt_rows;
              if (b & 0x40) {
                reinterpret_cast<double*>(svals)[i0 + nleaf++] = sum;
                pos = 0;
              } else {
                pos = fill;
                acc = sum;
              }
              ip += cnt_rows;
            } else {  // a null row: it closes the leaf left open by the previous tile (only possible on the group's first row)
              if (pos > 0) {
                reinterpret_cast<double*>(svals)[i0 + nleaf++] = acc;
                pos = 0;
              }
              ++ip;
            }
          }
          if (i1 > i0) {
            open_pos[lane] = pos;
            open_acc[lane] = acc;
          }
          for (int j = 0; j < nleaf; ++j) flr_counter_push(csum, lane, cmask, root, reinterpret_cast<const double*>(svals)[i0 + j]);
        }
      } else if (!dense_pw && wave == 0) {
        const int i0 = (int)dstart[lane] + lane, i1 = (int)dstart[lane + 1] + lane;
        nrows += i1 - i0;
        if (sqdev_mean && i1 > i0 && !mu_known) {
          mu = sqdev_mean[gid_of_slot[((uint32_t)lane << low_bits) | (uint32_t)run]];
          mu_known = true;
        }
        // Pass A: leaf sums only.  Finished leaves are written back over the rows already consumed (a leaf has >= 1 row, so the
        // write index never passes the read index).  The counter pushes are NOT done here: lanes finish leaves at different
        // rows, so a push inside this loop would make the whole wave walk the (long) push path on nearly every row.
        int nleaf = 0;
        for (int ib = i0; ib < i1; ib += 8) {
          T xb[8];
          uint8_t nb[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = ib + u < i1 ? ib + u : i1 - 1;
            xb[u] = svals[i];
            nb[u] = nullable ? snull[i] : (uint8_t)0;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (ib + u >= i1) break;
            const T x = xb[u];
            const bool isnull = nb[u] != 0;
            bool close = false;
            if (!isnull) {
              ++nvalid;
              if (want_pw) {
                acc = (pos == 0 ? 0.0 : acc) + (sqdev_mean ? flr_sqdev(seg_to_f64(x), mu) : seg_to_f64(x));
                close = ++pos == 16;
              }
              if (want_is) isum += (unsigned long long)x;
              if (want_mm && x == x) {
                if (!has) { vmn = vmx = x; has = true; }
                else {
                  if (x < vmn) vmn = x;
                  if (x > vmx) vmx = x;
                }
                if constexpr (__is_same(T, double)) {
                  if (x == 0.0) zneg = __double_as_longlong(x) < 0 ? 1 : 0;  // the LAST zero of the group (rows are replayed in order)
                }
              }
            } else {
              close = want_pw && pos > 0;  // a null row closes the open leaf
            }
            if (close) {
              reinterpret_cast<double*>(svals)[i0 + nleaf++] = acc;
              pos = 0;
            }
          }
        }
        // Pass B: Arrow's binary counter over this tile's finished leaves (a handful per lane)
        for (int j = 0; j < nleaf; ++j) flr_counter_push(csum, lane, cmask, root, reinterpret_cast<const double*>(svals)[i0 + j]);
      }
    }
    if (wave == 0 && nrows > 0) {
      const uint32_t slot = ((uint32_t)lane << low_bits) | (uint32_t)run;
      const uint32_t oi = gid_of_slot[slot];
      if (want_pw) {
        if (pos > 0) flr_counter_push(csum, lane, cmask, root, acc);
        double total = 0.0;
        if (nvalid > 0) {
          double a = csum[0][lane];
          for (int i = 1; i <= root; ++i) a = csum[i][lane] + a;
          total = a;
        }
        if (out.sum_f) out.sum_f[oi] = total;
        if (out.mean) out.mean[oi] = nvalid ? total / (double)nvalid : 0.0;
      }
      if (want_is && out.sum_i) out.sum_i[oi] = (long long)isum;
      if (want_mm) {
        T nanv = T(0);
        if constexpr (__is_same(T, double)) nanv = __builtin_nan("");
        if (out.vmin) static_cast<T*>(out.vmin)[oi] = has ? vmn : nanv;
        if constexpr (__is_same(T, double)) {  // a group WITH nulls keeps the last of tied zero maxima (minmax.hpp)
          if (has && zneg >= 0 && vmx == 0.0 && nvalid < nrows) vmx = zneg ? -0.0 : 0.0;
        }
        if (out.vmax) static_cast<T*>(out.vmax)[oi] = has ? vmx : nanv;
      }
      if (out.count) out.count[oi] = nvalid;
      if (ok) ok[oi] = nvalid > 0;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- "next" aggregations on the grouped layout
__device__ __forceinline__ bool seg_row_is_null(const uint32_t* sorted_keys, const uint8_t* row_valid, int64_t valid_off, int64_t i) {
  if (sorted_keys) return (sorted_keys[i] >> 31) != 0;          // grouped (sorted) layout: the flag travelled with the slot
  return row_valid && !bit_get(row_valid, valid_off + i);       // segments of the original order (resample)
}
// d[i] = (x[i] - mean of x's segment)^2: the second pass of Arrow's variance.  One wave per segment, coalesced.
template <typename T>
__global__ void __launch_bounds__(256) k_seg_sqdev(const T* __restrict__ vals, const uint32_t* __restrict__ seg_start, int64_t nseg,
                                                   const double* __restrict__ mean_seg, double* __restrict__ d) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); k < nseg; k += nw) {
    const int64_t s = seg_start[k], e = seg_start[k + 1];
    const double mu = mean_seg[k];
    for (int64_t i = s + lane; i < e; i += 64) {
      // NaN operands: x86 SUBSD/MULSD hand back the first NaN operand unchanged, v_add_f64 with a negated source flips its sign;
      // spell the x86 result out so the NaN bits agree too
      const double v = (double)vals[i];
      const double x = v - mu;
      d[i] = v != v ? v : (mu != mu ? mu : x * x);
    }
  }
}
// product of the valid values of every segment in row order (sequential by definition: one multiply chain per group); first /
// last row of every segment.  One wave per segment: 1024 values at a time are loaded coalesced into LDS (null rows as the
// multiplicative identity), lane 0 runs the chain -- the loads, not the chain, bound the kernel.
template <typename T>
__global__ void __launch_bounds__(256) k_seg_product_first_last(const T* __restrict__ vals, const uint32_t* __restrict__ sorted_keys,
                                                                const uint8_t* __restrict__ row_valid, int64_t valid_off,
                                                                const uint32_t* __restrict__ seg_start, int64_t nseg,
                                                                const uint32_t* __restrict__ out_index, T* __restrict__ prod,
                                                                uint8_t* __restrict__ prod_ok, T* __restrict__ first, uint8_t* __restrict__ first_ok,
                                                                T* __restrict__ last, uint8_t* __restrict__ last_ok) {
  __shared__ T stage_all[4][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* stage = stage_all[wave];
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t k = (int64_t)blockIdx.x * 4 + wave; k < nseg; k += nw) {
    const int64_t s = seg_start[k], e = seg_start[k + 1];
    const uint32_t oi = out_index ? out_index[k] : (uint32_t)k;
    if (prod) {
      T p = T(1);
      bool any = false;
      for (int64_t c0 = s; c0 < e; c0 += 1024) {
        const int cl = (int)(e - c0 < 1024 ? e - c0 : 1024);
        bool mine = false;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int idx = q * 64 + lane;
          if (idx < cl) {
            const bool isnull = seg_row_is_null(sorted_keys, row_valid, valid_off, c0 + idx);
            stage[idx] = isnull ? T(1) : vals[c0 + idx];
            mine |= !isnull;
          }
        }
        any |= __any(mine);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
          int i = 0;
          for (; i + 16 <= cl; i += 16) {  // the 16 LDS reads are issued together; only the multiplies form the chain
            T x[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) x[q] = stage[i + q];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              if constexpr (__is_same(T, double)) p = p * x[q];
              else p = (T)((unsigned long long)p * (unsigned long long)x[q]);
            }
          }
          for (; i < cl; ++i) {
            if constexpr (__is_same(T, double)) p = p * stage[i];
            else p = (T)((unsigned long long)p * (unsigned long long)stage[i]);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (lane == 0) {
        prod[oi] = p;
        if (prod_ok) prod_ok[oi] = any;
      }
    }
    if (lane == 0) {
      if (first) {
        first[oi] = e > s ? vals[s] : T(0);
        if (first_ok) first_ok[oi] = e > s && !seg_row_is_null(sorted_keys, row_valid, valid_off, s);
      }
      if (last) {
        last[oi] = e > s ? vals[e - 1] : T(0);
        if (last_ok) last_ok[oi] = e > s && !seg_row_is_null(sorted_keys, row_valid, valid_off, e - 1);
      }
    }
  }
}
// var = m2 / count (ddof = 0); stddev = sqrt(var)
__global__ void k_var_finish(const double* __restrict__ m2, const long long* __restrict__ count, int64_t G, double* __restrict__ var,
                             double* __restrict__ sd) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < G; g += stride) {
    const double v = count[g] > 0 ? m2[g] / (double)count[g] : 0.0;
    if (var) var[g] = v;
    if (sd) sd[g] = v != v ? v : sqrt(v);  // a NaN variance passes through unchanged (x86 sqrtsd keeps the operand's NaN bits)
  }
}

// ---------------------------------------------------------------- nullable values in very long groups.
// One wave per group is hopeless for a group of 1e8 rows with nulls (leaves restart at every run of valid rows, so the work cannot
// be cut into aligned sub-segments the way dense values are).  The grouped values of such a group are one contiguous slice: the
// whole-column kernels (pdx_aggregate: window scan + pairwise tree levels, all workgroups on one slice) reduce it exactly.
struct HugePred {
  const uint32_t* seg_start;
  __device__ bool operator()(int64_t k) const { return (int64_t)seg_start[k + 1] - (int64_t)seg_start[k] > kHugeNullable; }
};
struct HugeEmit {
  const uint32_t* seg_start;
  const uint32_t* out_index;
  int64_t* rec;  // [3 * pos]: start, end, output index
  __device__ void operator()(int64_t pos, int64_t k) const {
    rec[3 * pos] = seg_start[k];
    rec[3 * pos + 1] = seg_start[k + 1];
    rec[3 * pos + 2] = out_index ? out_index[k] : k;
  }
};
// validity bitmap of the grouped layout from the flag bit that travelled with the slots
__global__ void k_flags_to_bitmap(const uint32_t* __restrict__ sorted_keys, int64_t n, uint64_t* __restrict__ words) {
  const int lane = threadIdx.x & 63;
  const int64_t nwords = (n + 63) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < nwords; w += nw) {
    const int64_t i = (w << 6) + lane;
    const uint64_t bal = __ballot(i < n && !(sorted_keys[i] >> 31));
    if (lane == 0) words[w] = bal;
  }
}
static int reduce_huge_nullable_groups(const void* vals, int value_dtype, const uint32_t* sorted_flag_keys, const uint8_t* row_valid, int64_t valid_off,
                                       const uint32_t* seg_start, int64_t nseg, const uint32_t* out_index, int64_t nrows, const SegOut& o, uint8_t* ok,
                                       Scratch& s, hipStream_t st) {
  if (nrows <= kHugeNullable) return PDX_OK;
  const int64_t maxH = nrows / kHugeNullable + 1;
  int64_t* rec = s.get<int64_t>((size_t)3 * (size_t)std::min<int64_t>(nseg, maxH));
  PDX_SCRATCH_CHECK(s);
  int64_t H = 0;
  PDX_TRY(compact_indices(nseg, HugePred{seg_start}, HugeEmit{seg_start, out_index, rec}, &H, s, st));
  if (H == 0) return PDX_OK;
  std::vector<int64_t> hrec((size_t)3 * (size_t)H);
  PDX_HIP(hipMemcpyAsync(hrec.data(), rec, hrec.size() * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  const uint8_t* bitmap = row_valid;
  int64_t bitmap_off = valid_off;
  if (sorted_flag_keys) {
    uint64_t* words = s.get<uint64_t>((size_t)((nrows + 63) >> 6) + 2);
    PDX_SCRATCH_CHECK(s);
    hipLaunchKernelGGL(k_flags_to_bitmap, dim3(grid_for(nrows, 256)), dim3(256), 0, st, sorted_flag_keys, nrows, words);
    PDX_LAUNCH_CHECK();
    bitmap = reinterpret_cast<const uint8_t*>(words);
    bitmap_off = 0;
  }
  PDX_HIP(hipStreamSynchronize(st));
  for (int64_t h = 0; h < H; ++h) {
    const int64_t start = hrec[3 * h], end = hrec[3 * h + 1], oi = hrec[3 * h + 2];
    pdx_column col{};
    col.dtype = value_dtype;
    col.length = end - start;
    col.offset = bitmap_off + start;  // values and bitmap share the element offset: rebase the values pointer instead
    col.null_count = -1;
    col.validity = bitmap;
    col.values = static_cast<const uint8_t*>(vals) - (size_t)bitmap_off * 8;
    auto put = [&](void* dst, const void* src, size_t bytes) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st); };
    pdx_scalar sc{};
    uint8_t valid_group = 0;
    if (o.sum_f || o.mean || o.sum_i) {
      if (o.sum_f || o.sum_i) {
        PDX_TRY(pdx_aggregate(PDX_AGG_SUM, &col, &sc, st));
        valid_group = (uint8_t)sc.is_valid;
        if (o.sum_f) PDX_HIP(put(o.sum_f + oi, &sc.v.f64, 8));
        if (o.sum_i) PDX_HIP(put(o.sum_i + oi, &sc.v.i64, 8));
      }
      if (o.mean) {
        PDX_TRY(pdx_aggregate(PDX_AGG_MEAN, &col, &sc, st));
        valid_group = (uint8_t)sc.is_valid;
        PDX_HIP(put(o.mean + oi, &sc.v.f64, 8));
      }
    }
    if (o.vmin) {
      PDX_TRY(pdx_aggregate(PDX_AGG_MIN, &col, &sc, st));
      valid_group = (uint8_t)sc.is_valid;
      PDX_HIP(put(static_cast<uint8_t*>(o.vmin) + 8 * oi, &sc.v, 8));
    }
    if (o.vmax) {
      PDX_TRY(pdx_aggregate(PDX_AGG_MAX, &col, &sc, st));
      valid_group = (uint8_t)sc.is_valid;
      PDX_HIP(put(static_cast<uint8_t*>(o.vmax) + 8 * oi, &sc.v, 8));
    }
    PDX_TRY(pdx_aggregate(PDX_AGG_COUNT, &col, &sc, st));
    if (!(o.sum_f || o.mean || o.sum_i || o.vmin || o.vmax)) valid_group = sc.v.i64 > 0;
    if (o.count) PDX_HIP(put(o.count + oi, &sc.v.i64, 8));
    if (ok) PDX_HIP(put(ok + oi, &valid_group, 1));
    PDX_HIP(hipStreamSynchronize(st));  // (the staged host scalars above must outlive their copies)
  }
  return PDX_OK;
}

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
  // ---- dense-domain fast path: valid keys span a small integer range -> slot = key - min (or its residue form), no table
  Slot* table = nullptr;
  unsigned int* dense_first = nullptr;
  long long dense_min = 0;
  unsigned int dense_mask = 0;
  unsigned int null_slot = 0;
  int64_t nslots = 0;
  const char* denv = getenv("PDX_GROUPBY_DENSE");
  const bool allow_dense = !(denv && denv[0] == '0');
  const bool lds_ok = [] { const char* e = getenv("PDX_DENSE_LDS"); return !(e && e[0] == '0'); }();
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
    int64_t prefix = std::min<int64_t>(n, std::max<int64_t>((int64_t)1 << 22, 16 * nslots));
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
  const bool spec_ok = [] { const char* e = getenv("PDX_DENSE_SPECULATE"); return !(e && e[0] == '0'); }();
  if (allow_dense && spec_ok && lds_ok && n >= ((int64_t)1 << 23)) {
    KeyRange* dsample = s.get<KeyRange>(64);
    if (s.failed) return PDX_OOM;
    hipLaunchKernelGGL(k_sample_key_range, dim3(64), dim3(1024), 0, st, keys, valid, key->offset, n, dsample);
    KeyRange hsv[64];
    PDX_HIP(hipMemcpyAsync(hsv, dsample, sizeof(hsv), hipMemcpyDeviceToHost, st));
    PDX_HIP(hipStreamSynchronize(st));
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
    gb->rows_part = gb->own<uint32_t>((size_t)n);
    uint32_t* chunk_sum = s.get<uint32_t>((size_t)(nchunks + 1) << kPartBits);  // + digit totals row
    long long* keys_part = s.get<long long>((size_t)n);
    if (s.failed || !gb->bucket8 || !gb->part_off || !gb->slot_part || !gb->rows_part) return PDX_OOM;
    {
      // one pass over the keys: bucket byte per row (kept: it is the digit of the value partition of every later aggregation)
      // + the partition histogram
      PDX_PROFILE("hash_bucket_hist", st);
      hipLaunchKernelGGL(k_hash_bucket_hist, dim3((unsigned)ntiles), dim3(kSortBlock), 0, st, keys, valid, key->offset, n, gb->bucket8, gb->part_off);
    }
    int rcp = radix_scan_only<kPartBits>(gb->part_off, ntiles, chunk_sum, true, st);
    if (rcp == PDX_OK)  // keys and row ids in ONE scatter (they used to be two kernels ranking the same bucket bytes: 5.0 + 2.5 ms per 1e9 rows)
      rcp = radix_scatter_with_rows<kPartBits>(gb->bucket8, reinterpret_cast<const uint64_t*>(keys), reinterpret_cast<uint64_t*>(keys_part),
                                               gb->rows_part, n, gb->part_off, valid, key->offset, st);
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
        hipLaunchKernelGGL(k_hash_probe_lds, dim3(1u << pb), dim3(kProbeBlock), 0, st, keys_part, gb->rows_part, boff, n,
                           table, cap, region, gb->slot_part, ctl, pb, head_rows);
        if (!chunks.empty()) {
          TailChunk* dchunks = s.get<TailChunk>(chunks.size());
          if (s.failed) {
            pool_free(table);
            return PDX_OOM;
          }
          PDX_HIP(hipMemcpyAsync(dchunks, chunks.data(), chunks.size() * sizeof(TailChunk), hipMemcpyHostToDevice, st));
          hipLaunchKernelGGL(k_hash_probe_lds_tail, dim3((unsigned)chunks.size()), dim3(kProbeBlock), 0, st, keys_part, gb->rows_part, dchunks, table, cap,
                             region, gb->slot_part, ctl, pb);
          PDX_HIP(hipStreamSynchronize(st));  // `chunks` (pageable host memory) must outlive the copy
        }
      } else {
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
  uint32_t* k0 = s.get<uint32_t>((size_t)G);
  uint32_t* v0 = s.get<uint32_t>((size_t)G);
  uint32_t* k1 = s.get<uint32_t>((size_t)G);
  uint32_t* v1 = s.get<uint32_t>((size_t)G);
  if (s.failed || !gb->occ_slot || !gb->gid_of_occ || !gb->uniques || !gb->unique_ok || !gb->first_rows) return PDX_OOM;
  hipMemcpyAsync(gb->occ_slot, occ_slot_tmp, (size_t)G * sizeof(uint32_t), hipMemcpyDeviceToDevice, st);
  // order groups by first occurrence: sort (first_row -> slot); first rows are distinct so any order of ties is moot
  const uint32_t *ks = nullptr, *vs = nullptr;
  rc = radix_sort_pairs<uint32_t>(occ_first_tmp, occ_slot_tmp, k0, v0, k1, v1, G, ilog2((uint64_t)n + 1) < 31 ? ilog2((uint64_t)n + 1) : 31,
                                  &ks, &vs, true, s, st);
  if (rc != PDX_OK) return rc;
  int g = grid_for(G, 256);
  hipLaunchKernelGGL(k_assign_gids, dim3(g), dim3(256), 0, st, table, dense_min, dense_mask, gb->gid_of_slot, ks, vs, G, null_slot, gb->uniques,
                     gb->unique_ok, gb->first_rows, region);
  hipLaunchKernelGGL(k_gid_of_occ, dim3(g), dim3(256), 0, st, gb->gid_of_slot, gb->occ_slot, G, gb->gid_of_occ);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return hip_fail(e, "pdx_groupby_create");
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
  gb->stream = st;  // frees of the handle's blocks are ordered behind this stream
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
  gb->stream = st;  // frees of the handle's blocks are ordered behind this stream
  if (gb->G) PDX_HIP(hipMemcpyAsync(out_rows, gb->first_rows, (size_t)gb->G * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}

int pdx_groupby_group_ids(pdx_groupby* gb, uint32_t* out_ids, void* stream) {
  if (!gb || !out_ids) return fail(PDX_INVALID, "pdx_groupby_group_ids: null argument");
  hipStream_t st = as_stream(stream);
  gb->stream = st;  // frees of the handle's blocks are ordered behind this stream
  if (gb->n == 0) return PDX_OK;
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
  gb->stream = st;  // frees of the handle's blocks are ordered behind this stream
  if (gb->n == 0) return PDX_OK;
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
  gb->stream = st;  // frees of the handle's blocks are ordered behind this stream
  const int64_t n = gb->n, G = gb->G;
  const uint8_t* vvalid = validity_or_null(values);
  SegOut o{};
  bool want_pw = false, want_mm = false, want_is = false, want_std5 = false;
  // the "next" kinds (variance, stddev, product, first, last) run after the five standard ones on the same grouped values
  double *var_out = nullptr, *std_out = nullptr;
  void *prod_out = nullptr, *first_out = nullptr, *last_out = nullptr;
  for (int k = 0; k < nk; ++k) {
    pdx_mut_column* oc = &outs[k];
    if (oc->length < G) return fail(PDX_INVALID, "pdx_groupby_agg: output too small");
    if (G && !oc->values) return fail(PDX_INVALID, "pdx_groupby_agg: null output buffer");
    int want_dt;
    bool needs_validity = vvalid != nullptr;
    switch (kinds[k]) {
      case PDX_AGG_SUM:
        want_dt = is_f ? PDX_FLOAT64 : PDX_INT64;
        if (is_f) { o.sum_f = static_cast<double*>(oc->values); want_pw = true; }
        else { o.sum_i = static_cast<long long*>(oc->values); want_is = true; }
        want_std5 = true;
        break;
      case PDX_AGG_MEAN: want_dt = PDX_FLOAT64; o.mean = static_cast<double*>(oc->values); want_pw = true; want_std5 = true; break;
      case PDX_AGG_MIN: want_dt = values->dtype; o.vmin = oc->values; want_mm = true; want_std5 = true; break;
      case PDX_AGG_MAX: want_dt = values->dtype; o.vmax = oc->values; want_mm = true; want_std5 = true; break;
      case PDX_AGG_COUNT: want_dt = PDX_INT64; o.count = static_cast<long long*>(oc->values); needs_validity = false; want_std5 = true; break;
      case PDX_AGG_VARIANCE: want_dt = PDX_FLOAT64; var_out = static_cast<double*>(oc->values); break;
      case PDX_AGG_STDDEV: want_dt = PDX_FLOAT64; std_out = static_cast<double*>(oc->values); break;
      case PDX_AGG_PRODUCT: want_dt = values->dtype; prod_out = oc->values; break;
      case PDX_AGG_FIRST: want_dt = values->dtype; first_out = oc->values; break;
      case PDX_AGG_LAST: want_dt = values->dtype; last_out = oc->values; break;
      default: return fail(PDX_INVALID, "pdx_groupby_agg: unknown aggregate kind");
    }
    if (oc->dtype != want_dt) return fail(PDX_INVALID, "pdx_groupby_agg: output dtype does not match the aggregate's result type");
    if (needs_validity && !oc->validity)
      return fail(PDX_INVALID, "pdx_groupby_agg: values carry nulls but an output has no validity buffer");
    oc->length = G;
    oc->null_count = needs_validity ? -1 : 0;
  }
  if (G == 0) return PDX_OK;
  Scratch s;
  const void* vals_sorted = nullptr;
  const uint32_t* keys_sorted = nullptr;
  const uint32_t* seg_start = nullptr;
  const uint32_t* out_index = nullptr;
  const uint8_t* row_valid = nullptr;  // segments mode reads validity in place
  if (gb->mode == 0) {
    const uint64_t* vin = static_cast<const uint64_t*>(values->values) + values->offset;
    const uint64_t* vs = nullptr;
    // ---- fused last digit (the five standard kinds): sort by all but the top 6 slot bits, then rank + reduce in one kernel
    // (diagnostic switches are read on every call: tests flip them inside one process)
    const bool flr_env = [] { const char* e = getenv("PDX_FUSED_LAST_DIGIT"); return !(e && e[0] == '0'); }();
    const int part = gb->slot_part ? gb->part_bits : 0;
    // partitioned hash slots: the table itself is a power of two; only the two special slots (null key, INT64_MIN key) need one
    // more bit, so without them the top digit is drawn from the table's own bits (all 64 values used)
    const int eff_bits = (gb->slot_part && !gb->special_slots) ? gb->slot_bits - 1 : gb->slot_bits;
    const int low_bits = eff_bits - kFlrBits;
    const bool std_only = (want_std5 || var_out || std_out) && !prod_out && !first_out && !last_out;  // (variance: two more fused passes)
    const int64_t flr_min_rows = [] { const char* e = getenv("PDX_FUSED_LAST_DIGIT_MIN_ROWS"); return e ? atoll(e) : (1ll << 22); }();
    // a run is one workgroup's sequential work: it has to span a few tiles to amortise its prologue (1e8 groups: 119-row runs)
    const int64_t flr_min_run = [] { const char* e = getenv("PDX_FUSED_LAST_DIGIT_MIN_RUN"); return e ? atoll(e) : 8192ll; }();
    const int flr_min_low = [] { const char* e = getenv("PDX_FUSED_LAST_DIGIT_MIN_LOW_BITS"); return e ? atoi(e) : 10; }();
    // (hash-partitioned slots: only without the special slots -- with them the top digit is half empty and the runs half as long,
    //  measured slower than the classic path)
    const bool flr_hash = [] { const char* e = getenv("PDX_FUSED_LAST_DIGIT_HASH"); return !(e && e[0] == '0'); }();
    bool flr = flr_env && std_only && n >= flr_min_rows && low_bits - part >= 4 && low_bits >= flr_min_low && low_bits <= 26 &&
               (!gb->slot_part || (flr_hash && !gb->special_slots)) && (n >> low_bits) >= flr_min_run;
    if (flr && !gb->slot_part && gb->pass0_off)  // the stored pass-0 offsets belong to the first digit of the FULL plan
      flr = make_sort_plan(gb->slot_bits, sort_max_bits()).bits[0] == make_sort_plan(low_bits, sort_max_bits()).bits[0];
    // Narrowing sort (dense slots, values without nulls, two passes below the fused digit): a digit that has been sorted on is
    // dropped from the key, so pass 0 writes 2-byte keys, pass 1 reads them and writes the top digit alone in a byte, which is all
    // the fused kernel reads: 12 B/row less traffic than carrying the 4-byte slot through.  Run starts then come from the scatter
    // offsets (k_level_starts) instead of a search in the sorted slots.
    const SortPlan low_plan = make_sort_plan(low_bits, sort_max_bits());
    const bool narrow_env = [] { const char* e = getenv("PDX_SORT_NARROW"); return !(e && e[0] == '0'); }();
    // (values with nulls: the null flag rides in the narrow key's top bit, read from the validity bitmap by pass 0 itself)
    const bool narrow = flr && narrow_env && !gb->slot_part && gb->pass0_off && low_plan.npasses == 2 &&
                        gb->slot_bits - low_plan.bits[0] <= (vvalid ? 15 : 16) && low_plan.bits[0] <= 8 && low_plan.bits[1] <= 8 &&
                        eff_bits == gb->slot_bits;
    const uint8_t* keys8 = nullptr;       // narrowing sort: the top digit of every partially sorted row
    bool sorted_done = false;             // narrowing sort, skewed keys: the classic path's inputs are already built
    uint32_t* ss_narrow = nullptr;
    if (flr) {
      const int64_t nruns = (int64_t)1 << low_bits;
      uint32_t* run_start = s.get<uint32_t>((size_t)nruns + 1);
      unsigned int* dmax = s.get<unsigned int>(1);
      uint8_t* okb = vvalid ? s.get<uint8_t>((size_t)G) : nullptr;
      PDX_SCRATCH_CHECK(s);
      uint64_t *nv0 = nullptr, *nv1 = nullptr;
      uint32_t *nhist = nullptr, *nchunk = nullptr;
      uint8_t* k8 = nullptr;
      if (narrow) {
        const int b0 = low_plan.bits[0], b1 = low_plan.bits[1];
        const int64_t ntiles = ceil_div(n, kSortTile), nchunks = ceil_div(ntiles, kColChunk);
        uint16_t* k16 = s.get<uint16_t>((size_t)n);
        k8 = s.get<uint8_t>((size_t)n);
        nv0 = s.get<uint64_t>((size_t)n);
        nv1 = s.get<uint64_t>((size_t)n);
        nhist = s.get<uint32_t>((size_t)ntiles << 8);
        nchunk = s.get<uint32_t>((size_t)(nchunks + 1) << 8);
        PDX_SCRATCH_CHECK(s);
        int rcn = PDX_OK;
#define NARROW_P0(B)                                                                                                                         \
  rcn = vvalid ? radix_scatter_narrow<B, uint64_t, uint32_t, uint16_t, true>(gb->slot_of_row, vin, k16, nv0, n, gb->pass0_off, st, vvalid, values->offset) \
               : radix_scatter_narrow<B, uint64_t, uint32_t, uint16_t>(gb->slot_of_row, vin, k16, nv0, n, gb->pass0_off, st)
        switch (b0) {
          case 4: NARROW_P0(4); break;
          case 5: NARROW_P0(5); break;
          case 6: NARROW_P0(6); break;
          case 7: NARROW_P0(7); break;
          default: NARROW_P0(8); break;
        }
#undef NARROW_P0
        PDX_TRY(rcn);
#define NARROW_P1(B)                                                                    \
  rcn = radix_offsets<B, uint16_t>(k16, n, 0, nhist, nchunk, true, st);                  \
  if (rcn == PDX_OK)                                                                    \
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
          PDX_PROFILE("run_starts", st);
          // (row 0 of the pass-0 offsets = where every first digit's rows begin in the input of pass 1)
          hipLaunchKernelGGL((k_level_starts<uint16_t>), dim3(1u << b0), dim3(256), 0, st, k16, n, gb->pass0_off, (int64_t)1 << b0, b0, b1, nhist, run_start);
        }
        keys8 = k8;
        vs = nv1;
      } else {
        PDX_TRY(sort_values_by_slot(gb, vin, vvalid, values->offset, [&](size_t bytes) { return (void*)s.get<uint8_t>(bytes); }, s, st, &keys_sorted, &vs,
                                    gb->slot_bits - low_bits));
      }
      unsigned int hmax = 0;
      {
        PDX_PROFILE("run_starts", st);
        PDX_HIP(hipMemsetAsync(dmax, 0, sizeof(unsigned int), st));
        if (!narrow) hipLaunchKernelGGL(k_run_starts, dim3(grid_for(nruns + 1, 256)), dim3(256), 0, st, keys_sorted, n, low_bits, nruns, run_start, dmax);
        hipLaunchKernelGGL(k_run_max_len, dim3(grid_for(nruns, 256)), dim3(256), 0, st, run_start, nruns, dmax);
        PDX_LAUNCH_CHECK();
        PDX_HIP(hipMemcpyAsync(&hmax, dmax, sizeof(hmax), hipMemcpyDeviceToHost, st));
        PDX_HIP(hipStreamSynchronize(st));
      }
      if (narrow && hmax > (1u << 19)) {
        // skewed keys: the fused kernel is skipped.  Finish the sort with the one pass that is left (on the byte digits) and take the
        // group offsets from its scatter offsets: one more level of k_level_starts gives the start of every slot's rows
        ss_narrow = s.get<uint32_t>((size_t)G + 1);
        uint32_t* slot_start = s.get<uint32_t>(((size_t)nruns << kFlrBits) + 1);
        PDX_SCRATCH_CHECK(s);
        PDX_TRY((radix_offsets<kFlrBits, uint8_t>(k8, n, 0, nhist, nchunk, true, st)));
        if (vvalid) {  // the classic nullable reducers read the null flag from bit 31 of a 4-byte key per grouped row: write flags only
          uint32_t* fkeys = s.get<uint32_t>((size_t)n);
          PDX_SCRATCH_CHECK(s);
          PDX_TRY((radix_scatter_narrow<kFlrBits, uint64_t, uint8_t, uint32_t, true>(k8, nv1, fkeys, nv0, n, nhist, st)));
          keys_sorted = fkeys;
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
        vs = nv0;
        sorted_done = true;
      }
      if (hmax <= (1u << 19)) {  // a run is walked by ONE workgroup: keep the longest one short (skewed keys take the classic path)
        auto launch_flr = [&](const SegOut& oo, bool pw, bool mm, bool is, uint8_t* okbytes, const double* sqmean) {
          PDX_PROFILE("fused_last_digit_reduce", st);
          const int grid = (int)std::min<int64_t>(nruns, (int64_t)kCUs * 24);
          const bool dense = pw && !mm && !is && !vvalid;
          // (values with nulls, sum / mean / count: the thread-per-leaf form is opt-in, PDX_FLR_NULL_PW=1 -- measured 11.3 ms against
          //  10.7 ms of the literal per-lane replay at 5 % nulls: its per-group walk over the leaf markers is as serial as the replay)
          const bool nullpw = pw && !mm && !is && vvalid && [] { const char* e = getenv("PDX_FLR_NULL_PW"); return e && e[0] == '1'; }();
#define FLR_LAUNCH(TT, DD)                                                                                                                       \
  if (keys8 && nullpw)                                                                                                                           \
    hipLaunchKernelGGL((k_flr_reduce<TT, false, uint8_t, true>), dim3(grid), dim3(kSortBlock), 0, st, keys8, reinterpret_cast<const TT*>(vs), run_start, \
                       nruns, low_bits, gb->gid_of_slot, oo, okbytes, (int)pw, (int)mm, (int)is, 1, sqmean);                                       \
  else if (nullpw)                                                                                                                               \
    hipLaunchKernelGGL((k_flr_reduce<TT, false, uint32_t, true>), dim3(grid), dim3(kSortBlock), 0, st, keys_sorted, reinterpret_cast<const TT*>(vs),   \
                       run_start, nruns, low_bits, gb->gid_of_slot, oo, okbytes, (int)pw, (int)mm, (int)is, 1, sqmean);                            \
  else if (keys8)                                                                                                                                \
    hipLaunchKernelGGL((k_flr_reduce<TT, DD, uint8_t>), dim3(grid), dim3(kSortBlock), 0, st, keys8, reinterpret_cast<const TT*>(vs), run_start, nruns, \
                       low_bits, gb->gid_of_slot, oo, okbytes, (int)pw, (int)mm, (int)is, vvalid ? 1 : 0, sqmean);                                 \
  else                                                                                                                                           \
    hipLaunchKernelGGL((k_flr_reduce<TT, DD>), dim3(grid), dim3(kSortBlock), 0, st, keys_sorted, reinterpret_cast<const TT*>(vs), run_start, nruns, \
                       low_bits, gb->gid_of_slot, oo, okbytes, (int)pw, (int)mm, (int)is, vvalid ? 1 : 0, sqmean)
          // one WAVE per run (flr_wave.hpp) for everything but the dense sum / mean / count: nullable values, min / max and int64 sums
          // were a per-lane replay by wave 0 alone in the workgroup-per-run kernel (5 % nulls: 10.7 -> 4.9 ms per 1e9 rows); the dense
          // fast path of k_flr_reduce (thread per leaf) is still ahead of the wave form (3.0 vs 3.3 ms).  PDX_FLR_WAVE=1 / 0 force either.
          const char* fw_env = getenv("PDX_FLR_WAVE");
          const bool wave_form = !nullpw && (fw_env ? fw_env[0] != '0' : !dense);
          if (wave_form) {
            const int wgrid = (int)std::min<int64_t>(nruns, (int64_t)kCUs * 12 * 4);
            const bool pw_only = pw && !mm && !is;
            // counter levels: a group cannot outgrow its run, and a leaf holds 16 rows unless nulls cut it short
            const uint64_t max_leaves = vvalid ? (uint64_t)hmax + 1 : (uint64_t)hmax / 16 + 2;
            const int fw_levels = std::max(2, ilog2(max_leaves + 1));  // 2^levels > leaves: level index <= levels - 1
            const size_t fw_lds = (size_t)fw_lds_bytes(vvalid != nullptr, fw_levels);
#define FW_LAUNCH(TT, KK, KPTR, NN, PP)                                                                                                  \
  hipLaunchKernelGGL((k_flr_wave<TT, KK, NN, PP>), dim3(wgrid), dim3(64), fw_lds, st, KPTR, reinterpret_cast<const TT*>(vs), n, run_start, nruns, low_bits, \
                     gb->gid_of_slot, oo, okbytes, (int)pw, (int)mm, (int)is, sqmean, fw_levels)
#define FW_DISPATCH(TT)                                                                        \
  if (keys8) {                                                                                 \
    if (vvalid) { if (pw_only) FW_LAUNCH(TT, uint8_t, keys8, true, true); else FW_LAUNCH(TT, uint8_t, keys8, true, false); }          \
    else { if (pw_only) FW_LAUNCH(TT, uint8_t, keys8, false, true); else FW_LAUNCH(TT, uint8_t, keys8, false, false); }              \
  } else {                                                                                     \
    if (vvalid) { if (pw_only) FW_LAUNCH(TT, uint32_t, keys_sorted, true, true); else FW_LAUNCH(TT, uint32_t, keys_sorted, true, false); } \
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
        };
        if (want_std5) launch_flr(o, want_pw, want_mm, want_is, okb, nullptr);
        PDX_LAUNCH_CHECK();
        uint8_t* ok2 = nullptr;
        if (var_out || std_out) {
          // Arrow's two passes on the same partially sorted rows: per-group mean, then the pairwise sum of (x - mean)^2
          double* mean_g = s.get<double>((size_t)G);
          double* m2 = s.get<double>((size_t)G);
          long long* cnt_g = s.get<long long>((size_t)G);
          uint8_t* ok1 = vvalid ? s.get<uint8_t>((size_t)G) : nullptr;
          ok2 = vvalid ? s.get<uint8_t>((size_t)G) : nullptr;
          PDX_SCRATCH_CHECK(s);
          SegOut o1{};
          o1.mean = mean_g;
          launch_flr(o1, true, false, false, ok1, nullptr);
          SegOut o2{};
          o2.sum_f = m2;
          o2.count = cnt_g;
          launch_flr(o2, true, false, false, ok2, mean_g);
          hipLaunchKernelGGL(k_var_finish, dim3(grid_for(G, 256)), dim3(256), 0, st, m2, cnt_g, G, var_out, std_out);
          PDX_LAUNCH_CHECK();
        }
        for (int k = 0; k < nk; ++k) {
          uint8_t* bits = static_cast<uint8_t*>(outs[k].validity);
          if (!bits) continue;
          const uint8_t* src = (kinds[k] == PDX_AGG_VARIANCE || kinds[k] == PDX_AGG_STDDEV) ? ok2 : okb;
          if (!src || kinds[k] == PDX_AGG_COUNT) PDX_HIP(hipMemsetAsync(bits, 0xFF, (size_t)((G + 7) / 8), st));
          else hipLaunchKernelGGL(k_pack_bytes, dim3(grid_for((G + 7) / 8, 256)), dim3(256), 0, st, src, G, bits);
        }
        PDX_LAUNCH_CHECK();
        PDX_HIP(hipStreamSynchronize(st));
        return PDX_OK;
      }
    }
    // stable sort of (slot, value) by slot: each group's values become contiguous in row order
    uint32_t* ss = sorted_done ? ss_narrow : s.get<uint32_t>((size_t)G + 1);
    PDX_SCRATCH_CHECK(s);
    if (sorted_done) {
      // (narrowing sort, skewed keys: values and group offsets were built above)
    } else if (flr) {
      // the fused kernel was skipped (a run longer than 2^19 rows: skewed keys): finish the sort with the one pass that is left
      uint32_t* k2 = s.get<uint32_t>((size_t)n);
      uint64_t* v2 = s.get<uint64_t>((size_t)n);
      PDX_SCRATCH_CHECK(s);
      const uint32_t* ks2 = nullptr;
      const uint64_t* vs2 = nullptr;
      PDX_TRY((radix_sort_pairs<uint64_t>(keys_sorted, vs, k2, v2, k2, v2, n, kFlrBits, &ks2, &vs2, true, s, st, low_bits)));
      keys_sorted = ks2;
      vs = vs2;
    } else if (G == 1 && gb->slot_of_row) {
      // one group: the rows are grouped as they stand (no sort); null flags, if any, still go into a key per row
      keys_sorted = gb->slot_of_row;
      if (vvalid) {
        uint32_t* fk1 = s.get<uint32_t>((size_t)n);
        PDX_SCRATCH_CHECK(s);
        hipLaunchKernelGGL(k_flag_keys, dim3(grid_for(n, 256, 4)), dim3(256), 0, st, gb->slot_of_row, vvalid, values->offset, n, fk1);
        keys_sorted = fk1;
      }
      vs = vin;
    } else {
      PDX_TRY(sort_values_by_slot(gb, vin, vvalid, values->offset, [&](size_t bytes) { return (void*)s.get<uint8_t>(bytes); }, s, st, &keys_sorted, &vs));
    }
    vals_sorted = vs;
    if (!sorted_done) {
      PDX_PROFILE("seg_starts", st);
      hipLaunchKernelGGL(k_seg_starts, dim3(grid_for(G + 1, 256)), dim3(256), 0, st, keys_sorted, n, gb->occ_slot, G, ss);
    }
    PDX_LAUNCH_CHECK();
    seg_start = ss;
    out_index = gb->gid_of_occ;
  } else {
    vals_sorted = static_cast<const uint64_t*>(values->values) + values->offset;
    seg_start = gb->seg_start;
    row_valid = vvalid;
  }
  const uint32_t* fk = (gb->mode == 0 && vvalid) ? keys_sorted : nullptr;  // null flags of the grouped layout
  // one segmented reduce of `vals` (float64 or int64 per f64) into `oo`; ok_bytes (nullable values only): 1 = the group has a valid value
  auto reduce = [&](const void* vals, bool f64, const SegOut& oo, bool pw, bool mm, bool is, const uint32_t* oidx, uint8_t* ok_bytes) -> int {
    PDX_PROFILE("seg_reduce", st);
    if (!vvalid) {
      if (f64) return launch_seg_reduce_dense<double>(static_cast<const double*>(vals), seg_start, G, oidx, oo, pw, mm, is, n, s, st);
      return launch_seg_reduce_dense<long long>(static_cast<const long long*>(vals), seg_start, G, oidx, oo, pw, mm, is, n, s, st);
    }
    int grid = (int)std::min<int64_t>(ceil_div(G, kSegWaves), (int64_t)kCUs * 8);
    if (f64)
      hipLaunchKernelGGL((k_seg_reduce_nullable<double>), dim3(grid), dim3(kSegWaves * 64), 0, st, static_cast<const double*>(vals), fk, row_valid,
                         values->offset, seg_start, G, oidx, oo, ok_bytes);
    else
      hipLaunchKernelGGL((k_seg_reduce_nullable<long long>), dim3(grid), dim3(kSegWaves * 64), 0, st, static_cast<const long long*>(vals), fk,
                         row_valid, values->offset, seg_start, G, oidx, oo, ok_bytes);
    PDX_LAUNCH_CHECK();
    return reduce_huge_nullable_groups(vals, f64 ? PDX_FLOAT64 : PDX_INT64, fk, row_valid, values->offset, seg_start, G, oidx, n, oo, ok_bytes, s, st);
  };
  auto pack_validity = [&](uint8_t* bits, const uint8_t* ok_bytes) {
    if (!bits) return;
    if (!ok_bytes) hipMemsetAsync(bits, 0xFF, (size_t)((G + 7) / 8), st);
    else hipLaunchKernelGGL(k_pack_bytes, dim3(grid_for((G + 7) / 8, 256)), dim3(256), 0, st, ok_bytes, G, bits);
  };
  uint8_t* ok = nullptr;
  if (vvalid) {
    ok = s.get<uint8_t>((size_t)G);
    PDX_SCRATCH_CHECK(s);
  }
  if (want_std5) {
    PDX_TRY(reduce(vals_sorted, is_f, o, want_pw, want_mm, want_is, out_index, ok));
    for (int k = 0; k < nk; ++k)
      if (kinds[k] <= PDX_AGG_COUNT) pack_validity(static_cast<uint8_t*>(outs[k].validity), kinds[k] == PDX_AGG_COUNT ? nullptr : ok);
    PDX_LAUNCH_CHECK();
  }
  if (var_out || std_out) {
    // Arrow's two passes: mean = pairwise sum / count, then the pairwise sum of (x - mean)^2 over the same valid runs
    double* mean_seg = s.get<double>((size_t)G);
    double* d = s.get<double>((size_t)n);
    double* m2 = s.get<double>((size_t)G);
    long long* cnt_g = s.get<long long>((size_t)G);
    uint8_t* ok2 = vvalid ? s.get<uint8_t>((size_t)G) : nullptr;
    uint8_t* ok_seg = vvalid ? s.get<uint8_t>((size_t)G) : nullptr;
    PDX_SCRATCH_CHECK(s);
    SegOut o1{};
    o1.mean = mean_seg;
    PDX_TRY(reduce(vals_sorted, is_f, o1, true, false, false, nullptr, ok_seg));  // segment order
    {
      PDX_PROFILE("seg_sqdev", st);
      const int grid = (int)std::min<int64_t>(ceil_div(G, 4), (int64_t)kCUs * 16);
      if (is_f) hipLaunchKernelGGL((k_seg_sqdev<double>), dim3(grid), dim3(256), 0, st, static_cast<const double*>(vals_sorted), seg_start, G, mean_seg, d);
      else hipLaunchKernelGGL((k_seg_sqdev<long long>), dim3(grid), dim3(256), 0, st, static_cast<const long long*>(vals_sorted), seg_start, G, mean_seg, d);
      PDX_LAUNCH_CHECK();
    }
    SegOut o2{};
    o2.sum_f = m2;
    o2.count = cnt_g;
    PDX_TRY(reduce(d, true, o2, true, false, false, out_index, ok2));
    hipLaunchKernelGGL(k_var_finish, dim3(grid_for(G, 256)), dim3(256), 0, st, m2, cnt_g, G, var_out, std_out);
    for (int k = 0; k < nk; ++k)
      if (kinds[k] == PDX_AGG_VARIANCE || kinds[k] == PDX_AGG_STDDEV) pack_validity(static_cast<uint8_t*>(outs[k].validity), ok2);
    PDX_LAUNCH_CHECK();
  }
  if (prod_out || first_out || last_out) {
    uint8_t* pok = (vvalid && prod_out) ? s.get<uint8_t>((size_t)G) : nullptr;
    uint8_t* fok = (vvalid && first_out) ? s.get<uint8_t>((size_t)G) : nullptr;
    uint8_t* lok = (vvalid && last_out) ? s.get<uint8_t>((size_t)G) : nullptr;
    PDX_SCRATCH_CHECK(s);
    {
      PDX_PROFILE("seg_product_first_last", st);
      const int pf_grid = (int)std::min<int64_t>(ceil_div(G, 4), (int64_t)kCUs * 16);
      if (is_f)
        hipLaunchKernelGGL((k_seg_product_first_last<double>), dim3(pf_grid), dim3(256), 0, st, static_cast<const double*>(vals_sorted), fk,
                           row_valid, values->offset, seg_start, G, out_index, static_cast<double*>(prod_out), pok, static_cast<double*>(first_out), fok,
                           static_cast<double*>(last_out), lok);
      else
        hipLaunchKernelGGL((k_seg_product_first_last<long long>), dim3(pf_grid), dim3(256), 0, st, static_cast<const long long*>(vals_sorted),
                           fk, row_valid, values->offset, seg_start, G, out_index, static_cast<long long*>(prod_out), pok,
                           static_cast<long long*>(first_out), fok, static_cast<long long*>(last_out), lok);
    }
    for (int k = 0; k < nk; ++k) {
      if (kinds[k] == PDX_AGG_PRODUCT) pack_validity(static_cast<uint8_t*>(outs[k].validity), pok);
      if (kinds[k] == PDX_AGG_FIRST) pack_validity(static_cast<uint8_t*>(outs[k].validity), fok);
      if (kinds[k] == PDX_AGG_LAST) pack_validity(static_cast<uint8_t*>(outs[k].validity), lok);
    }
    PDX_LAUNCH_CHECK();
  }
  PDX_HIP(hipStreamSynchronize(st));  // scratch is returned to the pool on exit
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
  // adjustDatesAnchored (src/resample.cpp:85-178), tz == ""
  auto floor_div = [](long long a, long long b) { long long q = a / b, r = a % b; return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q; };
  const long long day = 86400000000000LL;
  long long first = mn, last = mx, origin = 0;
  const bool is_shard = (origin_type & PDX_ORIGIN_SHARD) != 0;
  origin_type &= ~PDX_ORIGIN_SHARD;
  switch (origin_type) {
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
  if (mn < first) return fail(PDX_INVALID, "Values falls before first bin");
  if (mx > last_edge) return fail(PDX_INVALID, "Values falls after last bin");
  long long nbins = nedges - 1;
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

int pdx_resample_row_labels(pdx_groupby* gb, int64_t* out_labels, void* stream) {
  if (!gb || !out_labels) return fail(PDX_INVALID, "pdx_resample_row_labels: null argument");
  if (gb->mode != 1) return fail(PDX_INVALID, "pdx_resample_row_labels: handle was not created by pdx_resample_create");
  hipStream_t st = as_stream(stream);
  gb->stream = st;  // frees of the handle's blocks are ordered behind this stream
  if (gb->n) hipLaunchKernelGGL(k_row_labels, dim3(grid_for(gb->n, 256, 4)), dim3(256), 0, st, gb->bin, gb->label_base, gb->n, out_labels);
  PDX_LAUNCH_CHECK();
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}

}  // extern "C"

// =====================================================================================================================
// Exact multi-GPU fp64 sum: partial-tree exchange (see include/pdx/abi.h).  Thread-per-group kernels: every group's values
// are contiguous (grouped), a thread walks its group once.  Uncoalesced but short: c ~ rows/group/rank.
// =====================================================================================================================
namespace pdx {

__device__ __forceinline__ int64_t aligned_block_level(int64_t s, int64_t kl) {
  // largest j with s % 2^j == 0 and s + 2^j <= kl
  int tz = s == 0 ? 62 : __ffsll((unsigned long long)s) - 1;
  int64_t room = kl - s;
  int lg = 63 - __clzll((unsigned long long)room);
  return tz < lg ? tz : lg;
}
__device__ __forceinline__ int64_t partial_record_count(int64_t a, int64_t c) {
  if (c <= 0) return 0;
  int64_t b = a + c, kf = (a + 15) >> 4, kl = b >> 4;
  if (kf > kl) return c;  // the whole range lies inside one leaf
  int64_t cnt = (16 * kf - a) + (b - 16 * kl);
  for (int64_t s = kf; s < kl;) {
    s += (int64_t)1 << aligned_block_level(s, kl);
    ++cnt;
  }
  return cnt;
}

__global__ void k_grouped_counts(const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ gid_of_occ, int64_t G,
                                 int64_t* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < G; k += stride)
    out[gid_of_occ[k]] = (int64_t)seg_start[k + 1] - (int64_t)seg_start[k];
}
// emission order: record block j belongs to local group order[j] (order == nullptr: group j); occ_of_gid maps a local group id to
// its position in slot order (where its values live)
__global__ void k_occ_of_gid(const uint32_t* __restrict__ gid_of_occ, int64_t G, uint32_t* __restrict__ occ_of_gid) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < G; k += stride) occ_of_gid[gid_of_occ[k]] = (uint32_t)k;
}
__global__ void k_partial_plan(const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ occ_of_gid, const int64_t* __restrict__ order,
                               int64_t G, const int64_t* __restrict__ prefix, int64_t* __restrict__ rec_cnt) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < G; j += stride) {
    const int64_t lg = order ? order[j] : j;
    const uint32_t k = occ_of_gid[lg];
    rec_cnt[j] = partial_record_count(prefix[lg], (int64_t)seg_start[k + 1] - (int64_t)seg_start[k]);
  }
}
__global__ void __launch_bounds__(256) k_partial_fill(const double* __restrict__ vals, const uint32_t* __restrict__ seg_start,
                                                      const uint32_t* __restrict__ occ_of_gid, const int64_t* __restrict__ order, int64_t G,
                                                      const int64_t* __restrict__ prefix, const int64_t* __restrict__ gid_map,
                                                      const int64_t* __restrict__ rec_off, int64_t* __restrict__ rec_key,
                                                      double* __restrict__ rec_val, int wave_form_too) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < G; j += stride) {
    const int64_t lg = order ? order[j] : j;
    const uint32_t k = occ_of_gid[lg];
    const double* v = vals + seg_start[k];
    const int64_t c = (int64_t)seg_start[k + 1] - (int64_t)seg_start[k];
    if (c <= 0) continue;
    const int64_t a = prefix[lg], b = a + c;
    const int64_t kf = (a + 15) >> 4, kl = b >> 4;
    if (wave_form_too && (kf > kl || kl - kf <= 64)) continue;  // k_partial_fill_wave emits this group
    const int64_t gkey = gid_map[lg] * 64;
    int64_t pos = rec_off[j];
    if (kf > kl) {
      for (int64_t i = 0; i < c; ++i) { rec_key[pos] = gkey; rec_val[pos] = v[i]; ++pos; }
      continue;
    }
    const int64_t h = 16 * kf - a;
    for (int64_t i = 0; i < h; ++i) { rec_key[pos] = gkey; rec_val[pos] = v[i]; ++pos; }
    for (int64_t sidx = kf; sidx < kl;) {
      const int lvl = (int)aligned_block_level(sidx, kl);
      const int64_t nleaf = (int64_t)1 << lvl;
      // perfect tree over leaves [sidx, sidx + nleaf): replay the counter, all merges happen inside the block
      PairwiseCounter cn;
      cn.init();
      const double* lv = v + (16 * sidx - a);
      for (int64_t q = 0; q < nleaf; ++q) {
        double acc = 0.0;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc += lv[q * 16 + e];
        cn.push(acc, 0);
      }
      rec_key[pos] = gkey + lvl + 1;
      rec_val[pos] = cn.sum[lvl];
      ++pos;
      sidx += nleaf;
    }
    for (int64_t i = 16 * kl - a; i < c; ++i) { rec_key[pos] = gkey; rec_val[pos] = v[i]; ++pos; }
  }
}

// Wave-per-group form of k_partial_fill for groups with at most 64 interior leaves (<= ~1050 rows; longer ones keep the thread form): lane l
// sums interior leaf l (16 contiguous values), six shuffle steps build every aligned perfect subtree at once (t[k] at lane r = the tree
// over leaves [r, r + 2^k), left + right as the counter merges them), the boundary-leaf fragments are copied by the lanes.  The thread
// form walks each group with one thread and a 64-entry counter in scratch memory (2.2 ms per 5e8 rows).
__global__ void __launch_bounds__(256) k_partial_fill_wave(const double* __restrict__ vals, const uint32_t* __restrict__ seg_start,
                                                           const uint32_t* __restrict__ occ_of_gid, const int64_t* __restrict__ order, int64_t G,
                                                           const int64_t* __restrict__ prefix, const int64_t* __restrict__ gid_map,
                                                           const int64_t* __restrict__ rec_off, int64_t* __restrict__ rec_key,
                                                           double* __restrict__ rec_val) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); j < G; j += nw) {
    const int64_t lg = order ? order[j] : j;
    const uint32_t k = occ_of_gid[lg];
    const double* v = vals + seg_start[k];
    const int64_t c = (int64_t)seg_start[k + 1] - (int64_t)seg_start[k];
    if (c <= 0) continue;
    const int64_t a = prefix[lg], b = a + c;
    const int64_t kf = (a + 15) >> 4, kl = b >> 4;
    const int64_t gkey = gid_map[lg] * 64;
    const int64_t pos0 = rec_off[j];
    if (kf > kl) {  // the whole range lies inside one leaf (< 31 rows): fragments only
      for (int64_t i = lane; i < c; i += 64) {
        rec_key[pos0 + i] = gkey;
        rec_val[pos0 + i] = v[i];
      }
      continue;
    }
    const int nint = (int)(kl - kf);
    if (kl - kf > 64) continue;  // long group: k_partial_fill
    const int h = (int)(16 * kf - a);
    if (lane < h) {
      rec_key[pos0 + lane] = gkey;
      rec_val[pos0 + lane] = v[lane];
    }
    double t[7];
    t[0] = 0.0;
    if (lane < nint) {
      const double* lv = v + h + 16 * lane;
      double acc = 0.0;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc += lv[e];
      t[0] = acc;
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) t[q + 1] = t[q] + __shfl_down(t[q], 1 << q, 64);
    int64_t pos = pos0 + h;
    for (int64_t sidx = kf; sidx < kl;) {
      const int lvl = (int)aligned_block_level(sidx, kl);
      if (lane == (int)(sidx - kf)) {
        double val = t[0];
#pragma unroll
        for (int q = 1; q < 7; ++q) val = lvl == q ? t[q] : val;
        rec_key[pos] = gkey + lvl + 1;
        rec_val[pos] = val;
      }
      ++pos;
      sidx += (int64_t)1 << lvl;
    }
    const int ntail = (int)(b - 16 * kl);
    if (lane < ntail) {
      rec_key[pos + lane] = gkey;
      rec_val[pos + lane] = v[16 * kl - a + lane];
    }
  }
}

__global__ void k_replay_keys(const int64_t* __restrict__ rec_key, int64_t m, int64_t gid_lo, int64_t n_own, uint32_t* __restrict__ slot,
                              uint32_t* __restrict__ lvl, unsigned int* __restrict__ bad, int pack_shift) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
    int64_t key = rec_key[i];
    int64_t g = (key >> 6) - gid_lo;
    if (g < 0 || g >= n_own) {
      atomicExch(bad, 1u);
      g = 0;
    }
    // (pack_shift >= 0: the level rides above the slot bits of the sort key -- the sort only looks at the low bits)
    slot[i] = pack_shift >= 0 ? (uint32_t)g | ((uint32_t)(key & 63) << pack_shift) : (uint32_t)g;
    lvl[i] = (uint32_t)(key & 63);
  }
}
// lvl_shift >= 0: `lvl` holds the sorted keys with the record's level packed above bit lvl_shift (one sort carries it along)
__global__ void __launch_bounds__(256) k_replay(const uint32_t* __restrict__ seg_start, int64_t n_own, const double* __restrict__ val,
                                                const uint32_t* __restrict__ lvl, double* __restrict__ out, unsigned int* __restrict__ bad,
                                                int lvl_shift) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_own; g += stride) {
    PairwiseCounter cn;
    cn.init();
    double acc = 0.0;
    int fill = 0;
    bool any = false;
    for (int64_t i = seg_start[g]; i < (int64_t)seg_start[g + 1]; ++i) {
      uint32_t l = lvl_shift >= 0 ? (lvl[i] & kSortKeyMask) >> lvl_shift : lvl[i];
      any = true;
      if (l == 0) {  // fragment value: extend the running 16-value leaf
        acc += val[i];
        if (++fill == 16) {
          cn.push(acc, 0);
          acc = 0.0;
          fill = 0;
        }
      } else {
        if (fill != 0) atomicExch(bad, 2u);  // a node must start on a leaf boundary
        cn.push(val[i], (int)l - 1);
      }
    }
    if (fill) cn.push(acc, 0);
    if (!any) atomicExch(bad, 3u);
    out[g] = any ? cn.finish() : 0.0;
  }
}
__global__ void k_iota_u32(uint32_t* p, int64_t n) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = (uint32_t)i;
}

}  // namespace pdx

struct pdx_grouped {
  pdx_groupby* gb = nullptr;
  int64_t n = 0, G = 0;
  const double* vals_sorted = nullptr;
  uint32_t* seg_start = nullptr;  // G + 1, slot (occ) order
  uint32_t* occ_of_gid = nullptr; // G: local group id -> position in slot order
  int64_t* rec_off = nullptr;     // G + 1 after plan (emission order)
  const int64_t* prefix = nullptr;
  const int64_t* order = nullptr;
  int64_t total = -1;
  mutable hipStream_t stream = nullptr;
  std::vector<void*> owned;
  template <typename T>
  T* own(size_t count) {
    T* p = static_cast<T*>(pool_alloc((count ? count : 1) * sizeof(T)));
    if (p) owned.push_back(p);
    return p;
  }
  ~pdx_grouped() {
    StreamNote note(stream);
    pool_free_many(owned.data(), (int)owned.size());
  }
};

extern "C" {

int pdx_groupby_group_values(pdx_groupby* gb, const pdx_column* values, void* stream, pdx_grouped** out) {
  if (!gb || !out) return fail(PDX_INVALID, "pdx_groupby_group_values: null argument");
  PDX_TRY(check_column(values, "pdx_groupby_group_values"));
  if (gb->mode != 0) return fail(PDX_INVALID, "pdx_groupby_group_values: needs a hash group-by handle");
  if (values->dtype != PDX_FLOAT64 || validity_or_null(values)) return fail(PDX_NOT_IMPLEMENTED, "pdx_groupby_group_values: float64 values without nulls only");
  if (values->length != gb->n) return fail(PDX_INVALID, "pdx_groupby_group_values: values length differs from the grouped key length");
  hipStream_t st = as_stream(stream);
  gb->stream = st;  // frees of the handle's blocks are ordered behind this stream
  std::unique_ptr<pdx_grouped> gowner(new pdx_grouped());
  gowner->stream = st;
  pdx_grouped* g = gowner.get();
  g->gb = gb;
  g->n = gb->n;
  g->G = gb->G;
  *out = nullptr;
  const int64_t n = gb->n, G = gb->G;
  g->seg_start = g->own<uint32_t>((size_t)G + 1);
  g->occ_of_gid = g->own<uint32_t>((size_t)G);
  if (!g->seg_start || !g->occ_of_gid) return PDX_OOM;
  if (n > 0) {
    Scratch s;
    const uint32_t* ks = nullptr;
    const uint64_t* vs = nullptr;
    bool narrow_done = false;
    int rc = sort_values_narrow_full(gb, static_cast<const uint64_t*>(values->values) + values->offset,
                                     [&](size_t bytes) { return (void*)g->own<uint8_t>(bytes); }, s, st, &vs, g->seg_start, &narrow_done);
    if (rc != PDX_OK) return rc;
    if (!narrow_done) {
      rc = sort_values_by_slot(gb, static_cast<const uint64_t*>(values->values) + values->offset, nullptr, 0,
                               [&](size_t bytes) { return (void*)g->own<uint8_t>(bytes); }, s, st, &ks, &vs);
      if (rc != PDX_OK) return rc;
      hipLaunchKernelGGL(k_seg_starts, dim3(grid_for(G + 1, 256)), dim3(256), 0, st, ks, n, gb->occ_slot, G, g->seg_start);
    }
    g->vals_sorted = reinterpret_cast<const double*>(vs);
    hipLaunchKernelGGL(k_occ_of_gid, dim3(grid_for(G, 256)), dim3(256), 0, st, gb->gid_of_occ, G, g->occ_of_gid);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return hip_fail(e, "pdx_groupby_group_values");
  }
  *out = gowner.release();
  return PDX_OK;
}
int pdx_grouped_destroy(pdx_grouped* g) {
  delete g;
  return PDX_OK;
}
int pdx_grouped_counts(pdx_grouped* g, int64_t* out_counts, void* stream) {
  if (!g || !out_counts) return fail(PDX_INVALID, "pdx_grouped_counts: null argument");
  hipStream_t st = as_stream(stream);
  g->stream = st;  // frees of the handle's blocks are ordered behind this stream
  if (g->G) hipLaunchKernelGGL(k_grouped_counts, dim3(grid_for(g->G, 256)), dim3(256), 0, st, g->seg_start, g->gb->gid_of_occ, g->G, out_counts);
  PDX_LAUNCH_CHECK();
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}
int pdx_grouped_partial_plan(pdx_grouped* g, const int64_t* prefix, const int64_t* order, int64_t* out_total, void* stream) {
  if (!g || !prefix || !out_total) return fail(PDX_INVALID, "pdx_grouped_partial_plan: null argument");
  hipStream_t st = as_stream(stream);
  g->stream = st;  // frees of the handle's blocks are ordered behind this stream
  *out_total = 0;
  g->prefix = prefix;
  g->order = order;
  g->total = 0;
  if (g->G == 0) return PDX_OK;
  if (!g->rec_off) g->rec_off = g->own<int64_t>((size_t)g->G + 1);
  if (!g->rec_off) return PDX_OOM;
  Scratch s;
  int64_t* total = s.get<int64_t>(1);
  PDX_SCRATCH_CHECK(s);
  hipLaunchKernelGGL(k_partial_plan, dim3(grid_for(g->G, 256)), dim3(256), 0, st, g->seg_start, g->occ_of_gid, order, g->G, prefix, g->rec_off);
  PDX_TRY((device_exclusive_scan<int64_t, SumOp>(g->rec_off, g->rec_off, g->G, total, s, st)));
  PDX_HIP(hipMemcpyAsync(&g->total, total, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
  *out_total = g->total;
  return PDX_OK;
}
int pdx_grouped_partial_fill(pdx_grouped* g, const int64_t* gid_map, int64_t* rec_key, double* rec_val, void* stream) {
  if (!g || !gid_map || !rec_key || !rec_val) return fail(PDX_INVALID, "pdx_grouped_partial_fill: null argument");
  if (g->total < 0 || !g->prefix) return fail(PDX_INVALID, "pdx_grouped_partial_fill: call pdx_grouped_partial_plan first");
  hipStream_t st = as_stream(stream);
  g->stream = st;  // frees of the handle's blocks are ordered behind this stream
  if (g->G) {
    PDX_PROFILE("partial_fill", st);
    {
      const int wave_form = [] { const char* e = getenv("PDX_PARTIAL_FILL_WAVE"); return !(e && e[0] == '0'); }() ? 1 : 0;
      if (wave_form)
        hipLaunchKernelGGL(k_partial_fill_wave, dim3((unsigned)std::min<int64_t>(ceil_div(g->G, 4), (int64_t)kCUs * 32)), dim3(256), 0, st, g->vals_sorted,
                           g->seg_start, g->occ_of_gid, g->order, g->G, g->prefix, gid_map, g->rec_off, rec_key, rec_val);
  hipLaunchKernelGGL(k_partial_fill, dim3(grid_for(g->G, 256)), dim3(256), 0, st, g->vals_sorted, g->seg_start, g->occ_of_gid, g->order, g->G, g->prefix,
                       gid_map, g->rec_off, rec_key, rec_val, wave_form);
    }
  }
  PDX_LAUNCH_CHECK();
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}
int pdx_replay_partials(const int64_t* rec_key, const double* rec_val, int64_t m, int64_t gid_lo, int64_t n_own, double* out_sum, void* stream) {
  if (m < 0 || n_own < 0 || (m && (!rec_key || !rec_val)) || (n_own && !out_sum)) return fail(PDX_INVALID, "pdx_replay_partials: bad argument");
  if (m > 0x7FFFFFFFll || n_own > 0x3FFFFFFFll) return fail(PDX_NOT_IMPLEMENTED, "pdx_replay_partials: too many records / groups for one call");
  hipStream_t st = as_stream(stream);
  if (n_own == 0) return PDX_OK;
  Scratch s;
  uint32_t* slot = s.get<uint32_t>((size_t)m);
  uint32_t* lvl = s.get<uint32_t>((size_t)m);
  uint32_t *k0 = s.get<uint32_t>((size_t)m), *k1 = s.get<uint32_t>((size_t)m), *k2 = s.get<uint32_t>((size_t)m), *k3 = s.get<uint32_t>((size_t)m);
  uint64_t *v0 = s.get<uint64_t>((size_t)m), *v1 = s.get<uint64_t>((size_t)m);
  uint32_t *l0 = s.get<uint32_t>((size_t)m), *l1 = s.get<uint32_t>((size_t)m);
  uint32_t* ids = s.get<uint32_t>((size_t)n_own);
  uint32_t* ss = s.get<uint32_t>((size_t)n_own + 1);
  unsigned int* bad = s.get<unsigned int>(1);
  PDX_SCRATCH_CHECK(s);
  PDX_HIP(hipMemsetAsync(bad, 0, sizeof(unsigned int), st));
  int bits = ilog2((uint64_t)n_own + 1);
  if (bits < 1) bits = 1;
  // the record's level (6 bits) rides above the slot bits of the 31-bit sort key when there is room: ONE sort instead of two
  const int pack_shift = bits <= 25 ? 25 : -1;
  if (m) hipLaunchKernelGGL(k_replay_keys, dim3(grid_for(m, 256, 4)), dim3(256), 0, st, rec_key, m, gid_lo, n_own, slot, lvl, bad, pack_shift);
  const uint32_t *ks = slot, *ks2 = nullptr, *ls = lvl;
  const uint64_t* vs = reinterpret_cast<const uint64_t*>(rec_val);
  if (m) {
    ProfileTagOverride replay_tag("replay_sort");  // (not the per-row scatter passes the bench prices against the roofline)
    PDX_TRY(radix_sort_pairs<uint64_t>(slot, reinterpret_cast<const uint64_t*>(rec_val), k0, v0, k1, v1, m, bits, &ks, &vs, true, s, st));
    if (pack_shift < 0) PDX_TRY(radix_sort_pairs<uint32_t>(slot, lvl, k2, l0, k3, l1, m, bits, &ks2, &ls, true, s, st));
  }
  if (pack_shift >= 0) {
    hipLaunchKernelGGL(k_seg_starts_masked, dim3(grid_for(n_own + 1, 256)), dim3(256), 0, st, ks, m, (1u << pack_shift) - 1u, n_own, ss);
  } else {
    hipLaunchKernelGGL(k_iota_u32, dim3(grid_for(n_own, 256)), dim3(256), 0, st, ids, n_own);
    hipLaunchKernelGGL(k_seg_starts, dim3(grid_for(n_own + 1, 256)), dim3(256), 0, st, ks, m, ids, n_own, ss);
  }
  {
    PDX_PROFILE("replay_partials", st);
    hipLaunchKernelGGL(k_replay, dim3(grid_for(n_own, 256)), dim3(256), 0, st, ss, n_own, reinterpret_cast<const double*>(vs), pack_shift >= 0 ? ks : ls, out_sum,
                       bad, pack_shift);
  }
  PDX_LAUNCH_CHECK();
  unsigned int hbad = 0;
  PDX_HIP(hipMemcpyAsync(&hbad, bad, sizeof(hbad), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
  if (hbad == 1) return fail(PDX_INVALID, "pdx_replay_partials: record for a group outside [gid_lo, gid_lo + n_own)");
  if (hbad == 2) return fail(PDX_INVALID, "pdx_replay_partials: node record inside an unfinished leaf (records out of order)");
  if (hbad == 3) return fail(PDX_INVALID, "pdx_replay_partials: an owned group received no record");
  return PDX_OK;
}

}  // extern "C"
