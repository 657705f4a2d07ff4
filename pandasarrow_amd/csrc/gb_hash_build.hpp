// gb_hash_build.hpp -- part of groupby.hip (textually included there, in its namespace context; split by stage, kernels unchanged):
// general keys -> slots: the global table (tiny inputs), the hash partition and the LDS-resident bucket tables.
#pragma once


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
// (three 32-bit multiplies: the high word folded in with one, then the two rounds of the "lowbias32" integer mixer -- the probe
//  kernel is instruction-issue bound, and splitmix64's two 64-bit multiplies were a sixth of its vector instructions)
__device__ __forceinline__ uint32_t key_hash32(long long k, bool is_null) {
  uint32_t h = (uint32_t)(unsigned long long)k ^ ((uint32_t)((unsigned long long)k >> 32) * 0x85EBCA6Bu);
  h ^= h >> 16;
  h *= 0x7FEB352Du;
  h ^= h >> 15;
  h *= 0x846CA68Bu;
  h ^= h >> 16;
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
#ifndef PDX_HASH_PAIR
#define PDX_HASH_PAIR 1  // (compile-time diagnostic: 0 = one slot per LDS read in k_hash_probe_lds, probe sequences start on any slot)
#endif
constexpr unsigned int kProbeStartMask = PDX_HASH_PAIR ? ~1u : ~0u;
#ifndef PDX_HASH_U
#define PDX_HASH_U 4  // rows per thread and trip of k_hash_probe_lds (the next trip's rows are in flight during this one's probes)
#endif
constexpr int kProbeBlock = 1024;
// ROWS = true: the keys may be null (bit 31 of rows_part) -- every row's original index is read with its key and the first row of a slot is
// tracked as that index.  ROWS = false (no null keys): rows_part is NOT read per row; a slot's first row is tracked as the smallest
// POSITION inside the bucket (the partition is stable, so positions order a bucket's rows as their row numbers do) and turned into a row
// number once per slot when the region is written back: 4 B/row less to read.  With idx16 the 4-byte logical slot is not written either
// (slot = idx16 << pb | bucket; ensure_slot_part rebuilds it for the few callers that want it): 8 + 2 instead of 12 + 6 bytes per row.
template <bool ROWS>
__global__ void __launch_bounds__(kProbeBlock) k_hash_probe_lds(const long long* __restrict__ keys_part, const uint32_t* __restrict__ rows_part,
                                                                const uint32_t* __restrict__ bucket_off,
                                                                int64_t n, Slot* table, unsigned int cap, unsigned int region,
                                                                uint32_t* __restrict__ slot_part, HashCtl* ctl, unsigned int pb, int64_t head_rows,
                                                                uint16_t* __restrict__ idx16 /* slot index inside the bucket's region, or null */) {
  __shared__ __attribute__((aligned(16))) unsigned long long lkeys[kLdsRegionMax];
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
  constexpr int U = PDX_HASH_U;
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
      nrow[u] = ROWS ? (p < end ? rows_part[p] : 0u) : (uint32_t)(p - start);
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
        // The probe sequence starts on an EVEN slot and one 16-byte LDS read looks at two slots: the divergent chain loop of the wave
        // (it runs to the longest chain among its 64 x 4 rows) makes about half as many trips.  Still plain linear probing over slots
        // -- a key sits in the first slot of its sequence that was empty when it arrived -- so the chunked tail kernel, which walks
        // the same sequence slot by slot (same even start), finds every key where this kernel put it.
        unsigned int idx = (h[u] >> pb) & rmask & kProbeStartMask, probes = 0;
        for (;;) {
#if PDX_HASH_PAIR
          const ulonglong2 pr = *reinterpret_cast<const ulonglong2*>(&lkeys[idx]);
          unsigned long long c0 = pr.x, c1 = pr.y;
          if (c0 == (unsigned long long)key[u]) break;
          if (c1 == (unsigned long long)key[u]) { idx += 1; break; }
          if (c0 == (unsigned long long)kEmptyKey) {
            unsigned long long old = atomicCAS(&lkeys[idx], (unsigned long long)kEmptyKey, (unsigned long long)key[u]);
            if (old == (unsigned long long)kEmptyKey) {
              atomicAdd(&linserted, 1u);
              break;
            }
            if (old == (unsigned long long)key[u]) break;
            c1 = lkeys[idx + 1];  // the slot went to another key in the meantime: the second one may have changed as well
            if (c1 == (unsigned long long)key[u]) { idx += 1; break; }
          }
          if (c1 == (unsigned long long)kEmptyKey) {
            unsigned long long old = atomicCAS(&lkeys[idx + 1], (unsigned long long)kEmptyKey, (unsigned long long)key[u]);
            if (old == (unsigned long long)kEmptyKey) {
              atomicAdd(&linserted, 1u);
              idx += 1;
              break;
            }
            if (old == (unsigned long long)key[u]) { idx += 1; break; }
          }
          idx = (idx + 2) & rmask;
          probes += 2;
          if (probes > 512) {  // pathologically long probe chain: the host retries with a larger table (L2 path)
            atomicExch(&ctl->overflow, 1u);
            break;
          }
#else
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
#endif
        }
        if (r < lfirst[idx]) atomicMin(&lfirst[idx], r);
        logical = (idx << pb) | b;
      }
      if (idx16) idx16[p0 + (int64_t)u * kProbeBlock] = (uint16_t)((logical >> pb) & 0xFFFFu);
      else slot_part[p0 + (int64_t)u * kProbeBlock] = logical;
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
    // position inside the bucket -> row number (rows_part == null: the partition wrote no row ids; the positions are left in the table
    // and k_first_rows_from_pos turns them into rows from the partition's own offsets)
    if (!ROWS && rows_part && sl.first != kNoRow) sl.first = rows_part[start + sl.first] & 0x7FFFFFFFu;
    sl.gid = kNoRow;
    table[(int64_t)b * region + i] = sl;
  }
  if (tid == 0 && linserted) atomicAdd(&ctl->inserted, linserted);
  if (tid < 2 && lspecial[tid] != kNoRow) {
    unsigned int sp = lspecial[tid];
    if (!ROWS && rows_part) sp = rows_part[start + sp] & 0x7FFFFFFFu;  // (the INT64_MIN key; null keys take the ROWS form)
    atomicMin(&table[cap + tid].first, sp);
  }
}

// The partition without row ids (no null keys): a slot's first row is known as a POSITION p inside its bucket b.  The stable partition
// put the bucket's rows of tile t at [offsets[t][b], offsets[t + 1][b]) in row order, so the row is found from the partition's own
// tables: the tile by binary search down column b of the offsets, the row inside the tile as the (k + 1)-th row whose bucket byte is b.
// One wave per used slot: <= 18 dependent 4-byte reads + one 4 KB tile of bucket bytes (first rows cluster in the early tiles: L2 hits).
// Replaces 4 B/row of row ids written by the partition pass and the staging they took in its LDS (5.9 -> 3.9 ms per 1e9 rows).
__global__ void __launch_bounds__(256) k_first_rows_from_pos(Slot* __restrict__ table, unsigned int cap, unsigned int region,
                                                             const uint32_t* __restrict__ offsets /* [tiles][1 << kPartBits] */, int64_t ntiles,
                                                             const uint8_t* __restrict__ bucket8, int64_t n) {
  constexpr int R = 1 << kPartBits;
  static_assert(kSortTile == 4096, "one wave reads a tile as 64 lanes x 64 bucket bytes");
  const int lane = threadIdx.x & 63;
  // 64 slots per wave: every lane searches the tile of ITS slot (the dependent reads of 64 searches overlap), then the wave scans the tiles
  // of the used slots one after the other, the next tile's bytes requested before the current one is counted
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned int pos = kNoRow;
  if (i < (int64_t)cap + 2 && i != (int64_t)cap) pos = table[i].first;  // (the null-key slot is never used in this form)
  const bool used = pos != kNoRow;
  const unsigned int b = i < (int64_t)cap ? (unsigned int)(i / region) : 1u;  // the INT64_MIN key hashes to 1
  uint32_t tile = 0, k = 0;
  if (used) {
    const uint32_t P = offsets[b] + pos;
    int64_t lo = 0, hi = ntiles - 1;  // the last tile whose rows of bucket b start at or before P
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (offsets[mid * R + b] <= P) lo = mid;
      else hi = mid - 1;
    }
    tile = (uint32_t)lo;
    k = P - offsets[lo * R + b];
  }
  unsigned long long todo = __ballot(used);
  if (!todo) return;
  auto load_tile = [&](uint32_t t, unsigned int bb, unsigned long long (&w)[8]) {
    const int64_t base = (int64_t)t * kSortTile + (int64_t)lane * 64;
    const unsigned long long pat = 0x0101010101010101ull * bb;
    if (base + 64 <= n) {
      const ulonglong2* src = reinterpret_cast<const ulonglong2*>(bucket8 + base);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const ulonglong2 v = src[q];
        w[2 * q] = v.x ^ pat;
        w[2 * q + 1] = v.y ^ pat;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        unsigned long long v = 0;
        for (int j = 0; j < 8; ++j) {
          const int64_t r = base + q * 8 + j;
          v |= (unsigned long long)(r < n ? (uint8_t)(bucket8[r] ^ bb) : (uint8_t)0xFF) << (8 * j);
        }
        w[q] = v;
      }
    }
  };
  unsigned long long w[8], wn[8];
  int j = __ffsll((long long)todo) - 1;
  todo &= todo - 1;
  load_tile(__shfl(tile, j, 64), __shfl(b, j, 64), w);
  unsigned int my_row = 0;
  for (;;) {
    const int jn = todo ? __ffsll((long long)todo) - 1 : -1;
    if (jn >= 0) {
      todo &= todo - 1;
      load_tile(__shfl(tile, jn, 64), __shfl(b, jn, 64), wn);
    }
    const uint32_t kj = __shfl(k, j, 64), tj = __shfl(tile, j, 64);
    // 0x80 in every byte of w[q] that is zero (a row of the slot's bucket), exact (no borrow between bytes)
    unsigned int cnt = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const unsigned long long x = w[q], lowbits = 0x7F7F7F7F7F7F7F7Full;
      w[q] = ~(((x & lowbits) + lowbits) | x | lowbits);
      cnt += (unsigned int)__popcll(w[q]);
    }
    unsigned int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned int up = __shfl_up(incl, d, 64);
      if (lane >= d) incl += up;
    }
    const unsigned int excl = incl - cnt;
    const bool mine = kj >= excl && kj < incl;  // exactly one lane
    unsigned int row = 0;
    if (mine) {
      unsigned int left = kj - excl;
      bool done = false;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const unsigned int c = (unsigned int)__popcll(w[q]);
        if (!done && left < c) {
          unsigned long long m = w[q];
          for (unsigned int d = 0; d < left; ++d) m &= m - 1;  // drop the `left` lowest matches
          row = tj * (uint32_t)kSortTile + (uint32_t)lane * 64u + (uint32_t)q * 8u + ((unsigned int)__ffsll((long long)m) - 1) / 8;
          done = true;
        } else if (!done) {
          left -= c;
        }
      }
    }
    const unsigned long long who = __ballot(mine);
    const unsigned int found = __shfl(row, who ? __ffsll((long long)who) - 1 : 0, 64);
    if (lane == j) my_row = found;
    if (jn < 0) break;
    j = jn;
#pragma unroll
    for (int q = 0; q < 8; ++q) w[q] = wn[q];
  }
  if (used) table[i].first = my_row;
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
                                                                     unsigned int region, uint32_t* __restrict__ slot_part, HashCtl* ctl, unsigned int pb,
                                                                     uint16_t* __restrict__ idx16) {
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
        unsigned int idx = (key_hash32(key[u], false) >> pb) & rmask & kProbeStartMask, probes = 0;  // (the head kernel's sequence: even start)
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
      if (idx16) idx16[p0 + (int64_t)u * kProbeBlock] = (uint16_t)((logical >> pb) & 0xFFFFu);
      else slot_part[p0 + (int64_t)u * kProbeBlock] = logical;
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
// the 4-byte logical slot of every partitioned position from its 2-byte region index: slot = idx16 << kPartBits | bucket, the bucket of a
// position found in the 257 bucket starts (LDS).  Only the callers that sort 4-byte slots or map rows back want it.
__global__ void __launch_bounds__(256) k_slot_part_from_idx16(const uint16_t* __restrict__ idx16, const uint32_t* __restrict__ bucket_off, int64_t n,
                                                              uint32_t* __restrict__ slot_part) {
  constexpr int R = 1 << kPartBits;
  __shared__ uint32_t st[R + 1];
  for (int d = threadIdx.x; d <= R; d += 256) st[d] = d < R ? bucket_off[d] : (uint32_t)n;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
    int lo = 0, hi = R;  // the last bucket that starts at or before p
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int64_t)st[mid] <= p) lo = mid;
      else hi = mid;
    }
    slot_part[p] = ((uint32_t)idx16[p] << kPartBits) | (uint32_t)lo;
  }
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
