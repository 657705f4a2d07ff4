// gb_acc.hpp -- part of groupby.hip (namespace pdx): the ORDER-FREE group aggregates -- count, min, max, int64 sum -- without the value sort.
//
// Reference: GROUPBY_AGG(min | max | sum) src/pd_core_macros.h:80-147, GROUPBY_NUMERIC_AGG(count) 5-78, instantiated
// src/dataframe.cpp:1526-1534.  Per group these are Arrow's `min` / `max` / `sum` / `count` over the group's rows; none of them depends on
// the ORDER of the rows (int64 sums wrap; min / max compare; the one order rule -- which of two tied zeros of different sign is kept --
// is settled by a rare second pass, below), so the stable value sort the fp64 sums need (gb_layout.hpp) is not needed here.
//
// What bounds the design (tools/ubench/atomics.hip, profiles/r04_ubench_atomics.log, 1e9 rows into 1e6 slots on one MI355X):
// one 64-bit global atomic per row runs at 27 G atomics/s whatever the operation (37 ms), a plain random 8-byte load per row at
// 114 G/s (8.8 ms), min through "load, compare, atomic only on improvement" at 18 ms -- per-row access to a G-length table in
// L2 / Infinity Cache cannot come near a streaming pass (12 B/row = 2.4 ms at 5 TB/s).  LDS atomics can: 1.7 ms for the same rows
// once a workgroup's slots fit LDS.  So the path is
//   1. ONE partition pass by the low b0 slot bits -- the first pass of the sort, with the offsets pdx_groupby_create already holds
//      (dense slots: radix_scatter_narrow, 4 + 8 B read, 2 + 8 B written per row; hash-partitioned slots: the value partition by the
//      bucket byte, 1 + 8 read, 8 written; count alone: keys only, 4 read + 2 written, k_acc_part_keys);
//   2. k_acc: every workgroup walks an equal span of the partitioned rows and keeps the <= S = nslots >> b0 accumulators of the bucket it
//      is in IN LDS (ds_min_u64 / ds_max_u64 / ds_add_u64 / ds_add_u32 on order-preserving images of the values), flushing them as one
//      partial block per (workgroup, bucket) segment;
//   3. k_acc_finish: one thread per group folds the <= ~3 partial blocks that cover its bucket and writes the outputs in group order.
// Small slot domains (nslots accumulators fit LDS) skip step 1 and read slot_of_row / the values where they stand (12 B/row).
// fp64 min / max: NaN rows only set a flag (all-NaN group -> NaN); -0.0 orders below +0.0 in the image, and a group whose extreme is a
// zero AND that holds zeros of both signs is "ambiguous": Arrow keeps the FIRST of tied minima / maxima (the LAST maximum when the
// group has a null, minmax.hpp), so k_acc_zero_scan records the first / last zero row of exactly those groups and k_acc_zero_apply
// patches the sign.  The scan only runs when the finish kernel counted an ambiguous group (one 4-byte read-back per call).
#pragma once

constexpr int kAccBlock = 1024;
constexpr size_t kAccLdsBudget = 156 * 1024;  // of the 160 KB of a CU: one 1024-thread workgroup per CU
enum : unsigned { kAccMin = 1u, kAccMax = 2u, kAccSum = 4u, kAccCnt = 8u };

struct AccTables {  // partial accumulators: one S-entry block per segment (= workgroup w inside bucket b, index w + b)
  unsigned long long* pmin;
  unsigned long long* pmax;
  unsigned long long* psum;
  uint32_t* pcnt;
  uint32_t* pflags;  // 4 bit tables of SW = (S + 31) / 32 words per segment: has +0.0, has -0.0, has NaN, has null
};
struct AccLds {
  int off_min, off_max, off_sum, off_cnt, off_flags, total;
};
__host__ __device__ inline AccLds acc_lds_layout(unsigned want, int S, bool flags) {
  AccLds l{};
  int o = 0;
  l.off_min = o;
  if (want & kAccMin) o += S * 8;
  l.off_max = o;
  if (want & kAccMax) o += S * 8;
  l.off_sum = o;
  if (want & kAccSum) o += S * 8;
  l.off_cnt = o;
  if (want & kAccCnt) o += S * 4;
  l.off_flags = o;
  if (flags) o += 4 * ((S + 31) / 32) * 4;
  l.total = o;
  return l;
}

// the bit tables ride along with min / max: fp64 (zeros, NaN, nulls) and nullable int64 (table 2 = "a valid value was seen": the
// sentinels of the min / max tables are legitimate int64 values)
__host__ __device__ inline bool acc_has_flags(unsigned want, bool is_f, bool nullable) { return (want & (kAccMin | kAccMax)) && (is_f || nullable); }

// order-preserving unsigned images (-0.0 < +0.0; NaNs never enter)
__device__ __forceinline__ unsigned long long acc_ord(double x) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ unsigned long long acc_ord(long long x) { return (unsigned long long)x ^ 0x8000000000000000ull; }
template <typename T>
__device__ __forceinline__ T acc_unord(unsigned long long o);
template <>
__device__ __forceinline__ double acc_unord<double>(unsigned long long o) {
  return __longlong_as_double((long long)((o >> 63) ? (o & 0x7FFFFFFFFFFFFFFFull) : ~o));
}
template <>
__device__ __forceinline__ long long acc_unord<long long>(unsigned long long o) { return (long long)(o ^ 0x8000000000000000ull); }

__global__ void k_acc_bstart(const uint32_t* __restrict__ row0, int B, int64_t n, uint32_t* __restrict__ out) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d < B) out[d] = row0 ? row0[d] : 0u;
  if (d == B) out[B] = (uint32_t)n;
}

// Keys-only partition by the low BITS slot bits (count needs no values): rows of a tile are ranked with one LDS atomic per row --
// the partition need not be stable, nothing here depends on row order -- staged in output order and written as runs.
// out[p] = (slot >> BITS) | (null ? 0x8000 : 0).  offsets: the scanned [tiles][1 << BITS] histogram of the same digit.
template <int BITS, bool NULLABLE>
__global__ void __launch_bounds__(kSortBlock) k_acc_part_keys(const uint32_t* __restrict__ slot, const uint8_t* __restrict__ valid, int64_t voff, int64_t n,
                                                              const uint32_t* __restrict__ offsets, uint16_t* __restrict__ out, int xcd_swizzle) {
  constexpr int R = 1 << BITS;
  __shared__ uint32_t hist[R], gbase[R], stage[kSortTile], scan_smem[8];
  const int tid = threadIdx.x;
  int64_t tile = blockIdx.x;
  if (xcd_swizzle) {
    const int64_t per = (int64_t)gridDim.x >> 3;
    if (tile < per * 8) tile = (tile & 7) * per + (tile >> 3);
  }
  const int64_t base = tile * kSortTile;
  const int rows = (int)((n - base) < kSortTile ? (n - base) : kSortTile);
  for (int d = tid; d < R; d += kSortBlock) hist[d] = 0;
  __syncthreads();
  uint32_t key[kSortItems], rank[kSortItems];
#pragma unroll
  for (int s = 0; s < kSortItems; ++s) {
    const int r = s * kSortBlock + tid;
    key[s] = r < rows ? slot[base + r] : 0u;
  }
#pragma unroll
  for (int s = 0; s < kSortItems; ++s) {
    const int r = s * kSortBlock + tid;
    if (r < rows) {
      if (NULLABLE && !bit_get(valid, voff + base + r)) key[s] |= 0x80000000u;
      rank[s] = atomicAdd(&hist[key[s] & (R - 1)], 1u);
    }
  }
  __syncthreads();
  static_assert(R <= kSortBlock, "one digit per thread");
  const uint32_t mine = tid < R ? hist[tid] : 0u;
  uint32_t total;
  const uint32_t pre = block_exclusive_scan(mine, SumOp(), &total, scan_smem);
  if (tid < R) {
    hist[tid] = pre;
    gbase[tid] = offsets[tile * R + tid] - pre;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < kSortItems; ++s) {
    const int r = s * kSortBlock + tid;
    if (r < rows) {
      const uint32_t d = key[s] & (R - 1);
      stage[hist[d] + rank[s]] = ((key[s] & 0x7FFFFFFFu) >> BITS) | ((key[s] >> 31) << 15) | (d << 16);
    }
  }
  __syncthreads();
  for (int p = tid; p < rows; p += kSortBlock) {
    const uint32_t w = stage[p];
    out[gbase[w >> 16] + (uint32_t)p] = (uint16_t)w;
  }
}

template <typename KT>
__device__ __forceinline__ void acc_load4(const KT* p, uint32_t k[4]);
template <>
__device__ __forceinline__ void acc_load4<uint16_t>(const uint16_t* p, uint32_t k[4]) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  k[0] = v.x & 0xFFFFu; k[1] = v.x >> 16; k[2] = v.y & 0xFFFFu; k[3] = v.y >> 16;
}
template <>
__device__ __forceinline__ void acc_load4<uint32_t>(const uint32_t* p, uint32_t k[4]) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w;
}
template <>
__device__ __forceinline__ void acc_load4<uint8_t>(const uint8_t* p, uint32_t k[4]) {
  const uint32_t v = *reinterpret_cast<const uint32_t*>(p);
  k[0] = v & 0xFFu; k[1] = (v >> 8) & 0xFFu; k[2] = (v >> 16) & 0xFFu; k[3] = v >> 24;
}

// The walk shared by k_acc and k_acc_zero_scan: workgroup w owns positions [w * span, (w + 1) * span) of the partitioned rows and
// visits the buckets that intersect them in order.  fn_begin(b) / fn_row(key, value bits, position, b) / fn_end(b, w + b).
// Keys: KT = uint16 / uint8: local slot index, top bit = null flag when NULLABLE; KT = uint32 (no partition, B == 1): the slot itself,
// nulls read from the validity bitmap at the row (= position).
template <typename KT, bool NULLABLE, typename Begin, typename Row, typename End>
__device__ __forceinline__ void acc_walk(const KT* __restrict__ keys, const uint64_t* __restrict__ vals, const uint8_t* __restrict__ valid, int64_t voff,
                                         int64_t n, const uint32_t* __restrict__ bstart, int B, int64_t span, bool vec, Begin fn_begin, Row fn_row, End fn_end) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int64_t w = blockIdx.x;
  const int64_t s0 = w * span, s1 = (s0 + span < n) ? s0 + span : n;
  if (s0 >= s1) return;
  int b = 0;
  if (B > 1) {  // the last bucket that starts at or before s0 (bstart[0] == 0)
    int lo = 0, hi = B;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int64_t)bstart[mid] <= s0) lo = mid;
      else hi = mid;
    }
    b = lo;
  }
  constexpr uint32_t kNullBit = sizeof(KT) == 4 ? 0u : (1u << (8 * sizeof(KT) - 1));
  auto one = [&](int64_t pos, uint32_t kraw, uint64_t v) {
    bool isnull = false;
    uint32_t k = kraw;
    if (NULLABLE) {
      if (sizeof(KT) == 4) isnull = !bit_get(valid, voff + pos);
      else {
        isnull = (kraw & kNullBit) != 0;
        k = kraw & (kNullBit - 1u);
      }
    }
    fn_row(k, v, pos, isnull);
  };
  for (; b < B; ++b) {
    const int64_t bs = bstart[b], be = bstart[b + 1];
    if (bs >= s1) break;
    const int64_t lo = s0 > bs ? s0 : bs, hi = s1 < be ? s1 : be;
    if (lo >= hi) continue;
    fn_begin(b);
    if (vec) {
      int64_t lo4 = (lo + 3) & ~(int64_t)3;
      if (lo4 > hi) lo4 = hi;
      int64_t hi4 = hi & ~(int64_t)3;
      if (hi4 < lo4) hi4 = lo4;
      if (tid < (int)(lo4 - lo)) one(lo + tid, (uint32_t)keys[lo + tid], vals ? vals[lo + tid] : 0ull);
      if (tid < (int)(hi - hi4)) one(hi4 + tid, (uint32_t)keys[hi4 + tid], vals ? vals[hi4 + tid] : 0ull);
      for (int64_t i = lo4 + (int64_t)tid * 4; i < hi4; i += (int64_t)nthr * 4) {
        uint32_t k[4];
        acc_load4<KT>(keys + i, k);
        ulonglong2 a = make_ulonglong2(0, 0), c = make_ulonglong2(0, 0);
        if (vals) {
          a = *reinterpret_cast<const ulonglong2*>(vals + i);
          c = *reinterpret_cast<const ulonglong2*>(vals + i + 2);
        }
        one(i, k[0], a.x);
        one(i + 1, k[1], a.y);
        one(i + 2, k[2], c.x);
        one(i + 3, k[3], c.y);
      }
    } else {
      for (int64_t i = lo + tid; i < hi; i += nthr) one(i, (uint32_t)keys[i], vals ? vals[i] : 0ull);
    }
    fn_end(b, w + b);
  }
}

template <typename T, typename KT, bool NULLABLE>
__global__ void __launch_bounds__(kAccBlock) k_acc(const KT* __restrict__ keys, const uint64_t* __restrict__ vals, const uint8_t* __restrict__ valid, int64_t voff,
                                                   int64_t n, const uint32_t* __restrict__ bstart, int B, int S, int64_t span, int vec, unsigned want, AccTables P,
                                                   int combine) {
  extern __shared__ unsigned long long acc_lds[];
  constexpr bool F = __is_same(T, double);
  const bool flags = acc_has_flags(want, F, NULLABLE);
  const AccLds L = acc_lds_layout(want, S, flags);
  char* base = reinterpret_cast<char*>(acc_lds);
  unsigned long long* lmin = reinterpret_cast<unsigned long long*>(base + L.off_min);
  unsigned long long* lmax = reinterpret_cast<unsigned long long*>(base + L.off_max);
  unsigned long long* lsum = reinterpret_cast<unsigned long long*>(base + L.off_sum);
  uint32_t* lcnt = reinterpret_cast<uint32_t*>(base + L.off_cnt);
  uint32_t* fl = reinterpret_cast<uint32_t*>(base + L.off_flags);
  const int SW = (S + 31) >> 5;
  const int tid = threadIdx.x;
  const uint64_t* vptr = (want & (kAccMin | kAccMax | kAccSum)) ? vals : nullptr;
  acc_walk<KT, NULLABLE>(
      keys, vptr, valid, voff, n, bstart, B, span, vec != 0,
      [&](int) {
        for (int i = tid; i < S; i += kAccBlock) {
          if (want & kAccMin) lmin[i] = ~0ull;
          if (want & kAccMax) lmax[i] = 0ull;
          if (want & kAccSum) lsum[i] = 0ull;
          if (want & kAccCnt) lcnt[i] = 0u;
        }
        if (flags)
          for (int i = tid; i < 4 * SW; i += kAccBlock) fl[i] = 0u;
        __syncthreads();
      },
      [&](uint32_t k, uint64_t v, int64_t, bool isnull) {
        const uint32_t bit = 1u << (k & 31);
        // Hot keys: LDS atomics on ONE address serialise (a key holding 30 % of the rows fills its bucket: 64 turns per wave instruction).
        // The lanes that share the slot of the wave's first lane -- when there are at least eight of them -- combine their rows with
        // shuffles (the other lanes contribute identities) and their first lane applies ONE update per table; the rest of the wave goes
        // on lane by lane.  A compare, a ballot and a population count per row otherwise (`combine` = 0: diagnostic, always per lane).
        if (combine && __ballot(1) == ~0ull) {
          const uint32_t k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
          const bool mine = k == k0 && !(NULLABLE && isnull);
          const unsigned long long m = __ballot(mine);
          if (__popcll(m) >= 8) {
            const bool leader = (tid & 63) == __ffsll((long long)m) - 1;
            const uint32_t bit0 = 1u << (k0 & 31);
            if ((want & kAccCnt) && leader) atomicAdd(&lcnt[k0], (uint32_t)__popcll(m));
            if (want & kAccSum) {
              unsigned long long sacc = mine ? (unsigned long long)v : 0ull;
#pragma unroll
              for (int d = 32; d >= 1; d >>= 1) sacc += (unsigned long long)__shfl_xor((long long)sacc, d, 64);
              if (leader) atomicAdd(&lsum[k0], sacc);
            }
            if (want & (kAccMin | kAccMax)) {
              // min / max are idempotent: the sharing lanes READ the slot first (one address: a broadcast, no bank conflict) and send an
              // atomic only when their value improves it -- after a hot key's first rows almost none does.  (A shuffle reduction of two
              // 64-bit values costs as many LDS-crossbar trips as the serialised atomics it replaces: measured 5.2 vs 3.4 ms.)
              unsigned long long omn = ~0ull, omx = 0ull;
              if constexpr (F) {
                const double x = __longlong_as_double((long long)v);
                const bool nan = x != x;
                const unsigned long long bn = __ballot(mine && nan), bpz = __ballot(mine && x == 0.0 && (long long)v >= 0),
                                         bnz = __ballot(mine && x == 0.0 && (long long)v < 0);
                if (leader) {
                  if (bn) atomicOr(&fl[2 * SW + (k0 >> 5)], bit0);
                  if (bpz) atomicOr(&fl[k0 >> 5], bit0);
                  if (bnz) atomicOr(&fl[SW + (k0 >> 5)], bit0);
                }
                if (mine && !nan) omn = omx = acc_ord(x);
              } else {
                if (NULLABLE && leader && !(fl[2 * SW + (k0 >> 5)] & bit0)) atomicOr(&fl[2 * SW + (k0 >> 5)], bit0);
                if (mine) omn = omx = acc_ord((long long)v);
              }
              if ((want & kAccMin) && omn < lmin[k0]) atomicMin(&lmin[k0], omn);
              if ((want & kAccMax) && omx > lmax[k0]) atomicMax(&lmax[k0], omx);
            }
            if (mine) return;
          }
        }
        if (NULLABLE && isnull) {
          if (F && flags) atomicOr(&fl[3 * SW + (k >> 5)], bit);
          return;
        }
        if (want & kAccCnt) atomicAdd(&lcnt[k], 1u);
        if (want & kAccSum) atomicAdd(&lsum[k], (unsigned long long)v);
        if (want & (kAccMin | kAccMax)) {
          unsigned long long o;
          if constexpr (F) {
            const double x = __longlong_as_double((long long)v);
            if (x != x) {
              atomicOr(&fl[2 * SW + (k >> 5)], bit);
              return;
            }
            if (x == 0.0) atomicOr(&fl[((long long)v < 0 ? SW : 0) + (k >> 5)], bit);
            o = acc_ord(x);
          } else {
            if (NULLABLE && !(fl[2 * SW + (k >> 5)] & bit)) atomicOr(&fl[2 * SW + (k >> 5)], bit);  // a valid value was seen
            o = acc_ord((long long)v);
          }
          if (want & kAccMin) atomicMin(&lmin[k], o);
          if (want & kAccMax) atomicMax(&lmax[k], o);
        }
      },
      [&](int, int64_t seg) {
        __syncthreads();
        const int64_t o0 = seg * S;
        for (int i = tid; i < S; i += kAccBlock) {
          if (want & kAccMin) P.pmin[o0 + i] = lmin[i];
          if (want & kAccMax) P.pmax[o0 + i] = lmax[i];
          if (want & kAccSum) P.psum[o0 + i] = lsum[i];
          if (want & kAccCnt) P.pcnt[o0 + i] = lcnt[i];
        }
        if (flags)
          for (int i = tid; i < 4 * SW; i += kAccBlock) P.pflags[seg * 4 * SW + i] = fl[i];
        __syncthreads();
      });
}

// One thread per group: fold the partial blocks of the segments that cover the group's bucket, write the outputs in group order.
// amb[gid] (fp64 min / max): bit 0 = a tied-zero extreme needs the row scan, bit 1 = the group has a null, bit 2 / 3 = min / max is the one.
template <typename T>
__global__ void k_acc_finish(const uint32_t* __restrict__ occ_slot, const uint32_t* __restrict__ gid_of_occ, int64_t G, const uint32_t* __restrict__ bstart, int B,
                             int b0, int S, int64_t span, unsigned want, AccTables P, int nullable, SegOut out, uint8_t* __restrict__ ok,
                             uint8_t* __restrict__ amb, unsigned int* __restrict__ amb_count) {
  constexpr bool F = __is_same(T, double);
  const bool flags = acc_has_flags(want, F, nullable != 0);
  const int SW = (S + 31) >> 5;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < G; k += stride) {
    const uint32_t slot = occ_slot[k], gid = gid_of_occ ? gid_of_occ[k] : (uint32_t)k;
    const int b = (int)(slot & (uint32_t)(B - 1));
    const uint32_t i = slot >> b0;
    const int64_t bs = bstart[b], be = bstart[b + 1];
    unsigned long long mn = ~0ull, mx = 0ull, sm = 0ull, cnt = 0ull;
    uint32_t f[4] = {0u, 0u, 0u, 0u};
    if (be > bs)
      for (int64_t w = bs / span; w <= (be - 1) / span; ++w) {
        const int64_t seg = w + b, at = seg * S + i;
        if (want & kAccMin) { const unsigned long long v = P.pmin[at]; mn = v < mn ? v : mn; }
        if (want & kAccMax) { const unsigned long long v = P.pmax[at]; mx = v > mx ? v : mx; }
        if (want & kAccSum) sm += P.psum[at];
        if (want & kAccCnt) cnt += P.pcnt[at];
        if (flags)
          for (int t = 0; t < 4; ++t) f[t] |= (P.pflags[seg * 4 * SW + t * SW + (i >> 5)] >> (i & 31)) & 1u;
      }
    bool have_mn = mn != ~0ull, have_mx = mx != 0ull;
    bool any_valid = true;
    if (nullable) any_valid = (want & kAccCnt) ? cnt > 0 : (F ? (have_mn || have_mx || f[2]) : f[2] != 0u);
    if (!F) have_mn = have_mx = any_valid;  // (INT64_MAX / INT64_MIN are the sentinels' images: presence comes from the count / the flag)
    if (out.count && (want & kAccCnt)) out.count[gid] = (long long)cnt;
    if (out.sum_i && (want & kAccSum)) out.sum_i[gid] = (long long)sm;
    T none = T(0);
    if constexpr (F) none = __builtin_nan("");
    if (out.vmin && (want & kAccMin)) static_cast<T*>(out.vmin)[gid] = have_mn ? acc_unord<T>(mn) : none;
    if (out.vmax && (want & kAccMax)) static_cast<T*>(out.vmax)[gid] = have_mx ? acc_unord<T>(mx) : none;
    if (ok) ok[gid] = any_valid ? 1 : 0;
    if (F && flags) {
      // -0.0 < +0.0 in the image: a minimum of -0.0 in a group that also holds +0.0 (a maximum of +0.0 beside a -0.0) is a tie
      // Arrow settles by row order
      const bool min_tie = (want & kAccMin) && mn == acc_ord(-0.0) && f[0];
      const bool max_tie = (want & kAccMax) && mx == acc_ord(0.0) && f[1];
      const uint8_t a = (min_tie || max_tie) ? (uint8_t)(1u | (f[3] ? 2u : 0u) | (min_tie ? 4u : 0u) | (max_tie ? 8u : 0u)) : (uint8_t)0;
      amb[gid] = a;
      if (a) atomicAdd(amb_count, 1u);
    }
  }
}

// The rare second pass of fp64 min / max: first and last zero-valued row (position << 1 | sign bit) of every ambiguous group.
// The partition is stable and a group lives in one bucket, so positions order a group's rows as the rows themselves do.
template <typename KT, bool NULLABLE>
__global__ void __launch_bounds__(kAccBlock) k_acc_zero_scan(const KT* __restrict__ keys, const uint64_t* __restrict__ vals, const uint8_t* __restrict__ valid,
                                                             int64_t voff, int64_t n, const uint32_t* __restrict__ bstart, int B, int b0, int64_t span, int vec,
                                                             const uint32_t* __restrict__ gid_of_slot, const uint8_t* __restrict__ amb,
                                                             uint32_t* __restrict__ zfirst, uint32_t* __restrict__ zlast) {
  int cur = 0;
  acc_walk<KT, NULLABLE>(
      keys, vals, valid, voff, n, bstart, B, span, vec != 0, [&](int b) { cur = b; },
      [&](uint32_t k, uint64_t v, int64_t pos, bool isnull) {
        if (isnull || (v << 1) != 0ull) return;  // +0.0 / -0.0 only
        const uint32_t gid = gid_of_slot[(k << b0) | (uint32_t)cur];
        if (!(amb[gid] & 1u)) return;
        const uint32_t code = ((uint32_t)pos << 1) | (uint32_t)(v >> 63);
        atomicMin(&zfirst[gid], code);
        atomicMax(&zlast[gid], code);
      },
      [&](int, int64_t) {});
}
__global__ void k_acc_zero_apply(const uint8_t* __restrict__ amb, const uint32_t* __restrict__ zfirst, const uint32_t* __restrict__ zlast, int64_t G,
                                 double* __restrict__ vmin, double* __restrict__ vmax) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < G; g += stride) {
    const uint8_t a = amb[g];
    if (!(a & 1u)) continue;
    if ((a & 4u) && vmin) vmin[g] = (zfirst[g] & 1u) ? -0.0 : 0.0;
    if ((a & 8u) && vmax) vmax[g] = (((a & 2u) ? zlast[g] : zfirst[g]) & 1u) ? -0.0 : 0.0;  // a group with a null keeps the LAST tied maximum
  }
}

// ---------------------------------------------------------------- host side
struct AccTuning {
  bool on = true;                        // PDX_GROUPBY_ACC=0: never (every request takes the sorted layout)
  int64_t min_rows = (int64_t)1 << 20;   // PDX_ACC_MIN_ROWS: below this the classic path's handful of small kernels is as fast
  int wgs = kCUs;                        // PDX_ACC_WGS: workgroups of the accumulate pass (one per CU: > 80 KB of LDS each)
  bool sizes_cache = true;               // PDX_ACC_SIZES_CACHE=0 (benchmarks): count of a column without nulls is computed every time
  static AccTuning read() {
    AccTuning t;
    if (const char* e = getenv("PDX_GROUPBY_ACC")) t.on = e[0] != '0';
    if (const char* e = getenv("PDX_ACC_MIN_ROWS")) t.min_rows = atoll(e);
    if (const char* e = getenv("PDX_ACC_WGS")) if (atoi(e) > 0) t.wgs = atoi(e);
    if (const char* e = getenv("PDX_ACC_SIZES_CACHE")) t.sizes_cache = e[0] != '0';
    return t;
  }
};

// How the handle's rows reach LDS-sized buckets (ok == false: the request keeps the sorted layout)
struct AccGeom {
  bool ok = false;
  int mode = 0;     // 0: no partition (nslots accumulators fit LDS), 1: dense slots through pass 0 of the sort, 2: hash-partitioned slots
  int b0 = 0, B = 1, S = 0;
};
static AccGeom acc_geometry(const pdx_groupby* gb, bool nullable, unsigned want, bool is_f, const AccTuning& t) {
  AccGeom g;
  if (!t.on || gb->mode != 0 || gb->n < t.min_rows || gb->G < 1) return g;
  auto fits = [&](int S, unsigned kind) { return (size_t)acc_lds_layout(kind, S, acc_has_flags(kind, is_f, nullable)).total <= kAccLdsBudget; };
  // every requested kind must fit on its own (several kinds that do not fit together take several accumulate passes)
  auto all_fit = [&](int S) {
    for (unsigned kind : {kAccMin, kAccMax, kAccSum, kAccCnt})
      if ((want & kind) && !fits(S, kind)) return false;
    return !nullable || !(want & kAccSum) || fits(S, kAccSum | kAccCnt);  // (a nullable sum carries the valid count along)
  };
  if (gb->dense && gb->slot_of_row) {
    if (gb->nslots <= 0x7FFFFFFF && all_fit((int)gb->nslots)) {
      g.ok = true;
      g.mode = 0;
      g.b0 = 0;
      g.B = 1;
      g.S = (int)gb->nslots;
      return g;
    }
    if (!gb->pass0_off) return g;
    const int b0 = make_sort_plan(gb->slot_bits, sort_max_bits()).bits[0];  // the digit pdx_groupby_create's histogram belongs to
    if (b0 < 4 || b0 > 8 || gb->slot_bits - b0 > (nullable ? 15 : 16)) return g;
    const int S = (int)(((gb->nslots - 1) >> b0) + 1);
    if (!all_fit(S)) return g;
    g.ok = true;
    g.mode = 1;
    g.b0 = b0;
    g.B = 1 << b0;
    g.S = S;
    return g;
  }
  if (gb->slot_part && gb->idx16_part && !gb->digit2 && !gb->special_slots && !nullable && gb->part_bits == kPartBits && gb->bucket8 && gb->part_off) {
    const int64_t cap = gb->nslots - 2;
    const int S = (int)(cap >> kPartBits);
    if (S < 1 || S > kLdsRegionMax || !all_fit(S)) return g;
    g.ok = true;
    g.mode = 2;
    g.b0 = kPartBits;
    g.B = 1 << kPartBits;
    g.S = S;
  }
  return g;
}

// count of a column without nulls = the group sizes: computed once per handle (keys only), kept in the handle
static long long* acc_sizes_cache(pdx_groupby* gb) {
  if (!gb->sizes) gb->sizes = gb->own<long long>((size_t)gb->G);
  return gb->sizes;
}

template <typename T, typename KT, bool NULLABLE>
static int acc_launch(const KT* keys, const uint64_t* vals, const uint8_t* valid, int64_t voff, int64_t n, const uint32_t* bstart, const AccGeom& g, int64_t span,
                      int W, bool vec, unsigned want, const AccTables& P, hipStream_t st) {
  const size_t lds = (size_t)acc_lds_layout(want, g.S, acc_has_flags(want, __is_same(T, double), NULLABLE)).total;
  auto kfn = k_acc<T, KT, NULLABLE>;
  PDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  static const int combine = [] { const char* e = getenv("PDX_ACC_WAVE_COMBINE"); return (e && e[0] == '0') ? 0 : 1; }();
  hipLaunchKernelGGL(kfn, dim3((unsigned)W), dim3(kAccBlock), lds, st, keys, vals, valid, voff, n, bstart, g.B, g.S, span, vec ? 1 : 0, want, P, combine);
  PDX_LAUNCH_CHECK();
  return PDX_OK;
}

// The order-free kinds of one request: count / min / max / int64 sum of `values` into `oo` (group order), ok_bytes (values with nulls):
// 1 = the group has a valid value.  Leaves the stream un-synchronised except for the one read-back of the tie counter.
static int reduce_acc(pdx_groupby* gb, const AccGeom& g, const pdx_column* values, const SegOut& oo, uint8_t* ok_bytes, const AccTuning& t, Scratch& s,
                      hipStream_t st, std::string* plan) {
  const int64_t n = gb->n, G = gb->G;
  const bool is_f = values->dtype == PDX_FLOAT64;
  const uint8_t* vvalid = validity_or_null(values);
  const bool nullable = vvalid != nullptr;
  const uint64_t* vin = static_cast<const uint64_t*>(values->values) + values->offset;
  unsigned want = 0;
  if (oo.vmin) want |= kAccMin;
  if (oo.vmax) want |= kAccMax;
  if (oo.sum_i) want |= kAccSum;
  if (oo.count) want |= kAccCnt;
  if (!want) return fail(PDX_INVALID, "internal: reduce_acc without an order-free kind");
  SegOut o = oo;
  // count of a column without nulls = the group sizes, whatever the column: served from the handle once known
  bool sizes_fill = false;
  if ((want & kAccCnt) && !nullable) {
    if (gb->sizes_ready && t.sizes_cache) {
      PDX_HIP(hipMemcpyAsync(oo.count, gb->sizes, (size_t)G * 8, hipMemcpyDeviceToDevice, st));
      want &= ~kAccCnt;
      o.count = nullptr;
    } else {
      sizes_fill = acc_sizes_cache(gb) != nullptr;
    }
  }
  std::string desc = std::string("slots=") + slots_name(gb);
  if (!want) {
    if (plan) *plan = desc + " sort=none layout=none reducer=sizes_cache";
    return PDX_OK;
  }
  const bool need_vals = (want & (kAccMin | kAccMax | kAccSum)) != 0;
  // ---- step 1: the rows in LDS-sized buckets
  const void* keys = nullptr;
  int key_bytes = 4;
  const uint64_t* vals = nullptr;
  uint32_t* bstart = s.get<uint32_t>((size_t)g.B + 1);
  PDX_SCRATCH_CHECK(s);
  if (g.mode == 0) {
    keys = gb->slot_of_row;
    vals = need_vals ? vin : nullptr;
    hipLaunchKernelGGL(k_acc_bstart, dim3(1), dim3(64), 0, st, (const uint32_t*)nullptr, 1, n, bstart);
    desc += " sort=none";
  } else if (g.mode == 1) {
    uint16_t* k16 = s.get<uint16_t>((size_t)n + 8);
    uint64_t* nv = need_vals ? s.get<uint64_t>((size_t)n + 8) : nullptr;
    PDX_SCRATCH_CHECK(s);
    int rc = PDX_OK;
    if (need_vals) {
#define ACC_P0(B)                                                                                                                                  \
  rc = nullable ? radix_scatter_narrow<B, uint64_t, uint32_t, uint16_t, true>(gb->slot_of_row, vin, k16, nv, n, gb->pass0_off, st, vvalid, values->offset) \
                : radix_scatter_narrow<B, uint64_t, uint32_t, uint16_t>(gb->slot_of_row, vin, k16, nv, n, gb->pass0_off, st)
      switch (g.b0) {
        case 4: ACC_P0(4); break;
        case 5: ACC_P0(5); break;
        case 6: ACC_P0(6); break;
        case 7: ACC_P0(7); break;
        default: ACC_P0(8); break;
      }
#undef ACC_P0
      desc += " sort=part:" + std::to_string(g.b0);
    } else {
      PDX_PROFILE("acc_partition_keys", st);
      const unsigned ntiles = (unsigned)ceil_div(n, kSortTile);
      const int swz = sort_xcd_swizzle();
#define ACC_PK(B)                                                                                                                                     \
  if (nullable) hipLaunchKernelGGL((k_acc_part_keys<B, true>), dim3(ntiles), dim3(kSortBlock), 0, st, gb->slot_of_row, vvalid, values->offset, n, gb->pass0_off, k16, swz); \
  else hipLaunchKernelGGL((k_acc_part_keys<B, false>), dim3(ntiles), dim3(kSortBlock), 0, st, gb->slot_of_row, vvalid, values->offset, n, gb->pass0_off, k16, swz)
      switch (g.b0) {
        case 4: ACC_PK(4); break;
        case 5: ACC_PK(5); break;
        case 6: ACC_PK(6); break;
        case 7: ACC_PK(7); break;
        default: ACC_PK(8); break;
      }
#undef ACC_PK
      PDX_LAUNCH_CHECK();
      desc += " sort=part_keys:" + std::to_string(g.b0);
    }
    PDX_TRY(rc);
    keys = k16;
    key_bytes = 2;
    vals = nv;
    hipLaunchKernelGGL(k_acc_bstart, dim3(ceil_div(g.B + 1, 256)), dim3(256), 0, st, gb->pass0_off, g.B, n, bstart);
  } else {
    uint64_t* nv = need_vals ? s.get<uint64_t>((size_t)n + 8) : nullptr;
    PDX_SCRATCH_CHECK(s);
    if (need_vals) PDX_TRY((radix_scatter_only<kPartBits, uint64_t, uint8_t>(gb->bucket8, vin, nullptr, nv, n, 0, false, gb->part_off, st)));
    keys = gb->idx16_part;
    key_bytes = 2;
    vals = nv;
    hipLaunchKernelGGL(k_acc_bstart, dim3(ceil_div(g.B + 1, 256)), dim3(256), 0, st, gb->part_off, g.B, n, bstart);
    desc += need_vals ? " sort=part:8" : " sort=none";
  }
  PDX_LAUNCH_CHECK();
  // ---- step 2 + 3: accumulate passes (the kinds that fit LDS together share one) and their finish kernels
  const int W = (int)std::max<int64_t>(1, std::min<int64_t>(t.wgs, ceil_div(n, 8192)));
  const int64_t span = round_up(ceil_div(n, W), 8);
  const int64_t nseg = (int64_t)W + g.B;
  const bool vec = ((reinterpret_cast<uintptr_t>(keys) & 15) == 0) && (!vals || (reinterpret_cast<uintptr_t>(vals) & 15) == 0);
  std::vector<unsigned> passes;
  {
    unsigned cur = 0;
    const unsigned order[4] = {kAccMin, kAccMax, kAccSum, kAccCnt};
    for (unsigned kind : order) {
      if (!(want & kind)) continue;
      const unsigned both = cur | kind;
      if (cur && (size_t)acc_lds_layout(both, g.S, acc_has_flags(both, is_f, nullable)).total > kAccLdsBudget) {
        passes.push_back(cur);
        cur = kind;
      } else {
        cur = both;
      }
    }
    if (cur) passes.push_back(cur);
    // values with nulls: a pass must be able to tell whether a group has a valid value (min / max tell; a lone sum needs the count)
    if (nullable)
      for (unsigned& p : passes)
        if (!(p & (kAccMin | kAccMax | kAccCnt))) p |= kAccCnt;
  }
  const int SW = (g.S + 31) >> 5;
  unsigned int* amb_count = s.get<unsigned int>(passes.size());
  PDX_SCRATCH_CHECK(s);
  PDX_HIP(hipMemsetAsync(amb_count, 0, passes.size() * sizeof(unsigned int), st));
  std::vector<uint8_t*> amb(passes.size(), nullptr);
  for (size_t pi = 0; pi < passes.size(); ++pi) {
    const unsigned p = passes[pi];
    const bool flags = acc_has_flags(p, is_f, nullable);
    AccTables P{};
    if (p & kAccMin) P.pmin = s.get<unsigned long long>((size_t)nseg * g.S);
    if (p & kAccMax) P.pmax = s.get<unsigned long long>((size_t)nseg * g.S);
    if (p & kAccSum) P.psum = s.get<unsigned long long>((size_t)nseg * g.S);
    if (p & kAccCnt) P.pcnt = s.get<uint32_t>((size_t)nseg * g.S);
    if (flags) {
      P.pflags = s.get<uint32_t>((size_t)nseg * 4 * SW);
      if (is_f) amb[pi] = s.get<uint8_t>((size_t)G);
    }
    PDX_SCRATCH_CHECK(s);
    {
      PDX_PROFILE("acc_reduce", st);
      int rc = PDX_OK;
#define ACC_GO(TT)                                                                                                                                      \
  if (key_bytes == 4) rc = nullable ? acc_launch<TT, uint32_t, true>(static_cast<const uint32_t*>(keys), vals, vvalid, values->offset, n, bstart, g, span, W, vec, p, P, st) \
                                    : acc_launch<TT, uint32_t, false>(static_cast<const uint32_t*>(keys), vals, vvalid, values->offset, n, bstart, g, span, W, vec, p, P, st); \
  else rc = nullable ? acc_launch<TT, uint16_t, true>(static_cast<const uint16_t*>(keys), vals, vvalid, values->offset, n, bstart, g, span, W, vec, p, P, st)     \
                     : acc_launch<TT, uint16_t, false>(static_cast<const uint16_t*>(keys), vals, vvalid, values->offset, n, bstart, g, span, W, vec, p, P, st)
      if (is_f) { ACC_GO(double); } else { ACC_GO(long long); }
#undef ACC_GO
      PDX_TRY(rc);
    }
    {
      PDX_PROFILE("acc_finish", st);
      SegOut po{};
      if (p & kAccMin) po.vmin = o.vmin;
      if (p & kAccMax) po.vmax = o.vmax;
      if (p & kAccSum) po.sum_i = o.sum_i;
      if (p & kAccCnt) po.count = o.count;
      long long* cnt_tmp = nullptr;
      (void)cnt_tmp;
      const int grid = grid_for(G, 256);
      if (is_f)
        hipLaunchKernelGGL((k_acc_finish<double>), dim3(grid), dim3(256), 0, st, gb->occ_slot, gb->gid_of_occ, G, bstart, g.B, g.b0, g.S, span, p, P, nullable ? 1 : 0,
                           po, ok_bytes, amb[pi], amb_count + pi);
      else
        hipLaunchKernelGGL((k_acc_finish<long long>), dim3(grid), dim3(256), 0, st, gb->occ_slot, gb->gid_of_occ, G, bstart, g.B, g.b0, g.S, span, p, P,
                           nullable ? 1 : 0, po, ok_bytes, amb[pi], amb_count + pi);
      PDX_LAUNCH_CHECK();
    }
  }
  if (sizes_fill && o.count) {
    PDX_HIP(hipMemcpyAsync(gb->sizes, o.count, (size_t)G * 8, hipMemcpyDeviceToDevice, st));
    gb->sizes_ready = true;
  }
  // ---- the tie counter: one small read-back; the row scan runs only for inputs that hold such a group
  std::string tie_note;
  if (is_f && (want & (kAccMin | kAccMax))) {
    std::vector<unsigned int> h(passes.size(), 0u);
    PDX_HIP(hipMemcpyAsync(h.data(), amb_count, passes.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    PDX_HIP(hipStreamSynchronize(st));
    for (size_t pi = 0; pi < passes.size(); ++pi) {
      if (!h[pi]) continue;
      PDX_PROFILE("acc_zero_ties", st);
      uint32_t* zfirst = s.get<uint32_t>((size_t)G);
      uint32_t* zlast = s.get<uint32_t>((size_t)G);
      PDX_SCRATCH_CHECK(s);
      PDX_HIP(hipMemsetAsync(zfirst, 0xFF, (size_t)G * 4, st));
      PDX_HIP(hipMemsetAsync(zlast, 0, (size_t)G * 4, st));
#define ACC_ZS(KT, NN)                                                                                                                                \
  hipLaunchKernelGGL((k_acc_zero_scan<KT, NN>), dim3((unsigned)W), dim3(kAccBlock), 0, st, static_cast<const KT*>(keys), vals, vvalid, values->offset, n, bstart, g.B, \
                     g.b0, span, vec ? 1 : 0, gb->gid_of_slot, amb[pi], zfirst, zlast)
      if (key_bytes == 4) { if (nullable) ACC_ZS(uint32_t, true); else ACC_ZS(uint32_t, false); }
      else { if (nullable) ACC_ZS(uint16_t, true); else ACC_ZS(uint16_t, false); }
#undef ACC_ZS
      hipLaunchKernelGGL(k_acc_zero_apply, dim3(grid_for(G, 256)), dim3(256), 0, st, amb[pi], zfirst, zlast, G,
                         (passes[pi] & kAccMin) ? static_cast<double*>(o.vmin) : nullptr, (passes[pi] & kAccMax) ? static_cast<double*>(o.vmax) : nullptr);
      PDX_LAUNCH_CHECK();
      tie_note = " zero_ties=" + std::to_string(h[pi]);
    }
  }
  if (plan) *plan = desc + " layout=buckets:" + std::to_string(g.B) + "x" + std::to_string(g.S) + " reducer=lds_acc passes=" + std::to_string(passes.size()) + tie_note;
  return PDX_OK;
}
