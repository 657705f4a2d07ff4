// gb_flr_reduce.hpp -- part of groupby.hip (textually included there, in its namespace context; split by stage, kernels unchanged):
// fused last digit (workgroup per run) -- the wave-per-run form lives in flr_wave.hpp.
#pragma once

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
#ifndef PDX_FLR_ITEMS
#define PDX_FLR_ITEMS 10
#endif
#ifndef PDX_FLR_DENSE_WAVES
#define PDX_FLR_DENSE_WAVES 1
#endif
constexpr int kFlrDenseWaves = PDX_FLR_DENSE_WAVES;  // diagnostic: minimum waves per SIMD the dense instantiation is compiled for
constexpr int kFlrItems = PDX_FLR_ITEMS;   // rows per thread and tile: 3072-row tiles (measured best: 4096 -> 5.6 ms, 3072 -> 4.2 ms, 2048 -> 4.5 ms per 1e9 rows)
constexpr int kFlrTile = kSortBlock * kFlrItems;
// Dense path: the staged rows are laid out BY LEAF, element pair e of leaf f at doubles [e][f][2] -- the 64 lanes of the leaf phase (one
// leaf each) read 16 consecutive bytes per lane and instruction, free of bank conflicts, where rows staged in rank order put the lanes
// 128 bytes apart (two bank groups for 64 lanes).  A tile touches at most tile / 16 + 64 * 30 / 16 leaves (every group may continue an
// open leaf and leave one open).
#ifdef PDX_FLR_TIMING  // diagnostic build only (tools/): cycles per phase of the dense kernel, wave 0 and wave 1, summed over all workgroups
__device__ unsigned long long g_flr_cycles[2][12];
#define FLR_T(slot)                                                                                     \
  do {                                                                                                  \
    if (DENSE_PW && wave < 2) {                                                                         \
      const unsigned long long now_ = __builtin_readcyclecounter();                                     \
      tacc_[slot] += now_ - tmark_;                                                                     \
      tmark_ = now_;                                                                                    \
    }                                                                                                   \
  } while (0)
#else
#define FLR_T(slot) do { } while (0)
#endif
constexpr int kFlrMaxLeaves = kFlrTile / 16 + ((1 << kFlrBits) * 30) / 16;
constexpr int kFlrLeafStride = kFlrMaxLeaves + 1;
constexpr int kFlrDenseLevels = 16;  // the dense path has no nulls: a run of <= 2^19 rows (kFlrMaxRun) holds <= 2^15 leaves
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
// out[0] = rows of the longest run, out[1] = runs longer than `limit`, out[2] = their rows together
__global__ void k_run_max_len(const uint32_t* __restrict__ run_start, int64_t nruns, unsigned int* __restrict__ out, unsigned int limit) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned int m = 0, nl = 0, rl = 0;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nruns; r += stride) {
    unsigned int len = run_start[r + 1] - run_start[r];
    m = len > m ? len : m;
    if (len > limit) {
      ++nl;
      rl += len;
    }
  }
  for (int d = 32; d >= 1; d >>= 1) {
    unsigned int o = __shfl_xor(m, d, 64);
    m = o > m ? o : m;
    nl += __shfl_xor(nl, d, 64);
    rl += __shfl_xor(rl, d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (m) atomicMax(&out[0], m);
    if (nl) {
      atomicAdd(&out[1], nl);
      atomicAdd(&out[2], rl);
    }
  }
}
// the long runs, unordered (at most `cap` are kept; the host has the count already)
__global__ void k_long_runs_append(const uint32_t* __restrict__ run_start, int64_t nruns, unsigned int limit, unsigned int cap,
                                   uint32_t* __restrict__ list, unsigned int* __restrict__ count) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nruns; r += stride)
    if (run_start[r + 1] - run_start[r] > limit) {
      const unsigned int i = atomicAdd(count, 1u);
      if (i < cap) list[i] = (uint32_t)r;
    }
}
// ... in ascending run order, with the start of every run's rows in the side array (off[nl] = all of them); one workgroup of 256
constexpr int kMaxLongRuns = 256;
__global__ void __launch_bounds__(256) k_long_runs_order(const uint32_t* __restrict__ unordered, int nl, const uint32_t* __restrict__ run_start,
                                                         uint32_t* __restrict__ list, uint32_t* __restrict__ off) {
  __shared__ uint32_t a[kMaxLongRuns], b[kMaxLongRuns], len[kMaxLongRuns];
  const int t = threadIdx.x;
  if (t < nl) a[t] = unordered[t];
  __syncthreads();
  if (t < nl) {
    int rank = 0;
    for (int j = 0; j < nl; ++j) rank += a[j] < a[t];
    b[rank] = a[t];
  }
  __syncthreads();
  if (t < nl) {
    list[t] = b[t];
    len[t] = run_start[b[t] + 1] - run_start[b[t]];
  }
  __syncthreads();
  if (t == 0) {
    uint32_t acc = 0;
    for (int j = 0; j < nl; ++j) {
      off[j] = acc;
      acc += len[j];
    }
    off[nl] = acc;
  }
}
// side_key[t] = the full slot of the t-th row of the long runs (| bit 31: null value), side_val[t] = its value.  KT = uint8: the top
// digit alone (bit 7 = null flag) after a narrowing sort; uint32: the slot itself (bit 31 = null flag)
template <typename KT>
__global__ void __launch_bounds__(256) k_side_gather(const KT* __restrict__ keys, const uint64_t* __restrict__ vals, const uint32_t* __restrict__ run_start,
                                                     const uint32_t* __restrict__ list, const uint32_t* __restrict__ off, int nl, int low_bits, int64_t m,
                                                     uint32_t* __restrict__ side_key, uint64_t* __restrict__ side_val) {
  __shared__ uint32_t soff[kMaxLongRuns + 1], srun[kMaxLongRuns], sstart[kMaxLongRuns];
  for (int j = threadIdx.x; j <= nl; j += 256) {
    soff[j] = off[j];
    if (j < nl) {
      srun[j] = list[j];
      sstart[j] = run_start[list[j]];
    }
  }
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < m; t += stride) {
    int lo = 0, hi = nl - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if ((int64_t)soff[mid] <= t) lo = mid;
      else hi = mid - 1;
    }
    const int64_t src = (int64_t)sstart[lo] + (t - (int64_t)soff[lo]);
    const uint32_t k = keys[src];
    uint32_t slot;
    if (sizeof(KT) == 1) slot = (((k & ((1u << kFlrBits) - 1u)) << low_bits) | srun[lo]) | ((k >> 7) << 31);
    else slot = k;
    side_key[t] = slot;
    side_val[t] = vals[src];
  }
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
  double v = pw_merge(csum[0][lane], leaf);
  cmask ^= m;
  while ((cmask & m) == 0) {
    csum[cur][lane] = 0.0;
    ++cur;
    m <<= 1;
    v = pw_merge(csum[cur][lane], v);
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
__global__ void __launch_bounds__(kSortBlock, (DENSE_PW ? kFlrDenseWaves : 1)) k_flr_reduce(const KT* __restrict__ keys, const T* __restrict__ vals,
                                                           const uint32_t* __restrict__ run_start, int64_t nruns, int low_bits,
                                                           const uint32_t* __restrict__ gid_of_slot, SegOut out, uint8_t* __restrict__ ok,
                                                           int want_pw, int want_mm, int want_is, int nullable,
                                                           const double* __restrict__ sqdev_mean, unsigned int max_run) {
  // max_run: runs longer than this are not this kernel's (the layout's side form holds their rows)
  // sqdev_mean != nullptr (second pass of variance): every value x of group g enters the sum as (x - sqdev_mean[g])^2, with the
  // reference's x86 NaN propagation (see k_seg_sqdev)
  constexpr int R = 1 << kFlrBits;
  // staged rows of digit d start at dstart[d] + d: the digits' regions are ~64 rows = 512 B apart, so without the skew the 64
  // lanes of the replay (one digit each) would hit the same LDS bank on every read (measured: 3x slower)
  __shared__ __attribute__((aligned(16))) T svals[DENSE_PW ? 16 * kFlrLeafStride : kFlrTile + R];
  __shared__ __attribute__((aligned(8))) uint8_t snull[kFlrTile + R];
  __shared__ uint32_t cnt[kSortWaves][R];
  __shared__ unsigned long long match[kSortWaves][R];  // match-any words of the ranking (wave_match_rank)
  __shared__ uint32_t dstart[R + 1];
  constexpr int kLevels = DENSE_PW ? kFlrDenseLevels : kFlrLevels;
  __shared__ double csum[kLevels][R];
  // dense sum/mean/count fast path (no nulls, no min/max/int sum): one THREAD per 16-value leaf, then one lane per group for the
  // few counter pushes -- the open leaf of every group (rows so far + their sequential sum) lives in LDS between tiles
  __shared__ int open_pos[R];
  __shared__ double open_acc[R];
  __shared__ double mu_s[R];
  __shared__ int lp[R + 1];
  // dense path: digit of every leaf of the tile (written by the digit's lane for its first leaf and by the row that opens any later one)
  // and, per digit, first leaf | rows in the open leaf << 12 | rows in this tile << 16 -- one read each in the leaf phase
  __shared__ uint8_t leaf_d[DENSE_PW ? kFlrLeafStride + 3 : 4];
  __shared__ uint32_t dinfo[R];
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
#ifdef PDX_FLR_TIMING
  unsigned long long tacc_[10] = {};
#endif
  for (int64_t run = blockIdx.x; run < nruns; run += gridDim.x) {
    const int64_t s = run_start[run], e = run_start[run + 1];
    if (s == e || e - s > (int64_t)max_run) continue;
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
    int tile_c = 0, tile_lp = 0;  // dense path, wave 0: this digit's rows and first leaf in the current tile
    unsigned long long cmask = 0, isum = 0;
    long long nvalid = 0, nrows = 0;
    T vmn = T(0), vmx = T(0);
    int zneg = -1;  // sign of the last zero-valued valid row (-1: none)
    bool has = false;
    if (wave == 0)
      for (int l = 0; l < kLevels; ++l) csum[l][lane] = 0.0;
    uint32_t key[kFlrItems];
    T val[kFlrItems];
    auto load_tile = [&](int64_t t0) {
      const int rows = (int)(e - t0 < kFlrTile ? e - t0 : kFlrTile);
#pragma unroll
      for (int q = 0; q < kFlrItems; ++q) {
        // (rows behind the tile's end read its last row instead of branching around the load: they are never ranked or staged)
        const int r = wave * (64 * kFlrItems) + q * 64 + lane;
        const int rr = r < rows ? r : rows - 1;
        key[q] = keys[t0 + rr];
        val[q] = vals[t0 + rr];
      }
    };
    load_tile(s);
#ifdef PDX_FLR_TIMING
    unsigned long long tmark_ = __builtin_readcyclecounter();
#endif
    for (int64_t t0 = s; t0 < e; t0 += kFlrTile) {
      const int rows = (int)(e - t0 < kFlrTile ? e - t0 : kFlrTile);
      uint32_t rank[kFlrItems];
      FLR_T(0);  // wave 0: the previous tile's pushes; others: nothing
#pragma unroll
      for (int q = 0; q < kFlrItems; ++q) {
        const int r = wave * (64 * kFlrItems) + q * 64 + lane;
        const bool active = r < rows;
        const uint32_t d = sizeof(KT) == 4 ? ((key[q] & kSortKeyMask) >> low_bits) & (R - 1) : key[q] & (R - 1);
        rank[q] = wave_match_rank(match[wave], cnt[wave], d, active, lane, lt_mask);
      }
      FLR_T(1);  // ranking (includes waiting for this tile's loads)
      __syncthreads();
      FLR_T(2);
      if (tid < R) {  // exclusive prefix over waves per digit, then over digits (64 values: one wave)
        // (the four counters are read in one batch and written once: every dependent LDS round trip here is time the other three
        //  waves spend at the barrier)
        uint32_t cw[kSortWaves];
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) cw[w] = cnt[w][tid];
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
          const uint32_t c = cw[w];
          cw[w] = tot;
          tot += c;
        }
        uint32_t base, inc, ex;
        if (dense_pw) {
          // leaves touched by this tile, per group: the first one may continue the open leaf, the last one may stay open.  Rows and
          // leaves are scanned in one word (<= 2560 rows, <= 280 leaves)
          const int c = (int)tot;
          const uint32_t nl = c > 0 ? (uint32_t)(pos + c + 15) >> 4 : 0u;
          const uint32_t both = wave_inclusive_scan(tot | (nl << 16), SumOp());
          inc = both & 0xFFFFu;
          ex = inc - tot;
          const uint32_t lincl = both >> 16, lex = lincl - nl;
          // the staging base of a digit is its "leaf coordinate": 16 x first leaf + rows already in the open leaf, so base + rank =
          // 16 x leaf + element
          base = 16u * lex + (uint32_t)pos;
          tile_c = c;
          tile_lp = (int)lex;
          dinfo[tid] = lex | ((uint32_t)pos << 12) | ((uint32_t)c << 16);
          if (c > 0) leaf_d[lex] = (uint8_t)tid;
          if (tid == R - 1) lp[R] = (int)lincl;
          if (sqdev_mean && c > 0 && !mu_known) {
            mu = sqdev_mean[gid_of_slot[((uint32_t)lane << low_bits) | (uint32_t)run]];
            mu_s[lane] = mu;
            mu_known = true;
          }
          nrows += c;
          nvalid += c;
        } else {
          inc = wave_inclusive_scan(tot, SumOp());
          ex = inc - tot;
          base = ex;
          dstart[tid] = ex;
          if (tid == R - 1) dstart[R] = inc;
        }
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) cnt[w][tid] = cw[w] + base;
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
      }
      FLR_T(3);  // prefix
      __syncthreads();
      FLR_T(4);
      // (all bases are requested before the first staged write: reads behind a byte store would wait for it)
#pragma unroll
      for (int q = 0; q < kFlrItems; ++q) {
        const uint32_t d = sizeof(KT) == 4 ? ((key[q] & kSortKeyMask) >> low_bits) & (R - 1) : key[q] & (R - 1);
        rank[q] += cnt[wave][d];
      }
#pragma unroll
      for (int q = 0; q < kFlrItems; ++q) {
        const int r = wave * (64 * kFlrItems) + q * 64 + lane;
        if (r < rows) {
          const uint32_t d = sizeof(KT) == 4 ? ((key[q] & kSortKeyMask) >> low_bits) & (R - 1) : key[q] & (R - 1);
          uint32_t p = rank[q];
          if (dense_pw) {
            if ((p & 15u) == 0) leaf_d[p >> 4] = (uint8_t)d;  // this row opens a leaf
            p = ((p & 14u) >> 1) * (2 * kFlrLeafStride) + (p >> 4) * 2 + (p & 1u);
          } else {
            p += d;
          }
          svals[p] = val[q];
          if (nullable) snull[p] = (uint8_t)(key[q] >> (8 * (int)sizeof(KT) - 1));
        }
      }
      FLR_T(5);  // staging
      if (t0 + kFlrTile < e) load_tile(t0 + kFlrTile);  // in flight while wave 0 replays this tile
      FLR_T(9);  // issuing the next tile's loads
      __syncthreads();
      FLR_T(6);
      for (int d = tid; d < kSortWaves * R; d += kSortBlock) (&cnt[0][0])[d] = 0;  // (free again: the bases were consumed above)
      if (dense_pw) {
        const int NL = lp[R];
        for (int Lf = tid; Lf < NL; Lf += kSortBlock) {
          // element q of this leaf sits at [q >> 1][Lf][q & 1]; elements in front of q0 (summed into open_acc by an earlier tile) and
          // behind the leaf's last row are read and ignored.  All values are requested before the first add.
          const int d = leaf_d[Lf];
          double xs[16];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            struct alignas(16) Pair { T x, y; };
            const Pair pr = *reinterpret_cast<const Pair*>(svals + e * (2 * kFlrLeafStride) + 2 * Lf);
            xs[2 * e] = seg_to_f64(pr.x);
            xs[2 * e + 1] = seg_to_f64(pr.y);
          }
          const uint32_t di = dinfo[d];
          const double oa = open_acc[d];
          const int j = Lf - (int)(di & 0xFFFu), p0 = (int)((di >> 12) & 15u), c = (int)(di >> 16);
          const int q0 = j == 0 ? p0 : 0;
          int q1 = p0 + c - 16 * j;  // elements of the leaf filled so far
          q1 = q1 < 16 ? q1 : 16;
          double a = (j == 0 && p0 > 0) ? oa : 0.0;
          if (sqdev_mean) {
            const double m = mu_s[d];
#pragma unroll
            for (int q = 0; q < 16; ++q)
              if (q >= q0 && q < q1) a += flr_sqdev(xs[q], m);
          } else {
#pragma unroll
            for (int q = 0; q < 16; ++q)
              if (q >= q0 && q < q1) a += xs[q];
          }
          if (a != a) {  // (rare) the leaf again with the x86 NaN rule of a leaf: the earlier operand's NaN wins (pairwise.hpp)
            a = (j == 0 && p0 > 0) ? oa : 0.0;
            const double m = sqdev_mean ? mu_s[d] : 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q)
              if (q >= q0 && q < q1) a = pw_leaf_add(a, sqdev_mean ? flr_sqdev(xs[q], m) : xs[q]);
          }
          leafsum[Lf] = a;
        }
        FLR_T(7);  // leaf sums
        __syncthreads();
        FLR_T(8);
        if (wave == 0 && tile_c > 0) {
          // Arrow's binary counter over this tile's finished leaves.  Levels 0-2 live in registers for the duration (one batch of
          // reads, one of writes); a carry beyond them walks the LDS column as before (one leaf in eight).
          const int p0 = pos, c = tile_c;
          const int nl = (p0 + c + 15) >> 4, nfull = (p0 + c) >> 4;
          const double* lf = leafsum + tile_lp;
          const double last = lf[nl - 1];
          if (nfull > 0) {
            double c0 = csum[0][lane], c1 = csum[1][lane], c2 = csum[2][lane];
            double nxt = lf[0];
            for (int j = 0; j < nfull; ++j) {
              double v = nxt;
              if (j + 1 < nfull) nxt = lf[j + 1];
              v = pw_merge(c0, v);
              cmask ^= 1ull;
              if (cmask & 1ull) {
                c0 = v;
              } else {
                c0 = 0.0;
                v = pw_merge(c1, v);
                cmask ^= 2ull;
                if (cmask & 2ull) {
                  c1 = v;
                  root = root > 1 ? root : 1;
                } else {
                  c1 = 0.0;
                  v = pw_merge(c2, v);
                  cmask ^= 4ull;
                  if (cmask & 4ull) {
                    c2 = v;
                    root = root > 2 ? root : 2;
                  } else {
                    c2 = 0.0;
                    int cur = 3;
                    unsigned long long m = 8ull;
                    v = pw_merge(csum[3][lane], v);
                    cmask ^= m;
                    while ((cmask & m) == 0) {
                      csum[cur][lane] = 0.0;
                      ++cur;
                      m <<= 1;
                      v = pw_merge(csum[cur][lane], v);
                      cmask ^= m;
                    }
                    csum[cur][lane] = v;
                    root = cur > root ? cur : root;
                  }
                }
              }
            }
            csum[0][lane] = c0;
            csum[1][lane] = c1;
            csum[2][lane] = c2;
          }
          const int rem = (p0 + c) & 15;
          pos = rem;
          acc = rem ? last : 0.0;
          if (rem) open_acc[lane] = last;
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
              if (j < count) a = pw_leaf_add(a, sqdev_mean ? flr_sqdev(xs[j], m) : xs[j]);  // (opt-in NULL_PW form: per-add rule)
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
                acc = pw_leaf_add(pos == 0 ? 0.0 : acc, sqdev_mean ? flr_sqdev(seg_to_f64(x), mu) : seg_to_f64(x));
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
          for (int i = 1; i <= root; ++i) a = pw_merge(csum[i][lane], a);
          total = a;
        }
        if (out.sum_f) out.sum_f[oi] = total;
        if (out.mean) out.mean[oi] = nvalid ? pw_mean(total, (double)nvalid) : 0.0;
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
#ifdef PDX_FLR_TIMING
  if (DENSE_PW && wave < 2 && lane == 0)
    for (int i = 0; i < 10; ++i) atomicAdd(&g_flr_cycles[wave][i], tacc_[i]);
#endif
}
