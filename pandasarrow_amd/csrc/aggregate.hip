// aggregate.hip -- whole-array sum / mean / min / max / count for gfx950.
//
// Replaces CallFunction("sum"|"mean"|"min"|"max"|"count") reached from NDFrame::sum/mean/min/max/count
// (reference src/ndframe.cpp:26-31 macro, 119, 162-166, 220) and MinMax (src/resample.cpp:223).
// HBM-bound single pass: 8 B/row (+1/8 B validity).  fp64 sum/mean reproduce Arrow's pairwise tree exactly
// (see pairwise.hpp); integer sum wraps; integer mean sums the values converted to double with the same tree
// (Arrow 25.0.0 behaviour, pinned by tests/golden agg_i64_*); min/max skip NaN unless all values are NaN and keep
// the FIRST of tied values (0.0 vs -0.0), implemented as an order-independent (value, row) reduction.
#include <string.h>
#include "minmax.hpp"
#include "pairwise.hpp"
#include "scan.hpp"

namespace pdx {

constexpr int kLeafBlock = 256;                 // threads = leaves per block
constexpr int kLeafElems = kLeafBlock * 16;     // 4096 values per block
constexpr int kLeafPad = 17;                    // LDS stride per leaf (doubles): conflict-free ds_read_b64

template <typename T>
__device__ __forceinline__ double to_f64(T x) { return (double)x; }

// small device-to-host read (<= 64 bytes) through this thread's pinned slot: a copy into pageable memory is staged by the runtime
static int read_back(void* dst, const void* dev, size_t bytes, hipStream_t st) {
  void* pin = bytes <= 64 ? pinned_slot() : nullptr;
  PDX_HIP(hipMemcpyAsync(pin ? pin : dst, dev, bytes, hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
  if (pin) memcpy(dst, pin, bytes);
  return PDX_OK;
}

// ---------------------------------------------------------------- dense path, level 0
// block b: values [4096b, 4096b+4096) -> 256 leaf sums -> full block: one level-8 node, ragged last block: raw leaves
template <typename T>
__global__ void __launch_bounds__(kLeafBlock) k_sum_dense_level0(const T* __restrict__ v, int64_t n, double* __restrict__ nodes,
                                                                 double* __restrict__ tail) {
  __shared__ double lds[kLeafBlock * kLeafPad];
  __shared__ double red[4];
  const int t = threadIdx.x;
  int64_t base = (int64_t)blockIdx.x * kLeafElems;
  int64_t remain = n - base;
  if (remain >= kLeafElems) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      int idx = k * kLeafBlock + t;  // coalesced 8-byte lanes
      lds[idx + (idx >> 4)] = to_f64(v[base + idx]);
    }
    __syncthreads();
    double leaf = leaf_sum(&lds[t * kLeafPad], 16);
    double r = block_tree256(leaf, red);
    if (t == 0) nodes[blockIdx.x] = r;
  } else {
    for (int k = 0; k < 16; ++k) {
      int idx = k * kLeafBlock + t;
      if (idx < remain) lds[idx + (idx >> 4)] = to_f64(v[base + idx]);
    }
    __syncthreads();
    int64_t first = (int64_t)t * 16;
    if (first < remain) {
      int cnt = (int)((remain - first) < 16 ? (remain - first) : 16);
      tail[t] = leaf_sum(&lds[t * kLeafPad], cnt);
    }
  }
}

// one tree level: X[m] -> Y[j] = perfect tree of X[256j .. 256j+256) for full groups; the ragged rest is copied to `tail`
__global__ void __launch_bounds__(kLeafBlock) k_sum_tree_level(const double* __restrict__ x, int64_t m, double* __restrict__ y,
                                                               double* __restrict__ tail) {
  __shared__ double red[4];
  const int t = threadIdx.x;
  int64_t base = (int64_t)blockIdx.x * kLeafBlock;
  if (base + kLeafBlock <= m) {
    double r = block_tree256(x[base + t], red);
    if (t == 0) y[blockIdx.x] = r;
  } else if (base + t < m) {
    tail[t] = x[base + t];
  }
}

struct FinishArgs {
  const double* tails[9];  // tails[r]: ragged nodes of level 8r
  int counts[9];
  int nlevels;
};
// replay the counter over the ragged tails, top level first, then fold.  A level's tail (< 256 nodes, pushed when every lower
// counter level is empty) is what the counter turns into one perfect subtree per set bit of its length -- 128, 64, ... nodes,
// front to back -- so the wave builds those subtrees with shuffle trees and lane 0 pushes <= 8 finished nodes per level at their
// own levels, instead of replaying up to 255 single pushes through the private (scratch) counter array: ~100 us -> a few us per
// call, 7 % of a 1e9-row dense sum.
__global__ void __launch_bounds__(64) k_sum_finish(FinishArgs a, double* __restrict__ out) {
  __shared__ double t[9][256];
  const int lane = threadIdx.x;
  for (int r = 0; r < a.nlevels; ++r)
    for (int i = lane; i < a.counts[r]; i += 64) t[r][i] = a.tails[r][i];
  __syncthreads();
  PairwiseCounter c;
  c.init();
  bool any = false;
  for (int r = a.nlevels - 1; r >= 0; --r) {
    const int cnt = a.counts[r];
    int pos = 0;
    for (int b = 8; b >= 0; --b) {  // (a level-0 tail can hold exactly 256 leaves: 4081..4095 ragged rows)
      if (!((cnt >> b) & 1)) continue;
      double node;
      if (b == 8) {
        const double q0 = wave_tree64(t[r][lane]), q1 = wave_tree64(t[r][64 + lane]), q2 = wave_tree64(t[r][128 + lane]),
                     q3 = wave_tree64(t[r][192 + lane]);
        node = pw_merge(pw_merge(q0, q1), pw_merge(q2, q3));
      } else if (b == 7) {
        const double lo = wave_tree64(t[r][pos + lane]), hi = wave_tree64(t[r][pos + 64 + lane]);
        node = pw_merge(lo, hi);
      } else {
        node = wave_tree_levels(lane < (1 << b) ? t[r][pos + lane] : 0.0, b);
      }
      if (lane == 0) c.push(node, 8 * r + b);
      any = true;
      pos += 1 << b;
    }
  }
  if (lane == 0) *out = any ? c.finish() : 0.0;
}

// runs the tree over level arrays; level0_nodes = number of level-8 nodes already produced (dense) or, when
// `leaves` != nullptr, an explicit leaf array of `m_leaves` entries
static int run_tree(const double* leaves, int64_t m_leaves, double* nodes1, int64_t n_nodes1, double* tail0, int tail0_count,
                    double* result_dev, Scratch& s, hipStream_t st) {
  FinishArgs fa;
  fa.nlevels = 0;
  double* tails = s.get<double>(9 * 256);
  PDX_SCRATCH_CHECK(s);
  const double* cur = nullptr;
  int64_t cur_n = 0;
  int level = 0;
  if (leaves) {
    cur = leaves;
    cur_n = m_leaves;
  } else {
    // level 0 already done by k_sum_dense_level0
    fa.tails[0] = tail0;
    fa.counts[0] = tail0_count;
    cur = nodes1;
    cur_n = n_nodes1;
    level = 1;
  }
  while (true) {
    if (cur_n < 256) {
      fa.tails[level] = cur;
      fa.counts[level] = (int)cur_n;
      fa.nlevels = level + 1;
      break;
    }
    int64_t full = cur_n / 256;
    int rem = (int)(cur_n - full * 256);
    double* next = s.get<double>((size_t)full);
    PDX_SCRATCH_CHECK(s);
    double* tl = tails + level * 256;
    hipLaunchKernelGGL(k_sum_tree_level, dim3((unsigned)ceil_div(cur_n, 256)), dim3(kLeafBlock), 0, st, cur, cur_n, next, tl);
    fa.tails[level] = tl;
    fa.counts[level] = rem;
    cur = next;
    cur_n = full;
    ++level;
    if (level >= 8) return fail(PDX_INVALID, "pairwise tree too deep");
  }
  hipLaunchKernelGGL(k_sum_finish, dim3(1), dim3(64), 0, st, fa, result_dev);
  PDX_LAUNCH_CHECK();
  return PDX_OK;
}

template <typename T>
static int sum_dense(const T* v, int64_t n, double* result_dev, Scratch& s, hipStream_t st) {
  int64_t nblocks = ceil_div(n, kLeafElems);
  int64_t full = n / kLeafElems;
  int64_t rem = n - full * kLeafElems;
  double* nodes = s.get<double>((size_t)(full ? full : 1));
  double* tail0 = s.get<double>(256);
  PDX_SCRATCH_CHECK(s);
  hipLaunchKernelGGL((k_sum_dense_level0<T>), dim3((unsigned)nblocks), dim3(kLeafBlock), 0, st, v, n, nodes, tail0);
  PDX_LAUNCH_CHECK();
  return run_tree(nullptr, 0, nodes, full, tail0, (int)ceil_div(rem, 16), result_dev, s, st);
}

// ---------------------------------------------------------------- nullable path
// Arrow restarts its 16-value leaves at every run of valid rows.  All bookkeeping is per 1024-ROW SEGMENT (16 words of the
// validity bitmap = the 64 sixteen-row windows one wave owns); what used to be per 16-row window in global arrays (12 B per
// window, two device scans over n / 16 elements, workgroup barriers in the emit kernel: 4.7 ms per 1e9 rows) now stays inside a wave:
//   k_null_seg_state : one wave per 4096 rows (lane = bitmap word, 16 lanes = one segment): rows of the leaf still OPEN at every
//                      segment's end if the segment holds an invalid row, else "pass" (a full segment hands its input on:
//                      1024 % 16 == 0); also the valid-row count (one atomic per wave, not per tile)
//   "latest" device scan over segments -> rows already in the open leaf at every segment START
//   k_null_seg_count : same shape: leaves FINISHED inside every segment (+ the leaf still open at the end of the array) -> sum scan
//   k_null_seg_emit  : one WAVE per segment, lane = 16-row window, no workgroup barrier: values staged through the wave's own LDS
//                      region (coalesced loads), open-leaf position and first leaf index of every window from two wave scans
//                      seeded with the segment's carry-ins, ONE walk over the window's bits that sums and counts its leaves,
//                      leaf sums written straight to their places in the compact leaf array
// then the ordinary merge tree (run_tree) over the leaf array.  Traffic: the values once + ~3 x the bitmap + the leaf array.
__device__ __forceinline__ unsigned window_bits(const uint8_t* valid, int64_t off, int64_t n, int64_t w) {
  // 16 validity bits of window w (rows [16w, 16w+16)), zero beyond n
  const int64_t row = w << 4;
  if (row >= n) return 0u;
  const int64_t bit0 = off + row;
  const int64_t byte0 = bit0 >> 3;
  const int sh = (int)(bit0 & 7);
  const int64_t last_byte = (off + n + 7) >> 3;
  unsigned v = (unsigned)valid[byte0];
  if (byte0 + 1 < last_byte) v |= (unsigned)valid[byte0 + 1] << 8;
  if (sh && byte0 + 2 < last_byte) v |= (unsigned)valid[byte0 + 2] << 16;
  v = (v >> sh) & 0xFFFFu;
  const int64_t remain = n - row;
  if (remain < 16) v &= (1u << remain) - 1u;
  return v;
}
// rows of the open leaf after a 16-row window with bits m, entered with p rows open; *fin += leaves finished inside
__device__ __forceinline__ int window_walk(unsigned m, int p, int* fin) {
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    if ((m >> q) & 1u) {
      if (++p == 16) { ++*fin; p = 0; }
    } else if (p > 0) { ++*fin; p = 0; }
  }
  return p;
}
constexpr int kSegRows = 1024;       // rows per segment = 64 windows of 16 = one wave of the emit kernel
constexpr int kNullTileWaves = 4;    // 4096-row tiles (one wave each, lane = bitmap word) per workgroup of the two bitmap-only kernels
// lane l holds the tile's validity word l (rows [64 l, 64 l + 64) of the tile, zero beyond n)
__device__ __forceinline__ uint64_t tile_word(const uint8_t* valid, int64_t off, int64_t n, int64_t tile, int lane) {
  const int64_t row = tile * kLeafElems + (int64_t)lane * 64;
  if (row >= n) return 0ull;
  if (((off + row) & 63) == 0 && row + 64 <= n && (reinterpret_cast<uintptr_t>(valid) & 7) == 0)
    return reinterpret_cast<const uint64_t*>(valid)[(off + row) >> 6];  // the common case: one aligned 8-byte load instead of nine bytes
  return load_bits64(valid, off + row, off + n);
}
// leaves finished inside a 64-row word entered with p rows open, run by run (a 5 %-null word has ~4 runs; bit by bit it was 64 steps)
__device__ __forceinline__ int word_leaf_count(uint64_t w, int p) {
  int fin = 0, pos = 0;
  while (pos < 64) {
    const uint64_t x = w >> pos;
    const uint64_t nx = ~x;
    int ones = nx ? __builtin_ctzll(nx) : 64;
    ones = ones < 64 - pos ? ones : 64 - pos;
    if (ones > 0) {
      const int tot = p + ones;
      fin += tot >> 4;
      p = tot & 15;
      pos += ones;
    }
    if (pos >= 64) break;
    if (p > 0) {  // an invalid row closes the open leaf
      ++fin;
      p = 0;
    }
    const uint64_t y = w >> pos;
    const int zeros = y ? __builtin_ctzll(y) : 64 - pos;
    pos += zeros;
  }
  return fin | (p << 16);
}
__global__ void __launch_bounds__(kNullTileWaves * 64) k_null_seg_state(const uint8_t* __restrict__ valid, int64_t off, int64_t n, int64_t ntiles,
                                                                        int32_t* __restrict__ z /* [4 * ntiles] */,
                                                                        unsigned long long* __restrict__ valid_total) {
  const int lane = threadIdx.x & 63;
  unsigned long long vc = 0;
  // (grid-stride over tiles and ONE atomic per wave at the end: an atomic per tile on the single counter serialised the kernel --
  //  2.9 ms for the 244 k tiles of 1e9 rows)
  for (int64_t tile = (int64_t)blockIdx.x * kNullTileWaves + (threadIdx.x >> 6); tile < ntiles; tile += (int64_t)gridDim.x * kNullTileWaves) {
    const uint64_t w = tile_word(valid, off, n, tile, lane);
    const uint64_t nonfull = __ballot(w != ~0ull);
    vc += (unsigned long long)__popcll(w);
    // per segment (16 lanes): the valid rows behind the segment's last invalid row, modulo 16 (whole words behind it add 64 each)
    const int g = lane >> 4;
    const unsigned nf = (unsigned)(nonfull >> (16 * g)) & 0xFFFFu;
    const int hl = nf ? 16 * g + (31 - __builtin_clz(nf)) : -1;
    const uint64_t wl = __shfl(w, hl < 0 ? 0 : hl, 64);
    if ((lane & 15) == 0) z[tile * 4 + g] = hl < 0 ? -1 : (int32_t)(__builtin_clzll(~wl) & 15);
  }
  for (int d = 32; d > 0; d >>= 1) vc += __shfl_down(vc, d, 64);
  if (lane == 0 && vc) atomicAdd(valid_total, vc);
}
__global__ void __launch_bounds__(kNullTileWaves * 64) k_null_seg_count(const uint8_t* __restrict__ valid, int64_t off, int64_t n, int64_t ntiles,
                                                                        const int32_t* __restrict__ pos_in /* scanned z */,
                                                                        int64_t* __restrict__ count /* [4 * ntiles] */) {
  const int lane = threadIdx.x & 63;
  const int64_t tile = (int64_t)blockIdx.x * kNullTileWaves + (threadIdx.x >> 6);
  if (tile >= ntiles) return;
  const uint64_t w = tile_word(valid, off, n, tile, lane);
  // open-leaf rows at the start of every word: "latest" scan over the words of the tile, seeded with the tile's carry-in
  const int zw = w == ~0ull ? -1 : (int)(__builtin_clzll(~w) & 15);
  const int inc = wave_inclusive_scan(zw, LatestOp());
  const int exc = __shfl_up(inc, 1, 64);
  const int carry = pos_in[tile * 4] < 0 ? 0 : pos_in[tile * 4];
  int p = (lane == 0 || exc < 0) ? carry : exc;  // exc < 0: every earlier word of the tile is full (64 % 16 == 0: the carry passes through)
  const int fp = word_leaf_count(w, p);
  int fin = fp & 0xFFFF;
  p = fp >> 16;
  // the leaf still open at the end of the ARRAY: the last word that holds a row
  const int64_t row0 = tile * kLeafElems + (int64_t)lane * 64;
  if (row0 >= n) fin = 0;  // (a word past the end of the array: its zero bits are not rows)
  else if (row0 + 64 >= n && p > 0) ++fin;
  for (int d = 8; d > 0; d >>= 1) fin += __shfl_down(fin, d, 16);
  if ((lane & 15) == 0) count[tile * 4 + (lane >> 4)] = fin;
}
constexpr int kEmitWaves = 2;  // waves per workgroup of the emit kernel (waves are independent: the size only packs the LDS)
// Persistent waves: a wave takes segments seg, seg + stride, ... and requests the NEXT segment's rows (and its small inputs) before
// it works on the current one, so the ~2 us of memory latency per segment overlap with the staging / scan / walk of the previous
// one (one segment per short-lived wave: 2.8 ms per 1e9 rows, mostly waiting).
template <typename T>
__global__ void __launch_bounds__(kEmitWaves * 64) k_null_seg_emit(const T* __restrict__ v, const uint8_t* __restrict__ valid, int64_t off, int64_t n,
                                                                   const int32_t* __restrict__ seg_pos, const int64_t* __restrict__ seg_leaf_base,
                                                                   double* __restrict__ leaves) {
  __shared__ double lds_all[kEmitWaves][64 * kLeafPad];
  const int lane = threadIdx.x & 63;
  double* const lds = lds_all[threadIdx.x >> 6];
  const int64_t nsegs = (n + kSegRows - 1) / kSegRows, nwin = (n + 15) >> 4;
  const int64_t stride = (int64_t)gridDim.x * kEmitWaves;
  int64_t seg = (int64_t)blockIdx.x * kEmitWaves + (threadIdx.x >> 6);
  if (seg >= nsegs) return;  // (whole waves leave: nothing below is a workgroup barrier)
  T nv[16];
  T npt = T(0);
  unsigned nm = 0;
  int ncarry = 0;
  int64_t nleaf0 = 0;
  auto request = [&](int64_t sg) {
    const int64_t base = sg * kSegRows;
    const int rows = (int)((n - base) < kSegRows ? (n - base) : kSegRows);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int idx = k * 64 + lane;  // coalesced 8-byte lanes
      nv[k] = idx < rows ? v[base + idx] : T(0);
    }
    const int64_t w = sg * 64 + lane;
    nm = w < nwin ? window_bits(valid, off, n, w) : 0u;
    ncarry = seg_pos[sg];
    nleaf0 = seg_leaf_base[sg];
    npt = (lane < 16 && base >= 16) ? v[base - 16 + lane] : T(0);  // the 16 rows in front of the segment (lane 0's open leaf)
  };
  request(seg);
  for (; seg < nsegs; seg += stride) {
    const int64_t w = seg * 64 + lane;
    const bool in = w < nwin;
    const unsigned m = nm;
    const int carry_in = ncarry;
    const int64_t leaf0 = nleaf0;
    const double pt = to_f64(npt);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int idx = k * 64 + lane;
      lds[idx + (idx >> 4)] = to_f64(nv[k]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    double x[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) x[q] = lds[lane * kLeafPad + q];
    if (seg + stride < nsegs) request(seg + stride);  // in flight while this segment is reduced
    // open-leaf rows at every window start: "latest" wave scan of the windows' end states, seeded with the segment's carry-in
    const int zw = m == 0xFFFFu ? -1 : (int)__builtin_clz(~(m << 16));
    const int inc = wave_inclusive_scan(zw, LatestOp());
    const int exc = __shfl_up(inc, 1, 64);
    int p = (lane == 0 || exc < 0) ? (carry_in < 0 ? 0 : carry_in) : exc;
    // Partial sum of the leaf that is open at the window start = the previous window's sequential sum over its LAST p rows.  That
    // leaf always STARTS inside the previous window (a window of 16 rows closes at least one leaf unless it ends exactly on one), so
    // every lane can sum its own tail from its own registers -- no dependency on its incoming leaf -- and hand it to the next lane
    // with one shuffle (sixteen 8-byte shuffles per window before: ~64 LDS-crossbar instructions per segment).
    const int pend = m == 0xFFFFu ? p : zw;  // rows of the leaf open at MY window's end (all valid: what came in comes out)
    double tail = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (q >= 16 - pend) tail = pw_leaf_add(tail, x[q]);
    double acc = __shfl_up(tail, 1, 64);
    {  // lane 0: the rows in front of the segment sit one per lane in pt (lanes 0..15): scalar reads, sequential adds
      double a0 = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int lo = __builtin_amdgcn_readlane((int)(__double_as_longlong(pt) & 0xFFFFFFFFll), q);
        const int hi = __builtin_amdgcn_readlane((int)(__double_as_longlong(pt) >> 32), q);
        const double t0 = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
        if (q >= 16 - p) a0 = pw_leaf_add(a0, t0);
      }
      if (lane == 0) acc = a0;
    }
    if (!in || p == 0) acc = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // every lane holds its values: its LDS row now collects its leaf sums
    int j = 0;
    if (in) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if ((m >> q) & 1u) {
          acc = pw_leaf_add(p == 0 ? 0.0 : acc, x[q]);
          if (++p == 16) { lds[lane * kLeafPad + j++] = acc; p = 0; }
        } else if (p > 0) { lds[lane * kLeafPad + j++] = acc; p = 0; }
      }
      if (w == nwin - 1 && p > 0) lds[lane * kLeafPad + j++] = acc;
    }
    const int incj = wave_inclusive_scan(j, SumOp());
    double* const dst = leaves + leaf0 + (incj - j);
    for (int t = 0; t < j; ++t) dst[t] = lds[lane * kLeafPad + t];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the row is read out before the next segment is staged
  }
}

template <typename T>
static int sum_nullable(const T* v, const uint8_t* valid, int64_t off, int64_t n, double* result_dev, unsigned long long* valid_total,
                        Scratch& s, hipStream_t st) {
  const int64_t ntiles = ceil_div(n, kLeafElems), nseg = ntiles * 4;
  int32_t* z = s.get<int32_t>((size_t)nseg);
  int64_t* lc = s.get<int64_t>((size_t)nseg);
  int64_t* total = s.get<int64_t>(1);
  PDX_SCRATCH_CHECK(s);
  const unsigned grid = (unsigned)ceil_div(ntiles, kNullTileWaves);
  hipLaunchKernelGGL(k_null_seg_state, dim3(std::min<unsigned>(grid, kCUs * 8)), dim3(kNullTileWaves * 64), 0, st, valid, off, n, ntiles, z, valid_total);
  PDX_TRY((device_exclusive_scan<int32_t, LatestOp>(z, z, nseg, nullptr, s, st)));
  hipLaunchKernelGGL(k_null_seg_count, dim3(grid), dim3(kNullTileWaves * 64), 0, st, valid, off, n, ntiles, z, lc);
  PDX_TRY((device_exclusive_scan<int64_t, SumOp>(lc, lc, nseg, total, s, st)));
  int64_t m = 0;
  PDX_TRY(read_back(&m, total, sizeof(m), st));
  double* leaves = s.get<double>((size_t)(m ? m : 1));
  PDX_SCRATCH_CHECK(s);
  // persistent waves: launch exactly what is resident at once (146 VGPRs -> 3 waves per SIMD = 6 two-wave workgroups per CU; a grid sized
  // by the LDS alone -- 9 per CU -- left a third of the waves to run as a half-empty second round).  PDX_NULLSUM_WGS_PER_CU: diagnostic.
  static const int wgs_per_cu = [] { const char* e = getenv("PDX_NULLSUM_WGS_PER_CU"); return e && atoi(e) > 0 ? atoi(e) : 6; }();
  const unsigned egrid = (unsigned)std::min<int64_t>(ceil_div(ceil_div(n, kSegRows), kEmitWaves), (int64_t)kCUs * wgs_per_cu);
  hipLaunchKernelGGL((k_null_seg_emit<T>), dim3(egrid), dim3(kEmitWaves * 64), 0, st, v, valid, off, n, z, lc, leaves);
  PDX_LAUNCH_CHECK();
  return run_tree(leaves, m, nullptr, 0, nullptr, 0, result_dev, s, st);
}

// ---------------------------------------------------------------- count / integer sum / min-max
__global__ void k_count_valid(const uint8_t* __restrict__ valid, int64_t off, int64_t n, unsigned long long* __restrict__ total) {
  int64_t nwords = (n + 63) >> 6;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long vc = 0;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += stride)
    vc += __popcll(load_bits64(valid, off + (w << 6), off + n));
  for (int d = 32; d > 0; d >>= 1) vc += __shfl_down(vc, d, 64);
  if ((threadIdx.x & 63) == 0 && vc) atomicAdd(total, vc);
}

__global__ void __launch_bounds__(256) k_sum_i64(const int64_t* __restrict__ v, const uint8_t* __restrict__ valid, int64_t off,
                                                 int64_t n, unsigned long long* __restrict__ total) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    if (!valid || bit_get(valid, off + i)) acc += (unsigned long long)v[i];
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(total, acc);  // wrap-around add is order independent
}

template <typename T, bool ML>
__device__ __forceinline__ void block_reduce_extreme(Extreme<T, ML>& e, MinMaxPartial<T>* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int d = 32; d > 0; d >>= 1) {
    T omin = __shfl_down(e.vmin, d, 64), omax = __shfl_down(e.vmax, d, 64);
    long long ormin = __shfl_down(e.rmin, d, 64), ormax = __shfl_down(e.rmax, d, 64);
    e.merge(omin, ormin, omax, ormax);
  }
  if (lane == 0) smem[wave] = {e.vmin, e.vmax, e.rmin, e.rmax};
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) e.merge(smem[w].vmin, smem[w].rmin, smem[w].vmax, smem[w].rmax);
  }
}

// pass 1: grid-stride over rows -> one partial per block.  NaN rows are skipped (counted separately through `valid` only).
template <typename T, bool ML>
__global__ void __launch_bounds__(256) k_minmax_partial(const T* __restrict__ v, const uint8_t* __restrict__ valid, int64_t off,
                                                        int64_t n, MinMaxPartial<T>* __restrict__ partials) {
  __shared__ MinMaxPartial<T> smem[4];
  Extreme<T, ML> e;
  e.init();
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // four independent 8-byte streams per thread (the (value, row) reduction is order independent)
  for (; i + 3 * stride < n; i += 4 * stride) {
    T x[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      x[k] = v[i + k * stride];
      ok[k] = !valid || bit_get(valid, off + i + k * stride);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (ok[k] && x[k] == x[k]) e.add(x[k], i + k * stride);  // (NaN skipped; never true for integers)
  }
  for (; i < n; i += stride) {
    if (valid && !bit_get(valid, off + i)) continue;
    T x = v[i];
    if (x != x) continue;
    e.add(x, i);
  }
  block_reduce_extreme(e, smem);
  if (threadIdx.x == 0) partials[blockIdx.x] = {e.vmin, e.vmax, e.rmin, e.rmax};
}
template <typename T, bool ML>
__global__ void __launch_bounds__(256) k_minmax_final(const MinMaxPartial<T>* __restrict__ partials, int np,
                                                      MinMaxPartial<T>* __restrict__ out) {
  __shared__ MinMaxPartial<T> smem[4];
  Extreme<T, ML> e;
  e.init();
  for (int i = threadIdx.x; i < np; i += blockDim.x) e.merge(partials[i].vmin, partials[i].rmin, partials[i].vmax, partials[i].rmax);
  block_reduce_extreme(e, smem);
  if (threadIdx.x == 0) *out = {e.vmin, e.vmax, e.rmin, e.rmax};
}

// max_last: the array holds at least one null, so the LAST of tied maxima (0.0 / -0.0) wins (minmax.hpp)
template <typename T>
static int minmax_impl(const T* v, const uint8_t* valid, int64_t off, int64_t n, MinMaxPartial<T>* host_out, Scratch& s, hipStream_t st,
                       bool max_last = false) {
  int grid = grid_for(n, 256, 8);
  MinMaxPartial<T>* partials = s.get<MinMaxPartial<T>>((size_t)grid + 1);
  PDX_SCRATCH_CHECK(s);
  // the final kernel writes its one record into this thread's pinned slot when it fits
  MinMaxPartial<T>* pin = sizeof(MinMaxPartial<T>) <= 64 ? static_cast<MinMaxPartial<T>*>(pinned_slot()) : nullptr;
  MinMaxPartial<T>* fin = pin ? pin : partials + grid;
  if (max_last) {
    hipLaunchKernelGGL((k_minmax_partial<T, true>), dim3(grid), dim3(256), 0, st, v, valid, off, n, partials);
    hipLaunchKernelGGL((k_minmax_final<T, true>), dim3(1), dim3(256), 0, st, partials, grid, fin);
  } else {
    hipLaunchKernelGGL((k_minmax_partial<T, false>), dim3(grid), dim3(256), 0, st, v, valid, off, n, partials);
    hipLaunchKernelGGL((k_minmax_final<T, false>), dim3(1), dim3(256), 0, st, partials, grid, fin);
  }
  PDX_LAUNCH_CHECK();
  if (pin) {
    PDX_HIP(hipStreamSynchronize(st));
    memcpy(host_out, pin, sizeof(*host_out));
  } else {
    PDX_HIP(hipMemcpyAsync(host_out, partials + grid, sizeof(*host_out), hipMemcpyDeviceToHost, st));
    PDX_HIP(hipStreamSynchronize(st));
  }
  return PDX_OK;
}

int minmax_i64_host(const long long* v, int64_t n, long long* mn, long long* mx, Scratch& s, hipStream_t st) {
  MinMaxPartial<long long> r;
  PDX_TRY(minmax_impl<long long>(v, nullptr, 0, n, &r, s, st));
  *mn = r.vmin;
  *mx = r.vmax;
  return PDX_OK;
}

int minmax_keys_host(const long long* v, const uint8_t* valid, int64_t off, int64_t n, MinMaxPartial<long long>* out, Scratch& s,
                     hipStream_t st) {
  PDX_PROFILE("key_minmax", st);
  return minmax_impl<long long>(v, valid, off, n, out, s, st);
}

// count of valid rows (host result)
int count_valid_host(const pdx_column* a, int64_t* out, Scratch& s, hipStream_t st) {
  const uint8_t* valid = validity_or_null(a);
  if (!valid || a->length == 0) {
    *out = a->length;
    return PDX_OK;
  }
  unsigned long long* total = s.get<unsigned long long>(1);
  PDX_SCRATCH_CHECK(s);
  PDX_HIP(hipMemsetAsync(total, 0, sizeof(*total), st));
  int64_t nwords = (a->length + 63) >> 6;
  hipLaunchKernelGGL(k_count_valid, dim3(grid_for(nwords, 256)), dim3(256), 0, st, valid, a->offset, a->length, total);
  PDX_LAUNCH_CHECK();
  unsigned long long h = 0;
  PDX_TRY(read_back(&h, total, sizeof(h), st));
  *out = (int64_t)h;
  return PDX_OK;
}

// fp64 pairwise sum of a column (double or int64 values converted to double); host result + valid count
template <typename T>
static int pairwise_sum_host(const pdx_column* a, double* sum_out, int64_t* count_out, Scratch& s, hipStream_t st) {
  const T* v = static_cast<const T*>(a->values) + a->offset;
  const uint8_t* valid = validity_or_null(a);
  int64_t n = a->length;
  if (n == 0) {
    *sum_out = 0.0;
    *count_out = 0;
    return PDX_OK;
  }
  // the finish kernel writes the scalar straight into this thread's pinned slot: no copy command between the kernel and the host
  double* pinned = static_cast<double*>(pinned_slot());
  double* res = pinned ? pinned : s.get<double>(1);
  PDX_SCRATCH_CHECK(s);
  if (!valid) {
    PDX_TRY(sum_dense<T>(v, n, res, s, st));
    *count_out = n;
  } else {
    unsigned long long* vt = s.get<unsigned long long>(1);
    PDX_SCRATCH_CHECK(s);
    PDX_HIP(hipMemsetAsync(vt, 0, sizeof(*vt), st));
    PDX_TRY(sum_nullable<T>(v, valid, a->offset, n, res, vt, s, st));
    unsigned long long h = 0;
    if (pinned) {
      PDX_HIP(hipMemcpyAsync(pinned + 1, vt, sizeof(h), hipMemcpyDeviceToHost, st));
    } else {
      PDX_HIP(hipMemcpyAsync(&h, vt, sizeof(h), hipMemcpyDeviceToHost, st));
    }
    PDX_HIP(hipStreamSynchronize(st));
    if (pinned) h = *reinterpret_cast<volatile unsigned long long*>(pinned + 1);
    *count_out = (int64_t)h;
    if (pinned) {
      *sum_out = *reinterpret_cast<volatile double*>(pinned);
      return PDX_OK;
    }
  }
  if (pinned) {
    PDX_HIP(hipStreamSynchronize(st));
    *sum_out = *reinterpret_cast<volatile double*>(pinned);
  } else {
    PDX_HIP(hipMemcpyAsync(sum_out, res, sizeof(double), hipMemcpyDeviceToHost, st));
    PDX_HIP(hipStreamSynchronize(st));
  }
  return PDX_OK;
}

}  // namespace pdx

using namespace pdx;

extern "C" int pdx_aggregate(int kind, const pdx_column* a, pdx_scalar* out, void* stream) {
  PDX_TRY(check_column(a, "pdx_aggregate"));
  if (!out) return fail(PDX_INVALID, "pdx_aggregate: null output");
  if (kind < PDX_AGG_SUM || kind > PDX_AGG_COUNT) return fail(PDX_INVALID, "pdx_aggregate: unknown kind");
  const bool is_f = a->dtype == PDX_FLOAT64;
  if (!is_f && a->dtype != PDX_INT64 && !(kind == PDX_AGG_COUNT) && !((kind == PDX_AGG_MIN || kind == PDX_AGG_MAX) && a->dtype == PDX_TIMESTAMP_NS))
    return fail(PDX_NOT_IMPLEMENTED, "pdx_aggregate: only int64/float64 columns are supported");
  hipStream_t st = as_stream(stream);
  Scratch s;
  out->is_valid = 0;
  out->count = 0;
  out->v.i64 = 0;
  int64_t n = a->length;
  if (kind == PDX_AGG_COUNT) {
    int64_t c = 0;
    PDX_TRY(count_valid_host(a, &c, s, st));
    out->dtype = PDX_INT64;
    out->is_valid = 1;
    out->v.i64 = c;
    out->count = c;
    return PDX_OK;
  }
  if (kind == PDX_AGG_SUM || kind == PDX_AGG_MEAN) {
    if (is_f || kind == PDX_AGG_MEAN) {
      double sum = 0;
      int64_t cnt = 0;
      if (is_f) PDX_TRY(pairwise_sum_host<double>(a, &sum, &cnt, s, st));
      else PDX_TRY(pairwise_sum_host<int64_t>(a, &sum, &cnt, s, st));
      out->dtype = PDX_FLOAT64;
      out->count = cnt;
      out->is_valid = cnt > 0;  // ScalarAggregateOptions::min_count = 1
      out->v.f64 = cnt ? (kind == PDX_AGG_MEAN ? (sum == sum ? sum / (double)cnt : sum) : sum) : 0.0;  // (a NaN sum is already quiet: NaN / n keeps its bits)
      return PDX_OK;
    }
    // integer sum: wrap-around, int64 -> int64
    int64_t cnt = 0;
    PDX_TRY(count_valid_host(a, &cnt, s, st));
    unsigned long long* total = s.get<unsigned long long>(1);
    PDX_SCRATCH_CHECK(s);
    PDX_HIP(hipMemsetAsync(total, 0, sizeof(*total), st));
    if (n)
      hipLaunchKernelGGL(k_sum_i64, dim3(grid_for(n, 256, 8)), dim3(256), 0, st, static_cast<const int64_t*>(a->values) + a->offset,
                         validity_or_null(a), a->offset, n, total);
    PDX_LAUNCH_CHECK();
    unsigned long long h = 0;
    PDX_TRY(read_back(&h, total, sizeof(h), st));
    out->dtype = PDX_INT64;
    out->count = cnt;
    out->is_valid = cnt > 0;
    out->v.i64 = (int64_t)h;
    return PDX_OK;
  }
  // min / max
  int64_t cnt = 0;
  PDX_TRY(count_valid_host(a, &cnt, s, st));
  out->count = cnt;
  out->dtype = a->dtype;
  if (cnt == 0) return PDX_OK;  // null
  if (is_f) {
    MinMaxPartial<double> r;
    PDX_TRY(minmax_impl<double>(static_cast<const double*>(a->values) + a->offset, validity_or_null(a), a->offset, n, &r, s, st,
                                /*max_last=*/cnt < n));
    out->is_valid = 1;
    if (r.rmin < 0) out->v.f64 = __builtin_nan("");  // every valid value is NaN
    else out->v.f64 = kind == PDX_AGG_MIN ? r.vmin : r.vmax;
  } else {
    MinMaxPartial<long long> r;
    PDX_TRY(minmax_impl<long long>(static_cast<const long long*>(a->values) + a->offset, validity_or_null(a), a->offset, n, &r, s, st));
    out->is_valid = 1;
    out->v.i64 = kind == PDX_AGG_MIN ? r.vmin : r.vmax;
  }
  return PDX_OK;
}
