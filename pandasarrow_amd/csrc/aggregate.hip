// aggregate.hip -- whole-array sum / mean / min / max / count for gfx950.
//
// Replaces CallFunction("sum"|"mean"|"min"|"max"|"count") reached from NDFrame::sum/mean/min/max/count
// (reference src/ndframe.cpp:26-31 macro, 119, 162-166, 220) and MinMax (src/resample.cpp:223).
// HBM-bound single pass: 8 B/row (+1/8 B validity).  fp64 sum/mean reproduce Arrow's pairwise tree exactly
// (see pairwise.hpp); integer sum wraps; integer mean sums the values converted to double with the same tree
// (Arrow 25.0.0 behaviour, pinned by tests/golden agg_i64_*); min/max skip NaN unless all values are NaN and keep
// the FIRST of tied values (0.0 vs -0.0), implemented as an order-independent (value, row) reduction.
#include "minmax.hpp"
#include "pairwise.hpp"
#include "scan.hpp"

namespace pdx {

constexpr int kLeafBlock = 256;                 // threads = leaves per block
constexpr int kLeafElems = kLeafBlock * 16;     // 4096 values per block
constexpr int kLeafPad = 17;                    // LDS stride per leaf (doubles): conflict-free ds_read_b64

template <typename T>
__device__ __forceinline__ double to_f64(T x) { return (double)x; }

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
// single thread: replay the counter over the ragged tails, top level first, then fold
__global__ void k_sum_finish(FinishArgs a, double* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  PairwiseCounter c;
  c.init();
  bool any = false;
  for (int r = a.nlevels - 1; r >= 0; --r)
    for (int i = 0; i < a.counts[r]; ++i) {
      c.push(a.tails[r][i], 8 * r);
      any = true;
    }
  *out = any ? c.finish() : 0.0;
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
// Arrow restarts its 16-value leaves at every run of valid rows.  One thread owns a 16-row window (= one leaf length):
//   k_null_window_state : rows at the END of the window that stay in an open leaf if the window has an invalid row, else "pass"
//   "latest" device scan: rows already in the open leaf at every window START (a full window passes the value through: 16 | 16)
//   k_null_window_count : leaves FINISHED inside the window (+ the leaf still open at the end of the array)  -> device sum scan
//   k_null_window_emit  : 4096 rows per block staged through LDS (coalesced loads); the open leaf's partial sum at a window
//                         start is the sequential sum of the previous window's last rows; every window walks its 16 bits once
//                         and writes the leaves it finishes, in order, into the compact leaf array
// then the ordinary merge tree (run_tree) over the leaf array.
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
__global__ void k_null_window_state(const uint8_t* __restrict__ valid, int64_t off, int64_t n, int64_t nwin, int32_t* __restrict__ z,
                                    unsigned long long* __restrict__ valid_total) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long vc = 0;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwin; w += stride) {
    const unsigned m = window_bits(valid, off, n, w);
    z[w] = m == 0xFFFFu ? -1 : (int32_t)__builtin_clz(~(m << 16));  // valid rows at the end of the window
    vc += __popc(m);
  }
  for (int d = 32; d > 0; d >>= 1) vc += __shfl_down(vc, d, 64);
  if ((threadIdx.x & 63) == 0 && vc) atomicAdd(valid_total, vc);
}
__global__ void k_null_window_count(const uint8_t* __restrict__ valid, int64_t off, int64_t n, int64_t nwin, const int32_t* __restrict__ pos_in,
                                    int64_t* __restrict__ count) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwin; w += stride) {
    const unsigned m = window_bits(valid, off, n, w);
    int p = pos_in[w] < 0 ? 0 : pos_in[w];
    int fin = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if ((m >> q) & 1u) {
        if (++p == 16) { ++fin; p = 0; }
      } else if (p > 0) { ++fin; p = 0; }
    }
    if (w == nwin - 1 && p > 0) ++fin;  // the leaf still open at the end of the array
    count[w] = fin;
  }
}
// (smaller workgroups than the dense kernels: two barriers and a serial 16-row loop per workgroup -- more, smaller ones overlap better)
constexpr int kEmitBlock = 128;
constexpr int kEmitElems = kEmitBlock * 16;
template <typename T>
__global__ void __launch_bounds__(kEmitBlock) k_null_window_emit(const T* __restrict__ v, const uint8_t* __restrict__ valid, int64_t off, int64_t n,
                                                                 int64_t nwin, const int32_t* __restrict__ pos_in,
                                                                 const int64_t* __restrict__ leaf_base, double* __restrict__ leaves) {
  __shared__ double lds[kEmitBlock * kLeafPad];
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * kEmitElems;
  const int rows = (int)((n - base) < kEmitElems ? (n - base) : kEmitElems);
  // the window's own small inputs (validity bits, open-leaf position, first leaf index) are requested together with the values:
  // asked for only after the barrier they cost a second and third memory round trip per block (SQ_WAIT_ANY was 86 % of wave cycles)
  const int64_t w = (int64_t)blockIdx.x * kEmitBlock + tid;
  const bool in = w < nwin;
  const unsigned m = in ? window_bits(valid, off, n, w) : 0u;
  const int p_in = in ? pos_in[w] : 0;
  const int64_t li_in = in ? leaf_base[w] : 0;
  __shared__ double prev_tail[16];  // first window of the block: the rows of an open leaf live in the previous block
  if (tid < 16) prev_tail[tid] = base >= 16 ? to_f64(v[base - 16 + tid]) : 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    int idx = k * kEmitBlock + tid;
    if (idx < rows) lds[idx + (idx >> 4)] = to_f64(v[base + idx]);
  }
  __syncthreads();
  // Every thread takes its window's 16 values (and the tail of the previous window, if a leaf is open) into registers; after a
  // barrier the staging area is reused for the block's finished leaf sums, which land at consecutive positions of `leaves`
  // (leaf_base is an exclusive scan over windows) and leave in one coalesced copy -- a store per finished leaf straight from the
  // row loop costs up to 16 mostly empty store instructions per wave (4.3 ms per 1e9 rows).
  __shared__ int64_t blk_lo_s, blk_hi_s;
  int p = p_in < 0 ? 0 : p_in;
  double x[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) x[q] = lds[tid * kLeafPad + q];
  // partial sum of the leaf that is open at the window start: its p rows are the LAST p rows of the previous window (all valid)
  double acc = 0.0;
  if (in && p > 0) {
    if (tid > 0) {
      for (int q = 16 - p; q < 16; ++q) acc += lds[(tid - 1) * kLeafPad + q];
    } else {
      for (int q = 16 - p; q < 16; ++q) acc += prev_tail[q];
    }
  }
  int64_t li = li_in;
  if (tid == 0) blk_lo_s = li;
  __syncthreads();
  const int64_t blk_lo = blk_lo_s;
  if (in) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if ((m >> q) & 1u) {
        acc = (p == 0 ? 0.0 : acc) + x[q];
        if (++p == 16) { lds[li++ - blk_lo] = acc; p = 0; }
      } else if (p > 0) { lds[li++ - blk_lo] = acc; p = 0; }
    }
    if (w == nwin - 1 && p > 0) lds[li++ - blk_lo] = acc;
    if (tid == kEmitBlock - 1 || w == nwin - 1) blk_hi_s = li;  // the block's last window
  }
  __syncthreads();
  const int cnt = (int)(blk_hi_s - blk_lo);
  for (int i = tid; i < cnt; i += kEmitBlock) leaves[blk_lo + i] = lds[i];
}

template <typename T>
static int sum_nullable(const T* v, const uint8_t* valid, int64_t off, int64_t n, double* result_dev, unsigned long long* valid_total,
                        Scratch& s, hipStream_t st) {
  const int64_t nwin = (n + 15) >> 4;
  int32_t* z = s.get<int32_t>((size_t)nwin);
  int64_t* lc = s.get<int64_t>((size_t)nwin);
  int64_t* total = s.get<int64_t>(1);
  PDX_SCRATCH_CHECK(s);
  int grid = grid_for(nwin, 256, 4);
  hipLaunchKernelGGL(k_null_window_state, dim3(grid), dim3(256), 0, st, valid, off, n, nwin, z, valid_total);
  PDX_TRY((device_exclusive_scan<int32_t, LatestOp>(z, z, nwin, nullptr, s, st)));
  hipLaunchKernelGGL(k_null_window_count, dim3(grid), dim3(256), 0, st, valid, off, n, nwin, z, lc);
  PDX_TRY((device_exclusive_scan<int64_t, SumOp>(lc, lc, nwin, total, s, st)));
  int64_t m = 0;
  PDX_HIP(hipMemcpyAsync(&m, total, sizeof(m), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
  double* leaves = s.get<double>((size_t)(m ? m : 1));
  PDX_SCRATCH_CHECK(s);
  hipLaunchKernelGGL((k_null_window_emit<T>), dim3((unsigned)ceil_div(n, kEmitElems)), dim3(kEmitBlock), 0, st, v, valid, off, n, nwin, z, lc, leaves);
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
  if (max_last) {
    hipLaunchKernelGGL((k_minmax_partial<T, true>), dim3(grid), dim3(256), 0, st, v, valid, off, n, partials);
    hipLaunchKernelGGL((k_minmax_final<T, true>), dim3(1), dim3(256), 0, st, partials, grid, partials + grid);
  } else {
    hipLaunchKernelGGL((k_minmax_partial<T, false>), dim3(grid), dim3(256), 0, st, v, valid, off, n, partials);
    hipLaunchKernelGGL((k_minmax_final<T, false>), dim3(1), dim3(256), 0, st, partials, grid, partials + grid);
  }
  PDX_LAUNCH_CHECK();
  PDX_HIP(hipMemcpyAsync(host_out, partials + grid, sizeof(*host_out), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
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
  PDX_HIP(hipMemcpyAsync(&h, total, sizeof(h), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
  *out = (int64_t)h;
  return PDX_OK;
}

// fp64 pairwise sum of a column (double or int64 values converted to double); host result + valid count
template <typename T>
static int pairwise_sum_host(const pdx_column* a, double* sum_out, int64_t* count_out, Scratch& s, hipStream_t st) {
  const T* v = static_cast<const T*>(a->values) + a->offset;
  const uint8_t* valid = validity_or_null(a);
  int64_t n = a->length;
  double* res = s.get<double>(1);
  unsigned long long* vt = s.get<unsigned long long>(1);
  PDX_SCRATCH_CHECK(s);
  PDX_HIP(hipMemsetAsync(vt, 0, sizeof(*vt), st));
  if (n == 0) {
    *sum_out = 0.0;
    *count_out = 0;
    return PDX_OK;
  }
  if (!valid) {
    PDX_TRY(sum_dense<T>(v, n, res, s, st));
    *count_out = n;
  } else {
    PDX_TRY(sum_nullable<T>(v, valid, a->offset, n, res, vt, s, st));
    unsigned long long h = 0;
    PDX_HIP(hipMemcpyAsync(&h, vt, sizeof(h), hipMemcpyDeviceToHost, st));
    PDX_HIP(hipStreamSynchronize(st));
    *count_out = (int64_t)h;
  }
  PDX_HIP(hipMemcpyAsync(sum_out, res, sizeof(double), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
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
      out->v.f64 = cnt ? (kind == PDX_AGG_MEAN ? sum / (double)cnt : sum) : 0.0;
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
    PDX_HIP(hipMemcpyAsync(&h, total, sizeof(h), hipMemcpyDeviceToHost, st));
    PDX_HIP(hipStreamSynchronize(st));
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
