// scan.hpp -- device-wide exclusive scan (reduce / scan-aggregates / apply) for gfx950, header-only templates.
// Used for radix-sort offsets, filter compaction, run/leaf bookkeeping.  Wave = 64: wave scans use
// __shfl_up over 64 lanes; a 256-thread block scans 2048 items per tile.
#pragma once
#include "pdx_common.hpp"

namespace pdx {

struct SumOp {
  template <typename T>
  __device__ __forceinline__ T operator()(T a, T b) const { return a + b; }
  template <typename T>
  __device__ __forceinline__ static T identity() { return T(0); }
};
// "latest defined": b if b != NONE else a.  NONE = -1.
struct LatestOp {
  template <typename T>
  __device__ __forceinline__ T operator()(T a, T b) const { return b != T(-1) ? b : a; }
  template <typename T>
  __device__ __forceinline__ static T identity() { return T(-1); }
};
struct MaxOp {
  template <typename T>
  __device__ __forceinline__ T operator()(T a, T b) const { return a > b ? a : b; }
  template <typename T>
  __device__ __forceinline__ static T identity() { return T(-1); }  // only used on non-negative data
};

constexpr int kScanBlock = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanBlock * kScanItems;

// inclusive scan across the 64 lanes of a wave
template <typename T, typename Op>
__device__ __forceinline__ T wave_inclusive_scan(T x, Op op) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T y = __shfl_up(x, d, 64);
    if (lane >= d) x = op(y, x);
  }
  return x;
}

// block-wide exclusive scan of one value per thread (blockDim.x == kScanBlock); returns exclusive prefix and total
template <typename T, typename Op>
__device__ __forceinline__ T block_exclusive_scan(T x, Op op, T* total, T* smem /* >= 8 entries */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = blockDim.x >> 6;
  T inc = wave_inclusive_scan(x, op);
  if (lane == 63) smem[wave] = inc;
  __syncthreads();
  T wave_prefix = Op::template identity<T>();
  T tot = Op::template identity<T>();
  for (int w = 0; w < nwaves; ++w) {
    T v = smem[w];
    if (w < wave) wave_prefix = op(wave_prefix, v);
    tot = op(tot, v);
  }
  __syncthreads();
  T exc = __shfl_up(inc, 1, 64);
  if (lane == 0) exc = Op::template identity<T>();
  *total = tot;
  return op(wave_prefix, exc);
}

template <typename T, typename Op>
__global__ void __launch_bounds__(kScanBlock) k_scan_reduce(const T* __restrict__ in, int64_t n, T* __restrict__ block_agg) {
  __shared__ T smem[8];
  Op op;
  int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  T acc = Op::template identity<T>();
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
    if (base + k < n) acc = op(acc, in[base + k]);
  T total;
  (void)block_exclusive_scan(acc, op, &total, smem);
  if (threadIdx.x == 0) block_agg[blockIdx.x] = total;
}

// single block: exclusive scan of up to kScanTile * many items sequentially by tiles (used on block aggregates)
template <typename T, typename Op>
__global__ void __launch_bounds__(kScanBlock) k_scan_single(T* __restrict__ data, int64_t n, T* __restrict__ total_out) {
  __shared__ T smem[8];
  Op op;
  T carry = Op::template identity<T>();
  for (int64_t tile = 0; tile < n; tile += kScanTile) {
    int64_t base = tile + (int64_t)threadIdx.x * kScanItems;
    T v[kScanItems];
    T acc = Op::template identity<T>();
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      v[k] = (base + k < n) ? data[base + k] : Op::template identity<T>();
      acc = op(acc, v[k]);
    }
    T total;
    T pre = block_exclusive_scan(acc, op, &total, smem);
    T run = op(carry, pre);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      if (base + k < n) data[base + k] = run;
      run = op(run, v[k]);
    }
    carry = op(carry, total);
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}

template <typename T, typename Op>
__global__ void __launch_bounds__(kScanBlock) k_scan_apply(const T* __restrict__ in, T* __restrict__ out, int64_t n,
                                                           const T* __restrict__ block_prefix) {
  __shared__ T smem[8];
  Op op;
  int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  T v[kScanItems];
  T acc = Op::template identity<T>();
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    v[k] = (base + k < n) ? in[base + k] : Op::template identity<T>();
    acc = op(acc, v[k]);
  }
  T total;
  T pre = block_exclusive_scan(acc, op, &total, smem);
  T run = op(block_prefix[blockIdx.x], pre);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k < n) out[base + k] = run;
    run = op(run, v[k]);
  }
}

// Exclusive scan of n items (in may alias out).  total_out (device pointer, may be null) receives the grand total.
// Scratch is taken from `s`.  Two levels handle n up to kScanTile * (anything): aggregates are scanned by one block.
template <typename T, typename Op>
int device_exclusive_scan(const T* in, T* out, int64_t n, T* total_out, Scratch& s, hipStream_t st) {
  if (n <= 0) {
    if (total_out) {
      // identity total
      // identity total: 0 for SumOp, all-ones (-1) for LatestOp/MaxOp
      PDX_HIP(hipMemsetAsync(total_out, __is_same(Op, SumOp) ? 0 : 0xFF, sizeof(T), st));
    }
    return PDX_OK;
  }
  int64_t nblocks = ceil_div(n, kScanTile);
  T* agg = s.get<T>((size_t)nblocks);
  PDX_SCRATCH_CHECK(s);
  hipLaunchKernelGGL((k_scan_reduce<T, Op>), dim3((unsigned)nblocks), dim3(kScanBlock), 0, st, in, n, agg);
  hipLaunchKernelGGL((k_scan_single<T, Op>), dim3(1), dim3(kScanBlock), 0, st, agg, nblocks, total_out);
  hipLaunchKernelGGL((k_scan_apply<T, Op>), dim3((unsigned)nblocks), dim3(kScanBlock), 0, st, in, out, n, agg);
  PDX_LAUNCH_CHECK();
  return PDX_OK;
}

}  // namespace pdx
