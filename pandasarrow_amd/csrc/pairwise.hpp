// pairwise.hpp -- device building blocks that reproduce Arrow's fp64 `sum` bit-for-bit.
//
// Arrow (C++ 25.0.0) sums doubles pairwise: sequential 16-value leaves (restarting at every run of valid values),
// leaf sums merged by a binary counter (node = left + right, the earlier half on the left), leftovers folded from the
// lowest level up.  The reference reaches it from NDFrame::sum/mean (src/ndframe.cpp:26-31,162,220) and once per
// group from GROUPBY_AGG / GROUPBY_NUMERIC_AGG (src/pd_core_macros.h:31,66,103,132).
//
// Decomposition used here: a perfect subtree over 2^k consecutive leaves is exactly what the counter produces for
// those leaves, so 256 threads reduce 256 leaves with 8 adjacent-pair steps (6 by wave shuffles, 2 through LDS),
// higher levels recurse on the node arrays, and only the ragged tails (< 256 nodes per level) go through a literal
// single-thread replay of the counter (PairwiseCounter).  No FMA, no reassociation: every add is one v_add_f64.
#pragma once
#include "pdx_common.hpp"

namespace pdx {

// literal replay of Arrow's counter for pushes at arbitrary levels (levels only ever decrease across calls within
// one logical array: top-level tails first, then lower-level tails -- see aggregate.hip)
struct PairwiseCounter {
  double sum[64];
  uint64_t mask;
  int root;
  __device__ void init() {
    for (int i = 0; i < 64; ++i) sum[i] = 0.0;
    mask = 0;
    root = 0;
  }
  __device__ void push(double x, int level) {
    int cur = level;
    uint64_t m = 1ull << level;
    sum[cur] += x;
    mask ^= m;
    while ((mask & m) == 0) {
      x = sum[cur];
      sum[cur] = 0.0;
      ++cur;
      m <<= 1;
      sum[cur] += x;
      mask ^= m;
    }
    if (cur > root) root = cur;
  }
  __device__ double finish() {
    for (int i = 1; i <= root; ++i) sum[i] += sum[i - 1];
    return sum[root];
  }
};

// perfect pairwise tree over the 64 lanes of a wave: lane 0 ends with ((l0+l1)+(l2+l3))+...  (6 levels)
__device__ __forceinline__ double wave_tree64(double x) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    double y = __shfl_down(x, d, 64);
    x = x + y;  // lanes that are multiples of 2d hold left + right; other lanes compute garbage that is never used
  }
  return x;
}
// perfect tree over 2^levels lanes starting at aligned lane groups (levels <= 6); result valid in the first lane of each group
__device__ __forceinline__ double wave_tree_levels(double x, int levels) {
  for (int s = 0; s < levels; ++s) {
    double y = __shfl_down(x, 1 << s, 64);
    x = x + y;
  }
  return x;
}

// perfect tree over the 256 threads of a block (thread t holds leaf t); result returned in thread 0.  smem >= 4 doubles.
__device__ __forceinline__ double block_tree256(double x, double* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double w = wave_tree64(x);
  if (lane == 0) smem[wave] = w;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) r = (smem[0] + smem[1]) + (smem[2] + smem[3]);
  __syncthreads();
  return r;
}

// sequential leaf: ((((0.0 + v0) + v1) + ...) + v[cnt-1]), cnt <= 16, values at stride `stride` doubles
__device__ __forceinline__ double leaf_sum(const double* v, int cnt, int stride = 1) {
  double acc = 0.0;
  if (cnt == 16) {
#pragma unroll
    for (int q = 0; q < 16; ++q) acc += v[q * stride];
  } else {
    for (int q = 0; q < cnt; ++q) acc += v[q * stride];
  }
  return acc;
}

}  // namespace pdx
