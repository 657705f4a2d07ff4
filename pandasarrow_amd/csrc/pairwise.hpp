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

// ---- NaN results carry the bits the reference's x86 host produces (round 4; measured against Arrow C++ 25, tests/test_oracle_golden_r4.py):
// an SSE add hands back its FIRST NaN operand (quieted), the second if only that is NaN, and the negative default NaN 0xFFF8000000000000 for
// inf + -inf; CDNA's v_add_f64 does neither reliably.  Which operand is "first" follows from how Arrow's loops were compiled:
//   inside a 16-value leaf (sum += value, the accumulator first)  : the EARLIER rows' NaN wins
//   in every merge of the tree (counter pushes, the final fold)    : the LATER operand's NaN wins (sum[cur] += b compiles to b + sum[cur])
// Every add of the tree code is written earlier + later; these two functions are that add with the NaN rule of its site.  The fix is a
// select on the (rare) NaN result: one compare per add on the fast path.
__device__ __forceinline__ double pw_quiet(double x) { return __longlong_as_double(__double_as_longlong(x) | 0x0008000000000000ll); }
__device__ __forceinline__ double pw_nan_of(double first, double second) {
  return first != first ? pw_quiet(first) : (second != second ? pw_quiet(second) : __longlong_as_double((long long)0xFFF8000000000000ull));
}
// (the test is made wave-uniform with a ballot: the fast path is the add, one compare and one scalar branch; the selects run only in a wave
//  that actually produced a NaN)
__device__ __forceinline__ double pw_leaf_add(double earlier, double later) {
  double r = earlier + later;
  if (__builtin_expect(__ballot(r != r) != 0ull, 0)) r = r == r ? r : pw_nan_of(earlier, later);
  return r;
}
__device__ __forceinline__ double pw_merge(double earlier, double later) {
  double r = earlier + later;
  if (__builtin_expect(__ballot(r != r) != 0ull, 0)) r = r == r ? r : pw_nan_of(later, earlier);
  return r;
}
// mean = sum / count: the x86 divide hands a NaN dividend back (quieted); the count is a positive number
__device__ __forceinline__ double pw_mean(double sum, double count) { return sum == sum ? sum / count : pw_quiet(sum); }
// a leaf sum that came out NaN, recomputed with the leaf rule (the plain chain is the fast path: NaN-ness never disappears in a sum)
template <typename Load>
__device__ __forceinline__ double pw_leaf_redo(int cnt, Load value_at) {
  double acc = 0.0;
  for (int q = 0; q < cnt; ++q) acc = pw_leaf_add(acc, value_at(q));
  return acc;
}

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
    sum[cur] = pw_merge(sum[cur], x);
    mask ^= m;
    while ((mask & m) == 0) {
      x = sum[cur];
      sum[cur] = 0.0;
      ++cur;
      m <<= 1;
      sum[cur] = pw_merge(sum[cur], x);
      mask ^= m;
    }
    if (cur > root) root = cur;
  }
  __device__ double finish() {
    for (int i = 1; i <= root; ++i) sum[i] = pw_merge(sum[i], sum[i - 1]);
    return sum[root];
  }
};

// ---- the partial-tree records of a group's rows [a, a + c) in its global row numbering (multi-GPU exchange, gb_partial_tree.hpp): the rows in
// front of the first and behind the last 16-row leaf boundary one by one, the leaves in between as the aligned blocks of the tree
__device__ __forceinline__ int64_t aligned_block_level(int64_t s, int64_t kl) {
  // largest j with s % 2^j == 0 and s + 2^j <= kl
  int tz = s == 0 ? 62 : __ffsll((unsigned long long)s) - 1;
  int64_t room = kl - s;
  int lg = 63 - __clzll((unsigned long long)room);
  return tz < lg ? tz : lg;
}
// record codes (key & 63): 0 = one row of a leaf begun on a lower rank; 1 .. 28 = a tree node of level code - 1; kPartialLeafCode + k = the
// first k rows (1 <= k <= 15) of a leaf as their sequential sum -- the owner continues that leaf with the next rank's rows.  (Until round 4
// these rows travelled one by one: with 125 rows per group and rank that was 7.5 of 17.5 records.)
constexpr int kPartialLeafCode = 32;
__device__ __forceinline__ int64_t partial_record_count(int64_t a, int64_t c) {
  if (c <= 0) return 0;
  int64_t b = a + c, kf = (a + 15) >> 4, kl = b >> 4;
  if (kf > kl) return c;  // the whole range lies inside one leaf (begun on a lower rank): every row is a record
  // the rows behind the last leaf boundary BEGIN a leaf: they leave as ONE record, their sequential sum (kPartialLeafCode + rows)
  int64_t cnt = (16 * kf - a) + (b > 16 * kl ? 1 : 0);
  for (int64_t s = kf; s < kl;) {
    s += (int64_t)1 << aligned_block_level(s, kl);
    ++cnt;
  }
  return cnt;
}

// perfect pairwise tree over the 64 lanes of a wave: lane 0 ends with ((l0+l1)+(l2+l3))+...  (6 levels)
__device__ __forceinline__ double wave_tree64(double x) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    double y = __shfl_down(x, d, 64);
    x = pw_merge(x, y);  // lanes that are multiples of 2d hold left + right; other lanes compute garbage that is never used
  }
  return x;
}
// perfect tree over 2^levels lanes starting at aligned lane groups (levels <= 6); result valid in the first lane of each group
__device__ __forceinline__ double wave_tree_levels(double x, int levels) {
  for (int s = 0; s < levels; ++s) {
    double y = __shfl_down(x, 1 << s, 64);
    x = pw_merge(x, y);
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
  if (threadIdx.x == 0) r = pw_merge(pw_merge(smem[0], smem[1]), pw_merge(smem[2], smem[3]));
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
  if (acc != acc) acc = pw_leaf_redo(cnt, [&](int q) { return v[q * stride]; });
  return acc;
}

}  // namespace pdx
