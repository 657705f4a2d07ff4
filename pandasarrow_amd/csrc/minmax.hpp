// minmax.hpp -- order-independent (value, row) extreme tracking: reproduces Arrow's min/max tie rule under any parallel
// reduction order.  Ties only show between values that compare equal but differ in bits (0.0 / -0.0).  Pinned against Arrow
// 25.0.0: min keeps the FIRST of them (min(0.0, -0.0) == 0.0); max keeps the FIRST when the array has no nulls and the LAST
// when it has at least one (its null-aware loop is compiled with fmax's operands the other way round) -- MAX_LAST selects that.
#pragma once
#include "pdx_common.hpp"

namespace pdx {

template <typename T, bool MAX_LAST = false>
struct Extreme {
  T vmin, vmax;
  long long rmin, rmax;
  __device__ void init() {
    rmin = rmax = -1;
    vmin = vmax = T(0);
  }
  __device__ static bool max_tie(long long row, long long cur) { return MAX_LAST ? row > cur : row < cur; }
  __device__ void add(T x, long long row) {
    if (rmin < 0 || x < vmin || (!(vmin < x) && row < rmin)) { vmin = x; rmin = row; }
    if (rmax < 0 || x > vmax || (!(vmax > x) && max_tie(row, rmax))) { vmax = x; rmax = row; }
  }
  __device__ void merge(T omin, long long ormin, T omax, long long ormax) {
    if (ormin >= 0 && (rmin < 0 || omin < vmin || (!(vmin < omin) && ormin < rmin))) { vmin = omin; rmin = ormin; }
    if (ormax >= 0 && (rmax < 0 || omax > vmax || (!(vmax > omax) && max_tie(ormax, rmax)))) { vmax = omax; rmax = ormax; }
  }
};
template <typename T>
struct MinMaxPartial {
  T vmin, vmax;
  long long rmin, rmax;
};

// Segments whose null presence is only known once they have been scanned (group-by): beside the first-wins extreme, a lane
// keeps the LAST zero-valued valid row it saw as (row << 1 | sign bit); -1 = none.  Reduced with max; applied by zero_tie_fix.
__device__ __forceinline__ long long zero_mark(double x, long long row) { return (row << 1) | (long long)(__double_as_longlong(x) < 0); }
__device__ __forceinline__ double zero_tie_fix(double vmax, long long zlast, bool segment_has_nulls) {
  if (segment_has_nulls && zlast >= 0 && vmax == 0.0) return (zlast & 1) ? -0.0 : 0.0;
  return vmax;
}

}  // namespace pdx
