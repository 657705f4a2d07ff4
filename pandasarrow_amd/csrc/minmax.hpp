// minmax.hpp -- order-independent (value, row) extreme tracking: reproduces Arrow's min/max tie rule (the FIRST of
// equal-comparing values wins, e.g. min(0.0, -0.0) == 0.0) under any parallel reduction order.
#pragma once
#include "pdx_common.hpp"

namespace pdx {

template <typename T>
struct Extreme {
  T vmin, vmax;
  long long rmin, rmax;
  __device__ void init() {
    rmin = rmax = -1;
    vmin = vmax = T(0);
  }
  __device__ void add(T x, long long row) {
    if (rmin < 0 || x < vmin || (!(vmin < x) && row < rmin)) { vmin = x; rmin = row; }
    if (rmax < 0 || x > vmax || (!(vmax > x) && row < rmax)) { vmax = x; rmax = row; }
  }
  __device__ void merge(T omin, long long ormin, T omax, long long ormax) {
    if (ormin >= 0 && (rmin < 0 || omin < vmin || (!(vmin < omin) && ormin < rmin))) { vmin = omin; rmin = ormin; }
    if (ormax >= 0 && (rmax < 0 || omax > vmax || (!(vmax > omax) && ormax < rmax))) { vmax = omax; rmax = ormax; }
  }
};
template <typename T>
struct MinMaxPartial {
  T vmin, vmax;
  long long rmin, rmax;
};


}  // namespace pdx
