// gb_seg_reduce.hpp -- part of groupby.hip (textually included there, in its namespace context; split by stage, kernels unchanged):
// segmented reducers over fully sorted values: one wave per group, batched short groups, many waves per long group, nullable.
#pragma once

// ---------------------------------------------------------------- segmented reduce (dense values: no nulls)
struct SegOut {
  double* sum_f;     // SUM of float64 values, or nullptr
  long long* sum_i;  // SUM of int64 values (wrapping)
  double* mean;
  void* vmin;        // T*
  void* vmax;        // T*
  long long* count;
};

constexpr int kSegWaves = 4;
constexpr int64_t kBigSeg = 65536;  // rows per sub-segment of a long group = 2^12 sixteen-value leaves: a full one is ONE level-12 node
constexpr int kBigLevels = 16;      // counter levels kept per sub-segment (0..12 are used)

template <typename T>
__device__ __forceinline__ double seg_to_f64(T x) { return (double)x; }

// uniform (whole-wave) replay of Arrow's counter with the level sums in LDS; lane 0 stores, every lane reads
__device__ __forceinline__ void lds_counter_push(double* csum, uint64_t& mask, int& root, double x, int level, int lane) {
  int cur = level;
  uint64_t mb = 1ull << level;
  double v = pw_merge(csum[cur], x);
  mask ^= mb;
  while ((mask & mb) == 0) {
    if (lane == 0) csum[cur] = 0.0;
    ++cur;
    mb <<= 1;
    v = pw_merge(csum[cur], v);
    mask ^= mb;
  }
  if (lane == 0) csum[cur] = v;
  if (cur > root) root = cur;
}

// The wave walks its (group, chunk) sequence with the NEXT chunk's 16 loads per lane already in flight while the current chunk
// is staged and reduced, and the bounds of the next group loaded one group ahead: without this every group pays a full
// dependent seg_start -> values memory round trip with nothing else to do (measured 3.0 -> see DESIGN.md).
template <typename T, bool WANT_PAIRWISE, bool WANT_MINMAX, bool WANT_ISUM>
__global__ void __launch_bounds__(kSegWaves * 64) k_seg_reduce(const T* __restrict__ vals, const uint32_t* __restrict__ seg_start,
                                                               int64_t nseg, const uint32_t* __restrict__ out_index, SegOut out, int64_t min_len) {
  // groups of <= min_len rows belong to k_seg_reduce_mid (batches of short groups per wave) and are skipped here like the long ones
  constexpr int LEAF = 16;              // Arrow's kBlockSize
  constexpr int kSegChunk = 64 * LEAF;  // values per wave-chunk = 64 leaves
  __shared__ double stage[kSegWaves][64 * 17];
  __shared__ double csum_all[kSegWaves][48];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* lds = stage[wave];
  double* csum = csum_all[wave];
  const int64_t nw = (int64_t)gridDim.x * kSegWaves;
  int64_t k = (int64_t)blockIdx.x * kSegWaves + wave;
  if (k >= nseg) return;
  // groups longer than kBigSeg rows are reduced by k_seg_reduce_sub / k_seg_combine_big (many waves per group): here they
  // are walked as empty segments whose result is not written
  int64_t s = seg_start[k], e = seg_start[k + 1];
  bool big = e - s > kBigSeg || e - s <= min_len;
  if (big) e = s;
  int64_t s_next = 0, e_next = 0;  // bounds of group k + nw
  bool big_next = false;
  if (k + nw < nseg) {
    s_next = seg_start[k + nw];
    e_next = seg_start[k + nw + 1];
    big_next = e_next - s_next > kBigSeg || e_next - s_next <= min_len;
    if (big_next) e_next = s_next;
  }
  int64_t c0 = 0;
  T cur[LEAF];
  {
    const int cl = (int)((e - s) < kSegChunk ? (e - s) : kSegChunk);
#pragma unroll
    for (int q = 0; q < LEAF; ++q) {
      int idx = q * 64 + lane;
      cur[q] = idx < cl ? vals[s + idx] : T(0);
    }
  }
  Extreme<T> ext;
  ext.init();
  unsigned long long isum = 0;
  uint64_t mask = 0;
  int root = 0;
  double single = 0.0;  // result when the group fits one chunk
  if (WANT_PAIRWISE && (e - s) > kSegChunk) {
    if (lane < 48) csum[lane] = 0.0;
  }
  for (;;) {
    const int64_t len = e - s;
    const bool multi = len > kSegChunk;
    const int cl = (int)((len - c0) < kSegChunk ? (len - c0) : kSegChunk);
    const bool last_chunk = c0 + kSegChunk >= len;
    // ---- issue the next chunk's loads
    const int64_t nk = last_chunk ? k + nw : k;
    const bool have_next = nk < nseg;
    const int64_t ns = last_chunk ? s_next : s, ne = last_chunk ? e_next : e, nc0 = last_chunk ? 0 : c0 + kSegChunk;
    T nxt[LEAF];
    if (have_next) {
      const int ncl = (int)((ne - ns - nc0) < kSegChunk ? (ne - ns - nc0) : kSegChunk);
#pragma unroll
      for (int q = 0; q < LEAF; ++q) {
        int idx = q * 64 + lane;
        nxt[q] = idx < ncl ? vals[ns + nc0 + idx] : T(0);
      }
    }
    int64_t s_nn = 0, e_nn = 0;
    bool big_nn = false;
    if (last_chunk && nk + nw < nseg) {  // bounds two groups ahead, consumed when the next group finishes
      s_nn = seg_start[nk + nw];
      e_nn = seg_start[nk + nw + 1];
      big_nn = e_nn - s_nn > kBigSeg || e_nn - s_nn <= min_len;
      if (big_nn) e_nn = s_nn;
    }
    // ---- current chunk
#pragma unroll
    for (int q = 0; q < LEAF; ++q) {
      int idx = q * 64 + lane;
      if (idx < cl) {
        T x = cur[q];
        if (WANT_PAIRWISE) lds[idx + (idx >> 4)] = seg_to_f64(x);
        if (WANT_MINMAX) {
          if (x == x) ext.add(x, (long long)(c0 + idx));
        }
        if (WANT_ISUM) isum += (unsigned long long)x;
      }
    }
    __builtin_amdgcn_wave_barrier();  // the LDS image is wave-private: in-order LDS issue makes it visible to all lanes
    if (WANT_PAIRWISE) {
      const int m = (cl + LEAF - 1) / LEAF;  // leaves in this chunk (wave-uniform)
      double x = 0.0;
      const int first = lane * LEAF;
      if (first < cl) {
        int cnt = cl - first < 16 ? cl - first : 16;
        x = leaf_sum(&lds[lane * 17], cnt);
      }
      // butterfly; pick the perfect subtrees that tile [0, m)
      double node[7];
#pragma unroll
      for (int sft = 0; sft < 6; ++sft) {
        node[sft] = 0.0;
        if ((m >> sft) & 1) node[sft] = __shfl(x, m & ~((2 << sft) - 1), 64);
        double y = __shfl_down(x, 1 << sft, 64);
        x = pw_merge(x, y);
      }
      node[6] = __shfl(x, 0, 64);
      if (!multi) {
        // fold ascending: acc = lowest node; acc = higher + acc
        bool have = false;
        double acc = 0.0;
#pragma unroll
        for (int sft = 0; sft <= 6; ++sft) {
          if ((m >> sft) & 1) {
            acc = have ? pw_merge(node[sft], acc) : node[sft];
            have = true;
          }
        }
        single = acc;
      } else {
#pragma unroll
        for (int sft = 6; sft >= 0; --sft)
          if ((m >> sft) & 1) lds_counter_push(csum, mask, root, node[sft], sft, lane);
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (last_chunk) {
      // ---- group k is complete
      const uint32_t oi = out_index ? out_index[k] : (uint32_t)k;
      double total = single;
      if (WANT_PAIRWISE && multi) {
        double acc = csum[0];
        for (int i = 1; i <= root; ++i) acc = pw_merge(csum[i], acc);
        total = acc;
      }
      if (WANT_MINMAX) {
        for (int d = 32; d > 0; d >>= 1) {
          T omin = __shfl_down(ext.vmin, d, 64), omax = __shfl_down(ext.vmax, d, 64);
          long long ormin = __shfl_down(ext.rmin, d, 64), ormax = __shfl_down(ext.rmax, d, 64);
          ext.merge(omin, ormin, omax, ormax);
        }
      }
      if (WANT_ISUM) {
        for (int d = 32; d > 0; d >>= 1) isum += __shfl_down(isum, d, 64);
      }
      if (lane == 0 && !big) {
        if (WANT_PAIRWISE) {
          if (out.sum_f) out.sum_f[oi] = total;
          if (out.mean) out.mean[oi] = pw_mean(total, (double)len);
        }
        if (WANT_ISUM && out.sum_i) out.sum_i[oi] = (long long)isum;
        if (WANT_MINMAX) {
          T nanv = T(0);
          if constexpr (__is_same(T, double)) nanv = __builtin_nan("");
          if (out.vmin) static_cast<T*>(out.vmin)[oi] = ext.rmin < 0 ? nanv : ext.vmin;
          if (out.vmax) static_cast<T*>(out.vmax)[oi] = ext.rmax < 0 ? nanv : ext.vmax;
        }
        if (out.count) out.count[oi] = (long long)len;
      }
      if (!have_next) break;
      // ---- reset the per-group state
      ext.init();
      isum = 0;
      mask = 0;
      root = 0;
      single = 0.0;
      big = big_next;
      s_next = s_nn;
      e_next = e_nn;
      big_next = big_nn;
      if (WANT_PAIRWISE && (ne - ns) > kSegChunk) {
        __builtin_amdgcn_wave_barrier();
        if (lane < 48) csum[lane] = 0.0;
      }
    }
    k = nk;
    s = ns;
    e = ne;
    c0 = nc0;
#pragma unroll
    for (int q = 0; q < LEAF; ++q) cur[q] = nxt[q];
  }
}

// ---------------------------------------------------------------- one wave reduces one contiguous segment (any length).
// Chunks of 1024 values (64 leaves), the next chunk's loads in flight while the current one is staged and reduced; the chunk's
// perfect subtrees go through the LDS-resident counter (csum/mask/root: Arrow's state after the segment; the caller folds it or
// stores it).  ext (wave-reduced, valid in lane 0) and isum (wave-reduced) cover the whole segment; rows are numbered from row_base.
template <typename T, bool WANT_PAIRWISE, bool WANT_MINMAX, bool WANT_ISUM, bool PREFETCH = true>
__device__ __forceinline__ void seg_chunked(const T* __restrict__ vals, int64_t s, int64_t len, long long row_base, int lane, double* lds /* 64*17 */,
                                            double* csum /* 48 */, Extreme<T>& ext, unsigned long long& isum, uint64_t& mask, int& root) {
  constexpr int LEAF = 16;
  constexpr int kSegChunk = 64 * LEAF;
  __builtin_amdgcn_wave_barrier();
  if (lane < 48) csum[lane] = 0.0;
  // PREFETCH = false (callers with many live registers of their own): plain load-then-reduce per chunk, half the registers
  T cur[LEAF];
  if (PREFETCH) {
    const int cl = (int)(len < kSegChunk ? len : kSegChunk);
#pragma unroll
    for (int q = 0; q < LEAF; ++q) {
      int idx = q * 64 + lane;
      cur[q] = idx < cl ? vals[s + idx] : T(0);
    }
  }
  for (int64_t c0 = 0; c0 < len; c0 += kSegChunk) {
    const int cl = (int)((len - c0) < kSegChunk ? (len - c0) : kSegChunk);
    T nxt[PREFETCH ? LEAF : 1];
    if (PREFETCH) {
      const int64_t n0 = c0 + kSegChunk;
      const int ncl = n0 < len ? (int)((len - n0) < kSegChunk ? (len - n0) : kSegChunk) : 0;
#pragma unroll
      for (int q = 0; q < LEAF; ++q) {
        int idx = q * 64 + lane;
        nxt[PREFETCH ? q : 0] = idx < ncl ? vals[s + n0 + idx] : T(0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < LEAF; ++q) {
        int idx = q * 64 + lane;
        cur[q] = idx < cl ? vals[s + c0 + idx] : T(0);
      }
    }
#pragma unroll
    for (int q = 0; q < LEAF; ++q) {
      int idx = q * 64 + lane;
      if (idx < cl) {
        T x = cur[q];
        if (WANT_PAIRWISE) lds[idx + (idx >> 4)] = seg_to_f64(x);
        if (WANT_MINMAX) {
          if (x == x) ext.add(x, row_base + (long long)(c0 + idx));
        }
        if (WANT_ISUM) isum += (unsigned long long)x;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (WANT_PAIRWISE) {
      const int m = (cl + LEAF - 1) / LEAF;
      double x = 0.0;
      const int first = lane * LEAF;
      if (first < cl) {
        int cnt = cl - first < 16 ? cl - first : 16;
        x = leaf_sum(&lds[lane * 17], cnt);
      }
      double node[7];
#pragma unroll
      for (int sft = 0; sft < 6; ++sft) {
        node[sft] = 0.0;
        if ((m >> sft) & 1) node[sft] = __shfl(x, m & ~((2 << sft) - 1), 64);
        double y = __shfl_down(x, 1 << sft, 64);
        x = pw_merge(x, y);
      }
      node[6] = __shfl(x, 0, 64);
#pragma unroll
      for (int sft = 6; sft >= 0; --sft)
        if ((m >> sft) & 1) lds_counter_push(csum, mask, root, node[sft], sft, lane);
    }
    __builtin_amdgcn_wave_barrier();
    if (PREFETCH) {
#pragma unroll
      for (int q = 0; q < LEAF; ++q) cur[q] = nxt[PREFETCH ? q : 0];
    }
  }
  if (WANT_MINMAX) {
    for (int d = 32; d > 0; d >>= 1) {
      T omin = __shfl_down(ext.vmin, d, 64), omax = __shfl_down(ext.vmax, d, 64);
      long long ormin = __shfl_down(ext.rmin, d, 64), ormax = __shfl_down(ext.rmax, d, 64);
      ext.merge(omin, ormin, omax, ormax);
    }
  }
  if (WANT_ISUM) {
    for (int d = 32; d > 0; d >>= 1) isum += __shfl_down(isum, d, 64);
  }
}

// ---------------------------------------------------------------- segmented reduce, short groups: batches of groups per wave.
// Groups of a few to a few hundred rows leave most of a wave idle in k_seg_reduce (and a thread per group thrashes the L1).  Here a wave takes a
// run of consecutive short groups (<= 64 groups, <= 1024 rows: the grouped values are contiguous), loads the whole run coalesced
// into LDS, sums the 16-value leaves with one lane per leaf and then combines every group's leaves with one lane per group
// (in-place perfect subtrees + ascending fold == Arrow's counter).  Groups longer than kMidLen are left to k_seg_reduce.
constexpr int kMidLen = 256;
constexpr int kMidRows = 1024;
template <typename T, bool WANT_PAIRWISE, bool WANT_MINMAX, bool WANT_ISUM>
__global__ void __launch_bounds__(kSegWaves * 64) k_seg_reduce_mid(const T* __restrict__ vals, const uint32_t* __restrict__ seg_start, int64_t nseg,
                                                                   const uint32_t* __restrict__ out_index, SegOut out, int64_t groups_per_wave) {
  __shared__ T stage_all[kSegWaves][64 * 17];  // >= kMidRows values; the padded 64 x 17 image when a longer group is chunked
  __shared__ double csum_all[kSegWaves][48];
  __shared__ double leaf_all[kSegWaves][kMidRows / 16 + 64];
  __shared__ int lp_all[kSegWaves][65];
  __shared__ int goff_all[kSegWaves][64];
  __shared__ int glen_all[kSegWaves][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* stage = stage_all[wave];
  double* leaf = leaf_all[wave];
  int* lp = lp_all[wave];
  int* goff = goff_all[wave];
  int* glen = glen_all[wave];
  const int64_t gw = (int64_t)blockIdx.x * kSegWaves + wave;
  int64_t k0 = gw * groups_per_wave;
  const int64_t kend = k0 + groups_per_wave < nseg ? k0 + groups_per_wave : nseg;
  while (k0 < kend) {
    const int64_t kk = k0 + lane;
    const int64_t b0 = seg_start[kk < kend ? kk : kend], b1 = seg_start[kk + 1 < kend ? kk + 1 : kend];
    const int64_t S = __shfl(b0, 0, 64);
    const int len = (int)(b1 - b0);
    const bool ok = kk < kend && len <= kMidLen && (b1 - S) <= kMidRows;
    const uint64_t okm = __ballot(ok);
    const int g = ~okm ? __ffsll((unsigned long long)~okm) - 1 : 64;  // leading run of short groups that fits
    if (g == 0) {
      // a longer group: the whole wave chunks through it (groups beyond kBigSeg belong to the many-waves path)
      const int64_t glen0 = __shfl(b1, 0, 64) - S;
      if (glen0 <= kBigSeg) {
        Extreme<T> ext;
        ext.init();
        unsigned long long isum = 0;
        uint64_t mask = 0;
        int root = 0;
        double* csum = csum_all[wave];
        seg_chunked<T, WANT_PAIRWISE, WANT_MINMAX, WANT_ISUM, false>(vals, S, glen0, 0ll, lane, reinterpret_cast<double*>(stage), csum, ext, isum, mask,
                                                                     root);
        double total = 0.0;
        if (WANT_PAIRWISE) {
          double acc = csum[0];
          for (int i = 1; i <= root; ++i) acc = pw_merge(csum[i], acc);
          total = acc;
        }
        if (lane == 0) {
          const uint32_t oi = out_index ? out_index[k0] : (uint32_t)k0;
          if (WANT_PAIRWISE) {
            if (out.sum_f) out.sum_f[oi] = total;
            if (out.mean) out.mean[oi] = pw_mean(total, (double)glen0);
          }
          if (WANT_ISUM && out.sum_i) out.sum_i[oi] = (long long)isum;
          if (WANT_MINMAX) {
            T nanv = T(0);
            if constexpr (__is_same(T, double)) nanv = __builtin_nan("");
            if (out.vmin) static_cast<T*>(out.vmin)[oi] = ext.rmin < 0 ? nanv : ext.vmin;
            if (out.vmax) static_cast<T*>(out.vmax)[oi] = ext.rmax < 0 ? nanv : ext.vmax;
          }
          if (out.count) out.count[oi] = (long long)glen0;
        }
        __builtin_amdgcn_wave_barrier();
      }
      k0 += 1;
      continue;
    }
    const int R = (int)(__shfl(b1, g - 1, 64) - S);
#pragma unroll
    for (int q = 0; q < kMidRows / 64; ++q) {
      int idx = q * 64 + lane;
      if (idx < R) stage[idx] = vals[S + idx];
    }
    const int nl = lane < g ? (len + 15) >> 4 : 0;
    const int incl = wave_inclusive_scan(nl, SumOp());
    const int excl = incl - nl;
    const int NL = __shfl(incl, 63, 64);
    lp[lane] = excl;
    if (lane == 63) lp[64] = NL;
    goff[lane] = (int)(b0 - S);
    glen[lane] = len;
    __builtin_amdgcn_wave_barrier();
    if (WANT_PAIRWISE) {
      for (int L = lane; L < NL; L += 64) {
        int lo = 0, hi = g - 1;
        while (lo < hi) {
          int mid = (lo + hi + 1) >> 1;
          if (lp[mid] <= L) lo = mid;
          else hi = mid - 1;
        }
        const int j = L - lp[lo];
        const int off = goff[lo] + 16 * j;
        int cnt = glen[lo] - 16 * j;
        cnt = cnt < 16 ? cnt : 16;
        double acc = 0.0;
        if (cnt == 16) {
#pragma unroll
          for (int q = 0; q < 16; ++q) acc += seg_to_f64(stage[off + q]);
        } else {
          for (int q = 0; q < cnt; ++q) acc += seg_to_f64(stage[off + q]);
        }
        leaf[L] = acc;
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (lane < g) {
      const uint32_t oi = out_index ? out_index[kk] : (uint32_t)kk;
      if (WANT_PAIRWISE) {
        double* x = leaf + excl;
        const int m = nl;
        // plain adds on the fast path: a NaN anywhere in the group's tree reaches the total, and only then is the tree walked again with
        // the x86 NaN rules of its adds (pairwise.hpp) -- the group's rows are still staged, its leaf slots are reused
        auto tree = [&](auto merge) {
          for (int stride = 1; stride < m; stride <<= 1)
            for (int i = 0; i + 2 * stride <= m; i += 2 * stride) x[i] = merge(x[i], x[i + stride]);
          double a = 0.0;
          bool have = false;
          int pos = m;
          for (int jb = 0; jb < 7; ++jb)
            if ((m >> jb) & 1) {
              pos -= 1 << jb;
              a = have ? merge(x[pos], a) : x[pos];
              have = true;
            }
          return a;
        };
        double acc = tree([](double e, double l) { return e + l; });
        if (acc != acc) {
          const T* gv = stage + (int)(b0 - S);
          for (int j = 0; j < m; ++j) {
            const int cnt = len - 16 * j < 16 ? len - 16 * j : 16;
            x[j] = pw_leaf_redo(cnt, [&](int q) { return seg_to_f64(gv[16 * j + q]); });
          }
          acc = tree([](double e, double l) { return pw_merge(e, l); });
        }
        if (out.sum_f) out.sum_f[oi] = acc;
        if (out.mean) out.mean[oi] = pw_mean(acc, (double)len);
      }
      if (WANT_MINMAX || WANT_ISUM) {
        const T* v = stage + (int)(b0 - S);
        unsigned long long isum = 0;
        T vmn = T(0), vmx = T(0);
        bool has = false;
        for (int r = 0; r < len; ++r) {
          T xv = v[r];
          if (WANT_ISUM) isum += (unsigned long long)xv;
          if (WANT_MINMAX && xv == xv) {
            if (!has) { vmn = vmx = xv; has = true; }
            else {
              if (xv < vmn) vmn = xv;
              if (xv > vmx) vmx = xv;
            }
          }
        }
        if (WANT_ISUM && out.sum_i) out.sum_i[oi] = (long long)isum;
        if (WANT_MINMAX) {
          T nanv = T(0);
          if constexpr (__is_same(T, double)) nanv = __builtin_nan("");
          if (out.vmin) static_cast<T*>(out.vmin)[oi] = has ? vmn : nanv;
          if (out.vmax) static_cast<T*>(out.vmax)[oi] = has ? vmx : nanv;
        }
      }
      if (out.count) out.count[oi] = (long long)len;
    }
    __builtin_amdgcn_wave_barrier();
    k0 += g;
  }
}

// ---------------------------------------------------------------- long groups: many waves per group.
// A group of more than kBigSeg rows is cut into sub-segments of kBigSeg rows (aligned to the group start, so every full
// sub-segment is a perfect subtree of 2^12 leaves = one level-12 node of Arrow's counter).  One wave reduces one sub-segment to
// its counter state; one thread per long group then replays the states in order (full ones are a single level-12 push, the last
// one pushes its <= 13 nodes from the highest level down, which is legal because everything before it is 2^12-aligned).
template <typename T>
struct SubState {
  double csum[kBigLevels];
  unsigned long long mask;
  unsigned long long isum;
  T vmin, vmax;
  long long rmin, rmax;
};
// The long groups, in no particular order (each is reduced on its own, so the order is immaterial): *count of them.  The kernels below read
// the count from the device -- their launches are sized for the most long groups `nrows` rows can hold -- so the host never waits for it.
__global__ void k_big_append(const uint32_t* __restrict__ seg_start, int64_t nseg, uint32_t* __restrict__ big_idx, int64_t* __restrict__ count) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nseg; k += stride)
    if ((int64_t)seg_start[k + 1] - (int64_t)seg_start[k] > kBigSeg) big_idx[atomicAdd(reinterpret_cast<unsigned long long*>(count), 1ull)] = (uint32_t)k;
}
// item_off[b] = first work item (sub-segment) of long group b; item_off[B] = number of items.  One workgroup.
__global__ void __launch_bounds__(256) k_big_offsets(const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ big_idx,
                                                     const int64_t* __restrict__ count, int64_t* __restrict__ item_off) {
  __shared__ int64_t smem[8];
  const int64_t B = *count;
  int64_t carry = 0;
  for (int64_t b0 = 0; b0 < B; b0 += 256) {
    int64_t b = b0 + threadIdx.x;
    int64_t nsub = 0;
    if (b < B) {
      const uint32_t k = big_idx[b];
      nsub = ((int64_t)seg_start[k + 1] - (int64_t)seg_start[k] + kBigSeg - 1) / kBigSeg;
    }
    int64_t total;
    int64_t pre = block_exclusive_scan(nsub, SumOp(), &total, smem);
    if (b < B) item_off[b] = carry + pre;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) item_off[B] = carry;
}
template <typename T, bool WANT_PAIRWISE, bool WANT_MINMAX, bool WANT_ISUM>
__global__ void __launch_bounds__(kSegWaves * 64) k_seg_reduce_sub(const T* __restrict__ vals, const uint32_t* __restrict__ seg_start,
                                                                   const uint32_t* __restrict__ big_idx, const int64_t* __restrict__ item_off,
                                                                   const int64_t* __restrict__ count, SubState<T>* __restrict__ state) {
  __shared__ double stage[kSegWaves][64 * 17];
  __shared__ double csum_all[kSegWaves][48];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* lds = stage[wave];
  double* csum = csum_all[wave];
  const int64_t B = *count;
  if (B == 0) return;
  const int64_t nitems = item_off[B];
  const int64_t nw = (int64_t)gridDim.x * kSegWaves;
  for (int64_t t = (int64_t)blockIdx.x * kSegWaves + wave; t < nitems; t += nw) {
    // long group of item t: last b with item_off[b] <= t
    int64_t lo = 0, hi = B - 1;
    while (lo < hi) {
      int64_t mid = (lo + hi + 1) >> 1;
      if (item_off[mid] <= t) lo = mid;
      else hi = mid - 1;
    }
    const uint32_t k = big_idx[lo];
    const int64_t j = t - item_off[lo];
    const int64_t s = (int64_t)seg_start[k] + j * kBigSeg;
    const int64_t gend = seg_start[k + 1];
    const int64_t e = s + kBigSeg < gend ? s + kBigSeg : gend;
    Extreme<T> ext;
    ext.init();
    unsigned long long isum = 0;
    uint64_t mask = 0;
    int root = 0;
    seg_chunked<T, WANT_PAIRWISE, WANT_MINMAX, WANT_ISUM>(vals, s, e - s, (long long)(j * kBigSeg), lane, lds, csum, ext, isum, mask, root);
    if (lane < kBigLevels) state[t].csum[lane] = csum[lane];
    if (lane == 0) {
      state[t].mask = mask;
      state[t].isum = isum;
      state[t].vmin = ext.vmin;
      state[t].vmax = ext.vmax;
      state[t].rmin = ext.rmin;
      state[t].rmax = ext.rmax;
    }
  }
}
template <typename T, bool WANT_PAIRWISE, bool WANT_MINMAX, bool WANT_ISUM>
__global__ void __launch_bounds__(64) k_seg_combine_big(const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ big_idx,
                                                        const int64_t* __restrict__ item_off, const int64_t* __restrict__ count,
                                                        const SubState<T>* __restrict__ state, const uint32_t* __restrict__ out_index, SegOut out) {
  // one wave per long group.  64 consecutive FULL sub-segments (64-aligned within the group) are a perfect subtree of level-12
  // nodes: one butterfly makes their level-18 node; everything else is replayed by lane 0.
  const int64_t b = blockIdx.x;
  if (b >= *count) return;
  const int lane = threadIdx.x;
  const uint32_t k = big_idx[b];
  const uint32_t oi = out_index ? out_index[k] : k;
  const long long len = (long long)seg_start[k + 1] - (long long)seg_start[k];
  PairwiseCounter c;  // used by lane 0 only
  if (lane == 0) c.init();
  Extreme<T> ext;
  ext.init();
  unsigned long long isum = 0;
  const int64_t begin = item_off[b], end = item_off[b + 1];
  for (int64_t t0 = begin; t0 < end; t0 += 64) {
    const int64_t t = t0 + lane;
    const bool have = t < end;
    unsigned long long m = 0;
    double v = 0.0;
    if (have) {
      m = state[t].mask;
      v = state[t].csum[12];
      if (WANT_MINMAX) ext.merge(state[t].vmin, state[t].rmin, state[t].vmax, state[t].rmax);
      if (WANT_ISUM) isum += state[t].isum;
    }
    if (WANT_PAIRWISE) {
      const bool all_full = (end - t0 >= 64) && __all(m == (1ull << 12));
      if (all_full) {
        const double node = wave_tree64(v);
        if (lane == 0) c.push(node, 18);
      } else if (lane == 0) {
        const int64_t cnt = end - t0 < 64 ? end - t0 : 64;
        for (int64_t i = 0; i < cnt; ++i) {
          const SubState<T>& st = state[t0 + i];
          for (int lvl = kBigLevels - 1; lvl >= 0; --lvl)
            if ((st.mask >> lvl) & 1) c.push(st.csum[lvl], lvl);
        }
      }
    }
  }
  if (WANT_MINMAX) {
    for (int d = 32; d > 0; d >>= 1) {
      T omin = __shfl_down(ext.vmin, d, 64), omax = __shfl_down(ext.vmax, d, 64);
      long long ormin = __shfl_down(ext.rmin, d, 64), ormax = __shfl_down(ext.rmax, d, 64);
      ext.merge(omin, ormin, omax, ormax);
    }
  }
  if (WANT_ISUM) {
    for (int d = 32; d > 0; d >>= 1) isum += __shfl_down(isum, d, 64);
  }
  if (lane != 0) return;
  if (WANT_PAIRWISE) {
    const double total = c.finish();
    if (out.sum_f) out.sum_f[oi] = total;
    if (out.mean) out.mean[oi] = pw_mean(total, (double)len);
  }
  if (WANT_ISUM && out.sum_i) out.sum_i[oi] = (long long)isum;
  if (WANT_MINMAX) {
    T nanv = T(0);
    if constexpr (__is_same(T, double)) nanv = __builtin_nan("");
    if (out.vmin) static_cast<T*>(out.vmin)[oi] = ext.rmin < 0 ? nanv : ext.vmin;
    if (out.vmax) static_cast<T*>(out.vmax)[oi] = ext.rmax < 0 ? nanv : ext.vmax;
  }
  if (out.count) out.count[oi] = len;
}

// ---------------------------------------------------------------- segmented reduce (nullable values): one wave per group.
// Arrow restarts the 16-value leaves at every run of valid rows, so leaf boundaries are data dependent.  Per 1024-row chunk a
// lane owns a 16-row window: the number of rows already in the leaf that is open at the window start comes from a "latest"
// scan across the lanes (a window is as long as a leaf, so a full window passes the count through unchanged), the open leaf's
// partial sum is the sequential sum of the previous window's last rows (one shuffle), and each lane walks its 16 validity bits
// emitting finished leaves in order.  The emitted leaf sums are merged with a butterfly whose lanes are aligned to the GLOBAL leaf
// index, so every perfect subtree it extracts is exactly a run of carries of Arrow's binary counter.
constexpr int64_t kHugeNullable = (int64_t)1 << 22;  // rows: beyond this a nullable group is not left to one wave
constexpr int kNullLeafCap = 64 * 9 + 8 + 64;  // a 16-row window emits at most 9 leaves (8 isolated values + the carried one); + the queue's tail

template <typename T>
__global__ void __launch_bounds__(kSegWaves * 64) k_seg_reduce_nullable(const T* __restrict__ vals, const uint32_t* __restrict__ sorted_keys,
                                                                        const uint8_t* __restrict__ row_valid, int64_t valid_off,
                                                                        const uint32_t* __restrict__ seg_start, int64_t nseg,
                                                                        const uint32_t* __restrict__ out_index, SegOut out,
                                                                        uint8_t* __restrict__ ok) {
  __shared__ double stage[kSegWaves][64 * 17];
  __shared__ double leafbuf[kSegWaves][kNullLeafCap];
  __shared__ double csum_all[kSegWaves][48];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* lds = stage[wave];
  double* leaves = leafbuf[wave];
  double* csum = csum_all[wave];
  int64_t gw = (int64_t)blockIdx.x * kSegWaves + wave;
  int64_t nw = (int64_t)gridDim.x * kSegWaves;
  for (int64_t k = gw; k < nseg; k += nw) {
    const int64_t s = seg_start[k], e = seg_start[k + 1];
    const int64_t len = e - s;
    if (len > kHugeNullable) continue;  // reduced slice by slice with the whole-column kernels (reduce_huge_nullable_groups)
    const uint32_t oi = out_index ? out_index[k] : (uint32_t)k;
    Extreme<T> ext;
    ext.init();
    long long zlast = -1;     // last valid zero-valued row of the group and its sign: the max tie rule of a group WITH nulls (minmax.hpp)
    unsigned long long isum = 0;
    long long nvalid = 0;
    uint64_t cmask = 0;       // binary counter occupancy (wave-uniform)
    int croot = 0;
    int pend = 0;             // finished leaves waiting in the queue for their block of 64 (wave-uniform, < 64 between chunks)
    int carry_pos = 0;        // rows already in the leaf that is open at the chunk start
    double carry_acc = 0.0;   // ... and their sequential sum
    if (lane < 48) csum[lane] = 0.0;
    for (int64_t c0 = 0; c0 < len; c0 += 1024) {
      const int cl = (int)((len - c0) < 1024 ? (len - c0) : 1024);
      // coalesced loads + validity words; lane l keeps the word that holds its window [16l, 16l+16)
      uint64_t myword = 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int idx = q * 64 + lane;
        bool v = false;
        if (idx < cl) {
          const int64_t i = s + c0 + idx;
          v = sorted_keys ? !(sorted_keys[i] >> 31) : (!row_valid || bit_get(row_valid, valid_off + i));
          T x = vals[i];
          lds[idx + (idx >> 4)] = (double)x;
          if (v) {
            isum += (unsigned long long)x;
            if (x == x) ext.add(x, (long long)(c0 + idx));
            if constexpr (__is_same(T, double)) {
              if (x == 0.0) {
                const long long zm = zero_mark(x, (long long)(c0 + idx));
                zlast = zm > zlast ? zm : zlast;
              }
            }
          }
        }
        const uint64_t bal = __ballot(v);
        nvalid += __popcll(bal);
        if ((lane >> 2) == q) myword = bal;
      }
      __builtin_amdgcn_wave_barrier();
      const unsigned m = (unsigned)(myword >> ((lane & 3) * 16)) & 0xFFFFu;
      const bool full = m == 0xFFFFu;
      const int t = full ? 16 : __builtin_clz(~(m << 16));  // valid rows at the END of the window (leading ones of m << 16)
      // rows in the open leaf at the start of every window ("latest" scan; a full window passes its own start value on)
      const int z = full ? -1 : t;
      const int inc = wave_inclusive_scan(z, LatestOp());
      const int exc = __shfl_up(inc, 1, 64);
      const int pos = (lane != 0 && exc >= 0) ? exc : carry_pos;  // exc < 0: every earlier window of this chunk is full
      // sequential sum of this window's last rows that stay in an open leaf (handed to the next window)
      const int cnt_tail = full ? pos : t;
      double tail = 0.0;
      for (int q = 16 - cnt_tail; q < 16; ++q) tail = pw_leaf_add(tail, lds[lane * 17 + q]);
      double acc = __shfl_up(tail, 1, 64);
      if (lane == 0) acc = carry_acc;
      // pass 1: number of leaves this window finishes
      int nfin = 0;
      {
        int p = pos;
        for (int q = 0; q < 16; ++q) {
          if ((m >> q) & 1u) {
            if (++p == 16) { ++nfin; p = 0; }
          } else if (p > 0) { ++nfin; p = 0; }
        }
      }
      int inc_n = wave_inclusive_scan(nfin, SumOp());
      const int base = inc_n - nfin;
      const int total_new = __shfl(inc_n, 63, 64);
      // pass 2: emit the finished leaves in order
      {
        int p = pos, w = pend + base;
        double a = pos > 0 ? acc : 0.0;
        for (int q = 0; q < 16; ++q) {
          if ((m >> q) & 1u) {
            a = pw_leaf_add(p == 0 ? 0.0 : a, lds[lane * 17 + q]);
            if (++p == 16) { leaves[w++] = a; p = 0; }
          } else if (p > 0) { leaves[w++] = a; p = 0; }
        }
      }
      // state handed to the next chunk
      const int last_inc = __shfl(inc, 63, 64);
      const double last_tail = __shfl(tail, 63, 64);
      carry_pos = last_inc < 0 ? carry_pos : last_inc;
      carry_acc = last_tail;
      __builtin_amdgcn_wave_barrier();
      // merge: the finished leaves queue up behind `pend` leaves left over from earlier chunks (the queue always starts at a
      // multiple of 64 of the group's leaf sequence); every full block of 64 is one perfect subtree = ONE level-6 push
      {
        const int total = pend + total_new;
        int b = 0;
        for (; b + 64 <= total; b += 64) {
          const double node = wave_tree64(leaves[b + lane]);
          lds_counter_push(csum, cmask, croot, __shfl(node, 0, 64), 6, lane);
        }
        const int rem = total - b;
        double keep = 0.0;
        if (b > 0 && lane < rem) keep = leaves[b + lane];
        __builtin_amdgcn_wave_barrier();
        if (b > 0 && lane < rem) leaves[lane] = keep;
        pend = rem;
      }
      __builtin_amdgcn_wave_barrier();
    }
    // the queue's tail (< 64 leaves, aligned to a multiple of 64): its perfect subtrees, highest first
    if (pend > 0) {
      double x0 = lane < pend ? leaves[lane] : 0.0;
      double x1 = pw_merge(x0, __shfl_down(x0, 1, 64));
      double x2 = pw_merge(x1, __shfl_down(x1, 2, 64));
      double x3 = pw_merge(x2, __shfl_down(x2, 4, 64));
      double x4 = pw_merge(x3, __shfl_down(x3, 8, 64));
      double x5 = pw_merge(x4, __shfl_down(x4, 16, 64));
      for (int sidx = 0; sidx < pend;) {
        int lg = 31 - __builtin_clz((unsigned)(pend - sidx));
        int tz = sidx == 0 ? 6 : __builtin_ctz((unsigned)sidx);
        const int j = tz < lg ? tz : lg;
        double xs = j == 0 ? x0 : j == 1 ? x1 : j == 2 ? x2 : j == 3 ? x3 : j == 4 ? x4 : x5;
        lds_counter_push(csum, cmask, croot, __shfl(xs, sidx, 64), j, lane);
        sidx += 1 << j;
      }
    }
    // the leaf still open at the end of the group
    if (carry_pos > 0) lds_counter_push(csum, cmask, croot, carry_acc, 0, lane);
    double total = 0.0;
    if (nvalid > 0) {
      double a = csum[0];
      for (int i = 1; i <= croot; ++i) a = pw_merge(csum[i], a);
      total = a;
    }
    for (int d = 32; d > 0; d >>= 1) {
      T omin = __shfl_down(ext.vmin, d, 64), omax = __shfl_down(ext.vmax, d, 64);
      long long ormin = __shfl_down(ext.rmin, d, 64), ormax = __shfl_down(ext.rmax, d, 64);
      ext.merge(omin, ormin, omax, ormax);
      isum += __shfl_down(isum, d, 64);
      const long long oz = __shfl_down(zlast, d, 64);
      zlast = oz > zlast ? oz : zlast;
    }
    if (lane == 0) {
      if (out.sum_f) out.sum_f[oi] = total;
      if (out.mean) out.mean[oi] = nvalid ? pw_mean(total, (double)nvalid) : 0.0;
      if (out.sum_i) out.sum_i[oi] = (long long)isum;
      T nanv = T(0);
      if constexpr (__is_same(T, double)) nanv = __builtin_nan("");
      if (out.vmin) static_cast<T*>(out.vmin)[oi] = ext.rmin < 0 ? nanv : ext.vmin;
      if constexpr (__is_same(T, double)) ext.vmax = zero_tie_fix(ext.vmax, zlast, nvalid < len);
      if (out.vmax) static_cast<T*>(out.vmax)[oi] = ext.rmax < 0 ? nanv : ext.vmax;
      if (out.count) out.count[oi] = nvalid;
      ok[oi] = nvalid > 0;
    }
    __builtin_amdgcn_wave_barrier();
  }
}
