// gb_dense_slots.hpp -- part of groupby.hip (textually included there, in its namespace context; split by stage, kernels unchanged):
// dense integer key domain -> slots (speculative single pass, LDS-resident seen bitmap, fused first histogram).
#pragma once

// Dense-domain fast path: when the valid keys span a small integer range the slot is key - min (no table, no probing):
// the only per-row memory access besides the streams is first[slot] (4 B, range-sized table that stays cache resident).
// Residue form (mask != 0): slot = key & mask.  Any window of <= mask + 1 consecutive integers has distinct residues, so this
// is the same perfect hash up to a rotation -- and it needs no minimum, which lets the build run in the SAME pass that
// computes the exact min/max (speculating on the width of the window; verified afterwards).
__device__ __forceinline__ unsigned int dense_slot_of(long long k, long long mn, unsigned int mask) {
  return mask ? ((unsigned int)(unsigned long long)k & mask) : (unsigned int)((unsigned long long)k - (unsigned long long)mn);
}
struct KeyRange {
  long long vmin, vmax;
  int any, pad;
};
// min/max of <= 65536 evenly spaced valid keys (64 workgroups, one sample per thread, one KeyRange per workgroup): the guess for
// the width of the key window
__global__ void __launch_bounds__(1024) k_sample_key_range(const long long* __restrict__ keys, const uint8_t* __restrict__ valid, int64_t off,
                                                           int64_t n, KeyRange* __restrict__ out) {
  __shared__ long long smn[16], smx[16];
  __shared__ int sany[16];
  const int64_t nsamp = n < 65536 ? n : 65536;
  long long mn = 0x7FFFFFFFFFFFFFFFll, mx = (long long)0x8000000000000000ull;
  int any = 0;
  {
    const int64_t j = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    int64_t i = j < nsamp ? (int64_t)((unsigned __int128)j * (unsigned __int128)n / (unsigned __int128)nsamp) : n;
    if (i < n && (!valid || bit_get(valid, off + i))) {
      long long k = keys[i];
      mn = k < mn ? k : mn;
      mx = k > mx ? k : mx;
      any = 1;
    }
  }
  for (int d = 32; d >= 1; d >>= 1) {
    long long a = __shfl_xor(mn, d, 64), b = __shfl_xor(mx, d, 64);
    int c = __shfl_xor(any, d, 64);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
    any |= c;
  }
  if ((threadIdx.x & 63) == 0) {
    smn[threadIdx.x >> 6] = mn;
    smx[threadIdx.x >> 6] = mx;
    sany[threadIdx.x >> 6] = any;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) {
      mn = smn[w] < mn ? smn[w] : mn;
      mx = smx[w] > mx ? smx[w] : mx;
      any |= sany[w];
    }
    out[blockIdx.x].vmin = mn;
    out[blockIdx.x].vmax = mx;
    out[blockIdx.x].any = any;
  }
}
__global__ void __launch_bounds__(256) k_dense_slots(const long long* __restrict__ keys, const uint8_t* __restrict__ valid, int64_t off,
                                                     int64_t n, long long mn, unsigned int mask, unsigned int range, unsigned int* first,
                                                     uint32_t* __restrict__ slot_of_row) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    unsigned int s = range;  // the null key's slot
    if (!valid || bit_get(valid, off + i)) s = dense_slot_of(keys[i], mn, mask);
    slot_of_row[i] = s;
    // only the first lane of every slot present in the wave needs the atomic (lanes hold ascending rows): with a handful of
    // distinct keys tens of thousands of in-flight atomicMin's on one word otherwise serialise in the L2 (measured 5.8 ms)
    const bool want = (unsigned int)i < first[s];
    const int lane = threadIdx.x & 63;
    uint64_t rem = __ballot(want);
    for (int rounds = 0; rem && rounds < 4; ++rounds) {
      const int leader = __ffsll((unsigned long long)rem) - 1;
      const unsigned int sl = (unsigned int)__shfl((int)s, leader, 64);
      const uint64_t grp = __ballot(want && s == sl) & rem;
      if (lane == leader) atomicMin(&first[s], (unsigned int)i);
      rem &= ~grp;
    }
    if ((rem >> lane) & 1) atomicMin(&first[s], (unsigned int)i);
  }
}

// Rows [row0, n) of the dense path once a prefix has been processed by k_dense_slots: `seen` has one bit per slot that already
// has a first row in the prefix.  Every row here is later than every prefix row, so a set bit means "not a first occurrence":
// the common case touches only the (cache-resident) bitmap instead of first[].
__global__ void k_seen_bitmap(const unsigned int* __restrict__ first, int64_t nslots, uint32_t* __restrict__ seen) {
  int64_t nwords = (nslots + 31) >> 5;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += stride) {
    uint32_t bits = 0;
    for (int k = 0; k < 32; ++k) {
      int64_t sl = (w << 5) + k;
      if (sl < nslots && first[sl] != kNoRow) bits |= 1u << k;
    }
    seen[w] = bits;
  }
}
__global__ void __launch_bounds__(256) k_dense_slots_tail(const long long* __restrict__ keys, const uint8_t* __restrict__ valid, int64_t off,
                                                          int64_t row0, int64_t n, long long mn, unsigned int mask, unsigned int range,
                                                          const uint32_t* __restrict__ seen, unsigned int* first,
                                                          uint32_t* __restrict__ slot_of_row) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = row0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // 8 independent key loads in flight per thread: the loop is otherwise latency bound (one 8-byte load per iteration)
  for (; i + 7 * stride < n; i += 8 * stride) {
    long long k[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) k[u] = keys[i + u * stride];
    unsigned int sl[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      sl[u] = range;
      if (!valid || bit_get(valid, off + i + u * stride)) sl[u] = dense_slot_of(k[u], mn, mask);
    }
    uint32_t w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = seen[sl[u] >> 5];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      slot_of_row[i + u * stride] = sl[u];
      if (!((w[u] >> (sl[u] & 31)) & 1u)) {
        unsigned int r = (unsigned int)(i + u * stride);
        if (r < first[sl[u]]) atomicMin(&first[sl[u]], r);
      }
    }
  }
  for (; i < n; i += stride) {
    unsigned int s = range;
    if (!valid || bit_get(valid, off + i)) s = dense_slot_of(keys[i], mn, mask);
    slot_of_row[i] = s;
    if (!((seen[s >> 5] >> (s & 31)) & 1u)) {
      if ((unsigned int)i < first[s]) atomicMin(&first[s], (unsigned int)i);
    }
  }
}

// Tile-shaped variant of the tail (one block = one 4096-row sort tile) that also produces the tile's pass-0 digit histogram, so
// the first pass of every later sort by slot needs no histogram read of slot_of_row.
template <int BITS>
__global__ void __launch_bounds__(kSortBlock) k_dense_slots_tail_hist(const long long* __restrict__ keys, const uint8_t* __restrict__ valid,
                                                                      int64_t off, int64_t tile0, int64_t n, long long mn, unsigned int mask,
                                                                      unsigned int range, const uint32_t* __restrict__ seen, int64_t track_from,
                                                                      unsigned int* first, uint32_t* __restrict__ slot_of_row,
                                                                      uint32_t* __restrict__ hist, long long* __restrict__ tile_min,
                                                                      long long* __restrict__ tile_max) {
  // rows < track_from went through the full first-row protocol already; tile_min / tile_max (optional): key range of the tile
  // (MAX / MIN sentinels when it has no valid key), reduced afterwards -- a shared accumulator would serialise 1e7 atomics
  constexpr int R = 1 << BITS;
  __shared__ uint32_t h[R];
  for (int d = threadIdx.x; d < R; d += kSortBlock) h[d] = 0;
  __syncthreads();
  const int64_t tile = tile0 + blockIdx.x;
  const int64_t base = tile * kSortTile;
  long long k[kSortItems];
#pragma unroll
  for (int u = 0; u < kSortItems; ++u) {
    int64_t i = base + u * kSortBlock + threadIdx.x;
    k[u] = i < n ? keys[i] : 0;
  }
  unsigned int sl[kSortItems];
  uint32_t w[kSortItems];
  long long kmn = 0x7FFFFFFFFFFFFFFFll, kmx = (long long)0x8000000000000000ull;
#pragma unroll
  for (int u = 0; u < kSortItems; ++u) {
    int64_t i = base + u * kSortBlock + threadIdx.x;
    sl[u] = range;
    if (i < n && (!valid || bit_get(valid, off + i))) {
      sl[u] = dense_slot_of(k[u], mn, mask);
      kmn = k[u] < kmn ? k[u] : kmn;
      kmx = k[u] > kmx ? k[u] : kmx;
    }
    w[u] = i < n ? seen[sl[u] >> 5] : ~0u;
  }
#pragma unroll
  for (int u = 0; u < kSortItems; ++u) {
    int64_t i = base + u * kSortBlock + threadIdx.x;
    if (i >= n) continue;
    slot_of_row[i] = sl[u];
    atomicAdd(&h[sl[u] & (R - 1)], 1u);
    if (!((w[u] >> (sl[u] & 31)) & 1u) && i >= track_from) {
      if ((unsigned int)i < first[sl[u]]) atomicMin(&first[sl[u]], (unsigned int)i);
    }
  }
  __shared__ long long smn[kSortWaves], smx[kSortWaves];
  if (tile_min) {
    for (int d = 32; d >= 1; d >>= 1) {
      long long a = __shfl_xor(kmn, d, 64), b = __shfl_xor(kmx, d, 64);
      kmn = a < kmn ? a : kmn;
      kmx = b > kmx ? b : kmx;
    }
    if ((threadIdx.x & 63) == 0) {
      smn[threadIdx.x >> 6] = kmn;
      smx[threadIdx.x >> 6] = kmx;
    }
  }
  __syncthreads();
  if (tile_min && threadIdx.x == 0) {
    for (int w = 1; w < kSortWaves; ++w) {
      kmn = smn[w] < kmn ? smn[w] : kmn;
      kmx = smx[w] > kmx ? smx[w] : kmx;
    }
    tile_min[tile] = kmn;
    tile_max[tile] = kmx;
  }
  for (int d = threadIdx.x; d < R; d += kSortBlock) hist[tile * R + d] = h[d];
}

// Same, for domains of <= 2^20 slots: persistent workgroups (one per CU) keep the whole `seen` bitmap in LDS (128 KB), so the
// per-row bitmap lookup is an LDS read instead of a random TA/L1 access; one wave owns one tile at a time (wave-private histogram).
constexpr int kDenseLdsWords = 32768;
constexpr int kDenseLdsBlock = 1024;
template <int BITS>
__global__ void __launch_bounds__(kDenseLdsBlock) k_dense_slots_tail_hist_lds(const long long* __restrict__ keys, const uint8_t* __restrict__ valid,
                                                                              int64_t off, int64_t tile0, int64_t ntiles, int64_t n, long long mn,
                                                                              unsigned int mask, unsigned int range,
                                                                              const uint32_t* __restrict__ seen, int nwords, int64_t track_from,
                                                                              unsigned int* first, uint32_t* __restrict__ slot_of_row,
                                                                              uint32_t* __restrict__ hist, KeyRange* __restrict__ range_out) {
  // rows < track_from already went through the full first-row protocol (k_dense_slots): here they only get their slot, their
  // histogram count and their share of the min/max.  range_out (optional): one exact KeyRange per workgroup.
  constexpr int R = 1 << BITS;
  constexpr int W = kDenseLdsBlock / 64;
  __shared__ uint32_t lseen[kDenseLdsWords];
  __shared__ uint32_t lh[W][R];
  __shared__ long long smn[W], smx[W];
  __shared__ int sany[W];
  for (int i = threadIdx.x; i < nwords; i += kDenseLdsBlock) lseen[i] = seen[i];
  for (int i = threadIdx.x; i < W * R; i += kDenseLdsBlock) (&lh[0][0])[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t nw = (int64_t)gridDim.x * W;
  long long kmn = 0x7FFFFFFFFFFFFFFFll, kmx = (long long)0x8000000000000000ull;
  int any = 0;
  bool few = true;  // wave-uniform
  for (int64_t tile = tile0 + (int64_t)blockIdx.x * W + wave; tile < ntiles; tile += nw) {
    const int64_t base = tile * kSortTile;
#pragma unroll 1
    for (int c = 0; c < kSortTile / 1024; ++c) {
      long long k[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        int64_t i = base + c * 1024 + u * 64 + lane;
        k[u] = i < n ? keys[i] : 0;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        int64_t i = base + c * 1024 + u * 64 + lane;
        const bool in = i < n;
        unsigned int sl = range;
        if (in && (!valid || bit_get(valid, off + i))) {
          sl = dense_slot_of(k[u], mn, mask);
          kmn = k[u] < kmn ? k[u] : kmn;
          kmx = k[u] > kmx ? k[u] : kmx;
          any = 1;
        }
        const uint32_t w = lseen[sl >> 5];
        if (in) slot_of_row[i] = sl;
        // few distinct digits (few distinct keys): lanes adding to the same LDS word serialise, so peel the rows off digit by
        // digit and add each digit's count once.  `few` is dropped for good the first time a step needs more than 4 rounds (a round costs about as much as a 4-way conflict).
        const unsigned int d = sl & (R - 1);
        if (few) {
          uint64_t rem = __ballot(in);
          int rounds = 0;
          while (rem && rounds < 4) {
            const int leader = __ffsll((unsigned long long)rem) - 1;
            const unsigned int dl = (unsigned int)__shfl((int)d, leader, 64);
            const uint64_t grp = __ballot(in && d == dl) & rem;
            if (lane == leader) atomicAdd(&lh[wave][dl], (uint32_t)__popcll(grp));
            rem &= ~grp;
            ++rounds;
          }
          if (rem) {
            few = false;
            if ((rem >> lane) & 1) atomicAdd(&lh[wave][d], 1u);
          }
        } else if (in) {
          atomicAdd(&lh[wave][d], 1u);
        }
        if (in && !((w >> (sl & 31)) & 1u) && i >= track_from) {
          if ((unsigned int)i < first[sl]) atomicMin(&first[sl], (unsigned int)i);
        }
      }
    }
    for (int d = lane; d < R; d += 64) {
      hist[tile * R + d] = lh[wave][d];
      lh[wave][d] = 0;
    }
  }
  if (range_out) {
    for (int d = 32; d >= 1; d >>= 1) {
      long long a = __shfl_xor(kmn, d, 64), b = __shfl_xor(kmx, d, 64);
      int c = __shfl_xor(any, d, 64);
      kmn = a < kmn ? a : kmn;
      kmx = b > kmx ? b : kmx;
      any |= c;
    }
    if (lane == 0) {
      smn[wave] = kmn;
      smx[wave] = kmx;
      sany[wave] = any;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < W; ++w) {
        kmn = smn[w] < kmn ? smn[w] : kmn;
        kmx = smx[w] > kmx ? smx[w] : kmx;
        any |= sany[w];
      }
      range_out[blockIdx.x].vmin = kmn;
      range_out[blockIdx.x].vmax = kmx;
      range_out[blockIdx.x].any = any;
    }
  }
}
