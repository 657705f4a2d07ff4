// radix_sort.hpp -- stable LSD radix sort of (uint32 key, payload) pairs for gfx950, header-only templates.
//
// Why a sort sits on the group-by path: the reference's per-group fp64 `sum` is Arrow's pairwise tree over the group's
// rows IN ROW ORDER (src/pd_core_macros.h:103 after Grouper::ApplyGroupings, src/dataframe.cpp:1546), so a bit-exact
// device result needs every group's values contiguous and in row order == a stable sort by group slot.
//
// One pass = three kernels, all HBM-streaming:
//   k_radix_hist    : per 4096-row tile, LDS-atomic digit histogram                  (reads 4 B/row)
//   column scan     : hist[tile][digit] -> global output offsets (3 tiny kernels)
//   k_radix_scatter : per tile, stable rank by wave-wide match-any through LDS (wave_match_rank) + per-wave LDS counters,
//                     rows staged in LDS in output order, written back as contiguous runs (reads 12 B, writes 12 B/row;
//                     narrowing passes write (key >> digit bits) in a narrower type)
// Wave = 64 lanes; a tile is 4 waves x 16 steps x 64 rows, so row order == (wave, step, lane) order.
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "pdx_common.hpp"
#include "scan.hpp"

namespace pdx {

constexpr int kSortBlock = 256;
constexpr int kSortItems = 16;
constexpr int kSortTile = kSortBlock * kSortItems;  // 4096 rows
constexpr int kSortWaves = kSortBlock / 64;
constexpr uint32_t kSortKeyMask = 0x7FFFFFFFu;
// the scatter kernel's own workgroup shape over the same 4096-row tile: more waves with fewer rows each hide more latency
constexpr int kScatBlock = 256;
constexpr int kScatItems = kSortTile / kScatBlock;
constexpr int kScatWaves = kScatBlock / 64;      // bit 31 of a key is a caller flag and never sorted on

// One histogram increment per lane -- unless many lanes of the wave hold the same digit (a key with a large share of the rows): LDS
// atomics on one word serialise, so the lanes that share the wave's current CANDIDATE digit are counted by their first lane alone.
// The candidate is the digit of the wave's first lane at the last step where fewer than eight lanes shared it (it settles on a hot
// digit within a few steps and costs a compare + ballot per step otherwise).
__device__ __forceinline__ void wave_hist_add(uint32_t* h, uint32_t d, uint32_t& cand) {
  const unsigned long long m = __ballot(d == cand);
  const int shared = __popcll(m);
  if (shared >= 8) {
    if (d != cand) atomicAdd(&h[d], 1u);
    else if ((int)__lane_id() == __ffsll((long long)m) - 1) atomicAdd(&h[d], (uint32_t)shared);
  } else {
    atomicAdd(&h[d], 1u);
    cand = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
  }
}

template <int BITS, typename K = uint32_t>
__global__ void __launch_bounds__(kSortBlock) k_radix_hist(const K* __restrict__ keys, int64_t n, int shift,
                                                           uint32_t* __restrict__ hist /* [tiles][1<<BITS] */) {
  constexpr int R = 1 << BITS;
  constexpr int NV4 = kSortTile * (int)sizeof(K) / 16;     // 16-byte loads per tile
  constexpr int NV = (NV4 + kSortBlock - 1) / kSortBlock;  // ... per thread (uint32: 4, uint16: 2, uint8: 1 at 4096-row tiles)
  __shared__ uint32_t h[R];
  for (int d = threadIdx.x; d < R; d += kSortBlock) h[d] = 0;
  __syncthreads();
  uint32_t cand = 0xFFFFFFFFu;
  int64_t base = (int64_t)blockIdx.x * kSortTile;
  if (base + kSortTile <= n && (reinterpret_cast<uintptr_t>(keys) & 15) == 0) {
    // full tile: 16-byte loads, all of them in flight before the LDS atomics
    const uint4* k4 = reinterpret_cast<const uint4*>(keys + base);
    uint4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (NV4 % kSortBlock == 0 || k * kSortBlock + (int)threadIdx.x < NV4) v[k] = k4[k * kSortBlock + threadIdx.x];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (!(NV4 % kSortBlock == 0 || k * kSortBlock + (int)threadIdx.x < NV4)) continue;
      const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (sizeof(K) == 4) {
          wave_hist_add(h, (w[j] >> shift) & (R - 1), cand);
        } else if constexpr (sizeof(K) == 2) {
          wave_hist_add(h, ((w[j] & 0xFFFFu) >> shift) & (R - 1), cand);
          wave_hist_add(h, ((w[j] >> 16) >> shift) & (R - 1), cand);
        } else {
#pragma unroll
          for (int b = 0; b < 4; ++b) wave_hist_add(h, (((w[j] >> (8 * b)) & 0xFFu) >> shift) & (R - 1), cand);
        }
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
      int64_t i = base + k * kSortBlock + threadIdx.x;
      if (i < n) wave_hist_add(h, ((uint32_t)keys[i] >> shift) & (R - 1), cand);
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < R; d += kSortBlock) hist[(int64_t)blockIdx.x * R + d] = h[d];
}

// ---- column scan of hist[tiles][R]: afterwards hist[t][d] = global output offset of tile t's first row with digit d
constexpr int kColChunk = 256;  // tiles per chunk
template <int BITS>
__global__ void __launch_bounds__(256) k_col_chunk_sums(const uint32_t* __restrict__ hist, int64_t ntiles,
                                                        uint32_t* __restrict__ chunk_sum /* [chunks][R] */) {
  constexpr int R = 1 << BITS;
  constexpr int T = R >= 256 ? 1 : 256 / R;  // narrow digits: T threads share a digit's column (the tiles of the chunk dealt round-robin)
  int64_t t0 = (int64_t)blockIdx.x * kColChunk;
  int64_t t1 = t0 + kColChunk < ntiles ? t0 + kColChunk : ntiles;
  if constexpr (T == 1) {
    for (int d = threadIdx.x; d < R; d += 256) {
      uint32_t acc = 0;
      for (int64_t t = t0; t < t1; ++t) acc += hist[t * R + d];
      chunk_sum[(int64_t)blockIdx.x * R + d] = acc;
    }
  } else {
    __shared__ uint32_t part[256];
    const int tl = threadIdx.x / R, d = threadIdx.x % R;
    uint32_t acc = 0;
    for (int64_t t = t0 + tl; t < t1; t += T) acc += hist[t * R + d];
    part[threadIdx.x] = acc;
    __syncthreads();
    if (tl == 0) {
#pragma unroll
      for (int j = 1; j < T; ++j) acc += part[j * R + d];
      chunk_sum[(int64_t)blockIdx.x * R + d] = acc;
    }
  }
}
// one workgroup per digit: exclusive scan of the digit's column of chunk sums; digit_total[d] = rows with that digit
template <int BITS>
__global__ void __launch_bounds__(256) k_col_chunk_scan(uint32_t* __restrict__ chunk_sum, int64_t nchunks, uint32_t* __restrict__ digit_total) {
  constexpr int R = 1 << BITS;
  __shared__ uint32_t smem[8];
  const int d = blockIdx.x;
  uint32_t carry = 0;
  for (int64_t c0 = 0; c0 < nchunks; c0 += 256) {
    int64_t c = c0 + threadIdx.x;
    uint32_t v = c < nchunks ? chunk_sum[c * R + d] : 0u;
    uint32_t total;
    uint32_t pre = block_exclusive_scan(v, SumOp(), &total, smem);
    if (c < nchunks) chunk_sum[c * R + d] = carry + pre;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) digit_total[d] = carry;
}
template <int BITS>
__global__ void __launch_bounds__(256) k_col_apply(uint32_t* __restrict__ hist, int64_t ntiles, const uint32_t* __restrict__ chunk_off,
                                                   const uint32_t* __restrict__ digit_total) {
  constexpr int R = 1 << BITS;
  constexpr int DPT = (R + 255) / 256;
  __shared__ uint32_t smem[8];
  __shared__ uint32_t dpre[R];
  {  // exclusive prefix over digits of the digit totals (R <= 2048 words, recomputed per workgroup: cheaper than another launch)
    uint32_t t[DPT], tsum = 0;
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
      int d = threadIdx.x * DPT + j;
      t[j] = d < R ? digit_total[d] : 0u;
      tsum += t[j];
    }
    uint32_t total;
    uint32_t pre = block_exclusive_scan(tsum, SumOp(), &total, smem);
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
      int d = threadIdx.x * DPT + j;
      if (d < R) dpre[d] = pre;
      pre += t[j];
    }
    __syncthreads();
  }
  int64_t t0 = (int64_t)blockIdx.x * kColChunk;
  int64_t t1 = t0 + kColChunk < ntiles ? t0 + kColChunk : ntiles;
  constexpr int T = R >= 256 ? 1 : 256 / R;  // narrow digits: T threads per column, each a contiguous share of the chunk's tiles
  if constexpr (T == 1) {
    for (int d = threadIdx.x; d < R; d += 256) {
      uint32_t off = chunk_off[(int64_t)blockIdx.x * R + d] + dpre[d];
      for (int64_t t = t0; t < t1; ++t) {
        uint32_t v = hist[t * R + d];
        hist[t * R + d] = off;
        off += v;
      }
    }
  } else {
    __shared__ uint32_t part[256];
    const int tl = threadIdx.x / R, d = threadIdx.x % R;
    const int64_t sub = (t1 - t0 + T - 1) / T;
    const int64_t ts = t0 + tl * sub, te = ts + sub < t1 ? ts + sub : t1;
    uint32_t acc = 0;
    for (int64_t t = ts; t < te; ++t) acc += hist[t * R + d];
    part[threadIdx.x] = acc;
    __syncthreads();
    uint32_t off = chunk_off[(int64_t)blockIdx.x * R + d] + dpre[d];
    for (int j = 0; j < tl; ++j) off += part[j * R + d];
    for (int64_t t = ts; t < te; ++t) {
      uint32_t v = hist[t * R + d];
      hist[t * R + d] = off;
      off += v;
    }
  }
}

// Stable rank of a lane's row among the rows of its wave step that carry the same digit, plus the wave's running digit counter.
// Match-any through LDS: every lane ORs its lane bit into the digit's 64-bit word (ds_or_b64) and reads the word back -- the
// lanes with the same digit.  The lowest of them clears the word and advances the counter.  A wave's LDS instructions execute in
// program order, so no barrier is involved.  (The ballot form -- one ballot per digit bit and a per-lane 64-bit select for each --
// cost ~100 vector instructions per step and made both kernels issue bound: SQ_ACTIVE_INST_ANY x waves per SIMD ~ 100 %.)
// match: the wave's [R] words, all zero on entry and again on exit; cnt: the wave's [R] running counters.
__device__ __forceinline__ uint32_t wave_match_rank(unsigned long long* match, uint32_t* cnt, uint32_t d, bool active, int lane, uint64_t lt_mask) {
  if (active) __hip_atomic_fetch_or(&match[d], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const unsigned long long peers = __hip_atomic_load(&match[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  const uint32_t base = __hip_atomic_load(&cnt[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (active && (peers & lt_mask) == 0) {  // the lowest lane of the digit
    __hip_atomic_store(&match[d], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __hip_atomic_store(&cnt[d], base + (uint32_t)__popcll(peers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  return base + (uint32_t)__popcll(peers & lt_mask);
}

// wave_match_rank for the scatter kernels, with the same shortcut as wave_hist_add: the lanes that hold the wave's candidate digit take
// their peers from one ballot instead of the LDS exchange (19 of 64 lanes ORing into one word cost 19 turns).  Ranks are the same.
__device__ __forceinline__ uint32_t wave_match_rank_hot(unsigned long long* match, uint32_t* cnt, uint32_t d, bool active, int lane, uint64_t lt_mask,
                                                        uint32_t& cand) {
  const unsigned long long hot = __ballot(active && d == cand);
  const bool many = __popcll(hot) >= 8;
  const bool mine = many && active && d == cand;
  if (!many) cand = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
  if (active && !mine) __hip_atomic_fetch_or(&match[d], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  unsigned long long peers = __hip_atomic_load(&match[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  const uint32_t base = __hip_atomic_load(&cnt[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (mine) peers = hot;
  if (active && (peers & lt_mask) == 0) {  // the lowest lane of the digit
    if (!mine) __hip_atomic_store(&match[d], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __hip_atomic_store(&cnt[d], base + (uint32_t)__popcll(peers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  return base + (uint32_t)__popcll(peers & lt_mask);
}

// ---- scatter
// IOTA: the payload is not read from memory: it is the row index (with bit 31 set when the row's validity bit is clear)
struct IotaSrc {
  const uint8_t* valid;
  int64_t off;
};
// K: element type of the digit source (uint32 sort keys, or uint8 when the caller kept only the digit itself)
// KO: element type of the keys written; `drop` low key bits are shifted out on the way (narrowing sort: once a digit has been
// sorted on, the later passes no longer need it, so the key shrinks from 4 to 2 to 1 byte as the sort proceeds)
// FLAGS: the top bit of a key (of K and of KO alike) is the row's null flag, carried from type to type; in a FLAGS pass over
// 4-byte keys with iota.valid set, the flag is not in the key yet: it is read from that validity bitmap here (one 64-bit window per
// wave step, loaded by lane `step` and broadcast), which saves the separate pass that used to OR it into the keys.
// EMIT_ROWS (the hash build's partition of the keys by their bucket byte: K = uint8, BITS = 8, shift = 0): beside the payload, every
// row's ORIGINAL index (| bit 31 when its validity bit in iota.valid is clear) is written to rows_out at the same position -- the
// row-id partition that used to be a second scatter kernel with the same ranking and the same digit reads.  The local row (12 bits)
// and the null flag ride in the unused upper bits of the staged digit word, so the kernel needs no extra LDS.
// PAYLOAD_DIGIT (8-byte payloads): there is no key array; the digit is taken from the payload's HIGH 32 bits (at `shift`) -- the
// upper half of a 64-bit sort key rides in the payload above a 32-bit row number (pdx_argsort, align.hip)
// HALF (the narrowing passes of 8-byte payloads, k_radix_scatter_occ4): the payload staging area holds HALF a tile and is filled and
// drained in two rounds, narrow keys are staged in their own width, and the kernel is compiled for 4 waves per SIMD (<= 128 VGPRs):
// 34.5 / 26.5 KB of LDS instead of 51.7 -> four workgroups per CU instead of three.
template <int BITS, typename V, bool WRITE_KEYS, bool IOTA, typename K, typename KO, bool FLAGS, bool EMIT_ROWS, bool PAYLOAD_DIGIT, bool HALF>
__device__ __forceinline__ void radix_scatter_body(const K* __restrict__ keys_in, const V* __restrict__ vals_in, KO* __restrict__ keys_out,
                                                   V* __restrict__ vals_out, int64_t n, int shift, const uint32_t* __restrict__ offsets /* [tiles][R] */,
                                                   int xcd_swizzle, IotaSrc iota, int drop, uint32_t* __restrict__ rows_out) {
  static_assert(!EMIT_ROWS || (sizeof(K) == 1 && BITS <= 8 && !WRITE_KEYS && !IOTA && !FLAGS), "EMIT_ROWS: byte digits, payload + row ids only");
  static_assert(!PAYLOAD_DIGIT || (sizeof(V) == 8 && sizeof(K) == 4 && !WRITE_KEYS && !IOTA && !FLAGS && !EMIT_ROWS), "PAYLOAD_DIGIT: 8-byte payload only");
  constexpr int R = 1 << BITS;
  constexpr int DPT = (R + kScatBlock - 1) / kScatBlock;
  __shared__ uint32_t cnt[kScatWaves][R];   // per-wave digit counters, later per-(wave,digit) local base
  __shared__ uint32_t gbase[R];             // global offset minus local start, per digit
  using SK = typename std::conditional<HALF && sizeof(K) == 2, uint16_t, uint32_t>::type;  // staged key (a 16-bit key needs no more)
  constexpr int kStage = HALF ? kSortTile / 2 : kSortTile;
  __shared__ SK skeys[kSortTile];
  __shared__ V svals[kStage];
  __shared__ uint32_t scan_smem[8];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give every XCD a CONTIGUOUS range of tiles.
  // Neighbouring tiles append to neighbouring addresses of every bucket; written through the same L2 their partial cache
  // lines merge before they leave the XCD (placement only changes speed, never the result).
  const int64_t ntiles_all = (int64_t)gridDim.x;
  int64_t tile = blockIdx.x;
  if (xcd_swizzle) {
    const int64_t per = ntiles_all >> 3;  // tiles per XCD (the remainder keeps its natural index)
    if (tile < per * 8) tile = (tile & 7) * per + (tile >> 3);
  }
  const int64_t tile_base = tile * kSortTile;
  const int tile_rows = (int)((n - tile_base) < kSortTile ? (n - tile_base) : kSortTile);

  // match-any words of the ranking: they live in the value staging area, which is not used before the barrier after the ranking
  // (wide digits with a 4-byte payload: own array)
  constexpr bool kMatchAliased = sizeof(V) * kStage >= sizeof(unsigned long long) * kScatWaves * R;
  __shared__ unsigned long long match_own[kMatchAliased ? 1 : kScatWaves * R];
  unsigned long long* match = kMatchAliased ? reinterpret_cast<unsigned long long*>(svals) : match_own;
  for (int d = tid; d < kScatWaves * R; d += kScatBlock) {
    (&cnt[0][0])[d] = 0;
    match[d] = 0;
  }
  __syncthreads();

  uint32_t key[kScatItems];
  V val[kScatItems];
  uint32_t rank[kScatItems];
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  // validity window of step `lane` (FLAGS pass over 4-byte keys): issued first, one aligned 8-byte load where the bitmap allows
  uint64_t vword = ~0ull;
  if constexpr (FLAGS && !IOTA && sizeof(K) == 4) {
    static_assert(kScatItems <= 64, "one validity window per lane");
    if (iota.valid) {
      const int64_t row0 = tile_base + wave * (64 * kScatItems) + (int64_t)lane * 64;
      if (lane < kScatItems && row0 < n) {
        const int64_t bit0 = iota.off + row0;
        if ((bit0 & 63) == 0 && row0 + 64 <= n && (reinterpret_cast<uintptr_t>(iota.valid) & 7) == 0)
          vword = reinterpret_cast<const uint64_t*>(iota.valid)[bit0 >> 6];
        else
          vword = load_bits64(iota.valid, bit0, iota.off + n);
      }
    }
  }
  // each wave owns rows [wave*1024, wave*1024+1024) of the tile; step s covers 64 consecutive rows
  // payload loads first (independent of everything below), then the digits
#pragma unroll
  for (int s = 0; s < kScatItems; ++s) {
    int r = wave * (64 * kScatItems) + s * 64 + lane;
    if (r < tile_rows) {
      if constexpr (IOTA) {
        int64_t row = tile_base + r;
        val[s] = (V)row | ((iota.valid && !bit_get(iota.valid, iota.off + row)) ? (V)0x80000000u : (V)0);
      } else {
        val[s] = vals_in[tile_base + r];
      }
    }
  }
  // byte digits: one 16-byte load per lane, transposed through the (still unused) key staging area -- sixteen 1-byte loads per
  // lane cost as many memory instructions as sixteen 4-byte ones (measured: 4.3 ms per 1e9 rows for 5 GB moved)
  bool bytes_staged = false;
  if constexpr (sizeof(K) < 4) {
    if (tile_rows == kSortTile && (reinterpret_cast<uintptr_t>(keys_in) & 15) == 0) {
      // a wave's 64 * kScatItems narrow keys = 4 * kScatItems * sizeof(K) lanes x 16 bytes
      constexpr int NV = 4 * kScatItems * (int)sizeof(K);
      for (int j = lane; j < NV; j += 64) {
        const uint4 v = reinterpret_cast<const uint4*>(keys_in + tile_base + wave * (64 * kScatItems))[j];
        reinterpret_cast<uint4*>(skeys)[wave * NV + j] = v;
      }
      __builtin_amdgcn_wave_barrier();
      bytes_staged = true;
    }
  }
#pragma unroll
  for (int s = 0; s < kScatItems; ++s) {
    int r = wave * (64 * kScatItems) + s * 64 + lane;
    if constexpr (PAYLOAD_DIGIT) key[s] = r < tile_rows ? (uint32_t)((unsigned long long)val[s] >> 32) : 0u;
    else if (bytes_staged) key[s] = reinterpret_cast<const K*>(skeys)[r];  // (key r of the tile: waves are laid out back to back)
    else key[s] = r < tile_rows ? (uint32_t)keys_in[tile_base + r] : 0u;
  }
  if constexpr (FLAGS && !IOTA && sizeof(K) == 4) {
    if (iota.valid) {
#pragma unroll
      for (int s = 0; s < kScatItems; ++s) {
        const uint64_t w = __shfl(vword, s, 64);
        key[s] = (key[s] & kSortKeyMask) | (((w >> lane) & 1) ? 0u : 0x80000000u);
      }
    }
  }
  if constexpr (EMIT_ROWS) {
#pragma unroll
    for (int s = 0; s < kScatItems; ++s) {
      const int r = wave * (64 * kScatItems) + s * 64 + lane;
      const bool isnull = iota.valid && r < tile_rows && !bit_get(iota.valid, iota.off + tile_base + r);
      key[s] = (key[s] & 0xFFu) | ((uint32_t)r << 8) | (isnull ? 0x80000000u : 0u);
    }
  }
  uint32_t cand = 0xFFFFFFFFu;
#pragma unroll
  for (int s = 0; s < kScatItems; ++s) {
    int r = wave * (64 * kScatItems) + s * 64 + lane;
    bool active = r < tile_rows;
    uint32_t d = (key[s] >> shift) & (R - 1);
    rank[s] = wave_match_rank_hot(match + wave * R, cnt[wave], d, active, lane, lt_mask, cand);
  }
  __syncthreads();
  // per digit: exclusive prefix over waves, tile totals, exclusive scan over digits
  uint32_t dsum[DPT];
  uint32_t tsum = 0;
#pragma unroll
  for (int j = 0; j < DPT; ++j) {
    int d = tid * DPT + j;
    uint32_t acc = 0;
    if (d < R) {
#pragma unroll
      for (int w = 0; w < kScatWaves; ++w) {
        uint32_t c = cnt[w][d];
        cnt[w][d] = acc;
        acc += c;
      }
    }
    dsum[j] = acc;
    tsum += acc;
  }
  uint32_t total;
  uint32_t pre = block_exclusive_scan(tsum, SumOp(), &total, scan_smem);
#pragma unroll
  for (int j = 0; j < DPT; ++j) {
    int d = tid * DPT + j;
    if (d < R) {
#pragma unroll
      for (int w = 0; w < kScatWaves; ++w) cnt[w][d] += pre;
      gbase[d] = offsets[tile * R + d] - pre;
    }
    pre += dsum[j];
  }
  __syncthreads();
  // stage rows in output order
  if constexpr (!HALF) {
#pragma unroll
    for (int s = 0; s < kScatItems; ++s) {
      int r = wave * (64 * kScatItems) + s * 64 + lane;
      if (r < tile_rows) {
        uint32_t d = (key[s] >> shift) & (R - 1);
        uint32_t p = cnt[wave][d] + rank[s];
        skeys[p] = (SK)key[s];
        svals[p] = val[s];
      }
    }
    __syncthreads();
  }
  auto store = [&](int p, V v) {
    uint32_t k = skeys[p];
    uint32_t d = (k >> shift) & (R - 1);
    uint32_t g = gbase[d] + (uint32_t)p;
    if (WRITE_KEYS) {
      if constexpr (FLAGS) {
        constexpr int FI = 8 * (int)sizeof(K) - 1, FO = 8 * (int)sizeof(KO) - 1;
        keys_out[g] = (KO)(((k & ((1u << FI) - 1u)) >> drop) | ((k >> FI) << FO));
      } else {
        keys_out[g] = (KO)(k >> drop);
      }
    }
    vals_out[g] = v;
    if constexpr (EMIT_ROWS) rows_out[g] = (uint32_t)(tile_base + ((k >> 8) & 0xFFFu)) | (k & 0x80000000u);
  };
  if constexpr (!HALF) {
    // contiguous runs per digit -> coalesced stores
    for (int p = tid; p < tile_rows; p += kScatBlock) store(p, svals[p]);
  } else {
    // keys once, payloads in two rounds through the half-size staging area (output positions [0, kStage) then [kStage, tile))
    uint32_t pos[kScatItems];
#pragma unroll
    for (int s = 0; s < kScatItems; ++s) {
      int r = wave * (64 * kScatItems) + s * 64 + lane;
      pos[s] = 0xFFFFFFFFu;
      if (r < tile_rows) {
        uint32_t d = (key[s] >> shift) & (R - 1);
        pos[s] = cnt[wave][d] + rank[s];
        skeys[pos[s]] = (SK)key[s];
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int s = 0; s < kScatItems; ++s)
        if ((pos[s] >> 11) == (uint32_t)h) svals[pos[s] & (kStage - 1)] = val[s];  // (kStage = 2048: bit 11 of the position picks the round)
      __syncthreads();
      const int lim = tile_rows - h * kStage < kStage ? tile_rows - h * kStage : kStage;
      for (int p = tid; p < lim; p += kScatBlock) store(h * kStage + p, svals[p]);
      if (h == 0) __syncthreads();
    }
  }
}

template <int BITS, typename V, bool WRITE_KEYS, bool IOTA = false, typename K = uint32_t, typename KO = uint32_t, bool FLAGS = false,
          bool EMIT_ROWS = false, bool PAYLOAD_DIGIT = false>
__global__ void __launch_bounds__(kScatBlock) k_radix_scatter(const K* __restrict__ keys_in, const V* __restrict__ vals_in,
                                                              KO* __restrict__ keys_out, V* __restrict__ vals_out, int64_t n,
                                                              int shift, const uint32_t* __restrict__ offsets /* [tiles][R] */,
                                                              int xcd_swizzle, IotaSrc iota = IotaSrc{nullptr, 0}, int drop = 0,
                                                              uint32_t* __restrict__ rows_out = nullptr) {
  radix_scatter_body<BITS, V, WRITE_KEYS, IOTA, K, KO, FLAGS, EMIT_ROWS, PAYLOAD_DIGIT, false>(keys_in, vals_in, keys_out, vals_out, n, shift, offsets,
                                                                                                xcd_swizzle, iota, drop, rows_out);
}
// four waves per SIMD (second launch bound = minimum waves per execution unit) for 8-byte payloads: see HALF above
template <int BITS, typename V, bool WRITE_KEYS, bool IOTA = false, typename K = uint32_t, typename KO = uint32_t, bool FLAGS = false,
          bool EMIT_ROWS = false, bool PAYLOAD_DIGIT = false>
__global__ void __launch_bounds__(kScatBlock, 4) k_radix_scatter_occ4(const K* __restrict__ keys_in, const V* __restrict__ vals_in, KO* __restrict__ keys_out,
                                                                      V* __restrict__ vals_out, int64_t n, int shift, const uint32_t* __restrict__ offsets,
                                                                      int xcd_swizzle, IotaSrc iota = IotaSrc{nullptr, 0}, int drop = 0,
                                                                      uint32_t* __restrict__ rows_out = nullptr) {
  static_assert(kSortTile == 4096 && sizeof(V) == 8 && !IOTA, "half staging: 2 x 2048 positions of an 8-byte payload read from memory");
  radix_scatter_body<BITS, V, WRITE_KEYS, IOTA, K, KO, FLAGS, EMIT_ROWS, PAYLOAD_DIGIT, true>(keys_in, vals_in, keys_out, vals_out, n, shift, offsets, xcd_swizzle,
                                                                                               iota, drop, rows_out);
}
// PDX_SCATTER_OCC4=0 (diagnostic): the 3-workgroups-per-CU form for 8-byte payloads too.  Measured on the headline (1e9 rows, two
// narrowing passes): 9.04-9.11 -> 8.71-8.84 ms for both passes together with the half-staged form (a 36-byte scratch spill per lane included).
inline bool scatter_occ4() {
  static const bool on = [] { const char* e = getenv("PDX_SCATTER_OCC4"); return !(e && e[0] == '0'); }();
  return on;
}
template <int BITS, typename V, bool WRITE_KEYS, bool IOTA = false, typename K = uint32_t, typename KO = uint32_t, bool FLAGS = false, bool EMIT_ROWS = false,
          bool PAYLOAD_DIGIT = false>
inline void launch_radix_scatter(int64_t ntiles, hipStream_t st, const K* kin, const V* vin, KO* kout, V* vout, int64_t n, int shift, const uint32_t* offsets,
                                 int swz, IotaSrc iota = IotaSrc{nullptr, 0}, int drop = 0, uint32_t* rows_out = nullptr) {
  if constexpr (sizeof(V) == 8 && !IOTA && BITS <= 8) {  // (wider digits: the counter arrays alone keep the kernel at 3 workgroups per CU)
    if (scatter_occ4()) {
      hipLaunchKernelGGL((k_radix_scatter_occ4<BITS, V, WRITE_KEYS, IOTA, K, KO, FLAGS, EMIT_ROWS, PAYLOAD_DIGIT>), dim3((unsigned)ntiles), dim3(kScatBlock), 0, st,
                         kin, vin, kout, vout, n, shift, offsets, swz, iota, drop, rows_out);
      return;
    }
  }
  hipLaunchKernelGGL((k_radix_scatter<BITS, V, WRITE_KEYS, IOTA, K, KO, FLAGS, EMIT_ROWS, PAYLOAD_DIGIT>), dim3((unsigned)ntiles), dim3(kScatBlock), 0, st, kin, vin,
                     kout, vout, n, shift, offsets, swz, iota, drop, rows_out);
}

// ---- host driver -----------------------------------------------------------------------------------------------
struct SortPlan {
  int npasses;
  int bits[8];
};
inline SortPlan make_sort_plan(int total_bits, int max_bits_per_pass = 8) {
  SortPlan p;
  if (total_bits < 1) total_bits = 1;
  p.npasses = (total_bits + max_bits_per_pass - 1) / max_bits_per_pass;
  int left = total_bits;
  int first = 0;
  // PDX_SORT_BITS0 (diagnostic): width of the first digit; the other passes share the rest evenly
  if (const char* e = getenv("PDX_SORT_BITS0")) {
    const int b0 = atoi(e), rest = p.npasses - 1;
    if (rest >= 1 && b0 >= 4 && b0 <= max_bits_per_pass && total_bits - b0 >= 4 * rest && total_bits - b0 <= max_bits_per_pass * rest) {
      p.bits[0] = b0;
      left -= b0;
      first = 1;
    }
  }
  for (int i = first; i < p.npasses; ++i) {
    int b = (left + (p.npasses - i) - 1) / (p.npasses - i);
    if (b < 4) b = 4;
    p.bits[i] = b;
    left -= b;
    if (left < 0) left = 0;
  }
  return p;
}

// column scan of raw per-tile histograms: afterwards hist[tile][digit] = output offset of the tile's first row with that digit
template <int BITS>
int radix_scan_only(uint32_t* hist, int64_t ntiles, uint32_t* chunk_sum, bool big, hipStream_t st) {
  int64_t nchunks = ceil_div(ntiles, kColChunk);
  PDX_PROFILE(big ? "radix_scan" : "radix_scan_small", st);
  hipLaunchKernelGGL((k_col_chunk_sums<BITS>), dim3((unsigned)nchunks), dim3(256), 0, st, hist, ntiles, chunk_sum);
  uint32_t* digit_total = chunk_sum + nchunks * ((int64_t)1 << BITS);  // (callers size chunk_sum with one extra row)
  hipLaunchKernelGGL((k_col_chunk_scan<BITS>), dim3(1 << BITS), dim3(256), 0, st, chunk_sum, nchunks, digit_total);
  hipLaunchKernelGGL((k_col_apply<BITS>), dim3((unsigned)nchunks), dim3(256), 0, st, hist, ntiles, chunk_sum, digit_total);
  PDX_LAUNCH_CHECK();
  return PDX_OK;
}
inline int radix_scan_dispatch(int bits, uint32_t* hist, int64_t ntiles, uint32_t* chunk_sum, bool big, hipStream_t st) {
  switch (bits) {
    case 4: return radix_scan_only<4>(hist, ntiles, chunk_sum, big, st);
    case 5: return radix_scan_only<5>(hist, ntiles, chunk_sum, big, st);
    case 6: return radix_scan_only<6>(hist, ntiles, chunk_sum, big, st);
    case 7: return radix_scan_only<7>(hist, ntiles, chunk_sum, big, st);
    case 8: return radix_scan_only<8>(hist, ntiles, chunk_sum, big, st);
    default: return fail(PDX_INVALID, "radix sort: unsupported digit width");
  }
}
// per-tile digit histogram + column scan
template <int BITS, typename K = uint32_t>
int radix_offsets(const K* kin, int64_t n, int shift, uint32_t* hist, uint32_t* chunk_sum, bool big, hipStream_t st) {
  int64_t ntiles = ceil_div(n, kSortTile);
  {
    PDX_PROFILE(big ? "radix_hist" : "radix_hist_small", st);
    hipLaunchKernelGGL((k_radix_hist<BITS, K>), dim3((unsigned)ntiles), dim3(kSortBlock), 0, st, kin, n, shift, hist);
  }
  return radix_scan_only<BITS>(hist, ntiles, chunk_sum, big, st);
}
// histogram of the digit at `shift` of the HIGH 32 bits of 8-byte payloads (see PAYLOAD_DIGIT)
template <int BITS>
__global__ void __launch_bounds__(kSortBlock) k_radix_hist_hi(const uint64_t* __restrict__ vals, int64_t n, int shift, uint32_t* __restrict__ hist) {
  constexpr int R = 1 << BITS;
  __shared__ uint32_t h[R];
  for (int d = threadIdx.x; d < R; d += kSortBlock) h[d] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortTile;
  uint64_t v[kSortItems];
#pragma unroll
  for (int k = 0; k < kSortItems; ++k) {
    const int64_t i = base + k * kSortBlock + threadIdx.x;
    v[k] = i < n ? vals[i] : 0;
  }
#pragma unroll
  for (int k = 0; k < kSortItems; ++k)
    if (base + k * kSortBlock + threadIdx.x < n) atomicAdd(&h[((uint32_t)(v[k] >> 32) >> shift) & (R - 1)], 1u);
  __syncthreads();
  for (int d = threadIdx.x; d < R; d += kSortBlock) hist[(int64_t)blockIdx.x * R + d] = h[d];
}
inline int sort_xcd_swizzle();
// one stable pass of 8-byte payloads by the digit at `shift` of their high 32 bits
template <int BITS>
int radix_pass_payload_hi(const uint64_t* vin, uint64_t* vout, int64_t n, int shift, uint32_t* hist, uint32_t* chunk_sum, hipStream_t st) {
  const int64_t ntiles = ceil_div(n, kSortTile);
  {
    PDX_PROFILE("radix_hist", st);
    hipLaunchKernelGGL((k_radix_hist_hi<BITS>), dim3((unsigned)ntiles), dim3(kSortBlock), 0, st, vin, n, shift, hist);
  }
  PDX_TRY((radix_scan_only<BITS>(hist, ntiles, chunk_sum, true, st)));
  PDX_PROFILE("radix_scatter", st);
  launch_radix_scatter<BITS, uint64_t, false, false, uint32_t, uint32_t, false, false, true>(ntiles, st, (const uint32_t*)nullptr, vin, (uint32_t*)nullptr, vout,
                                                                                           n, shift, hist, sort_xcd_swizzle());
  PDX_LAUNCH_CHECK();
  return PDX_OK;
}
inline int sort_xcd_swizzle() {
  static const int swz = [] { const char* e = getenv("PDX_SORT_XCD_SWIZZLE"); return (e && e[0] == '0') ? 0 : 1; }();
  return swz;
}
// stable scatter of one payload column by the digit, using offsets from radix_offsets (reusable for several payloads)
template <int BITS, typename V, typename K = uint32_t>
int radix_scatter_only(const K* kin, const V* vin, uint32_t* kout, V* vout, int64_t n, int shift, bool write_keys, const uint32_t* hist,
                       hipStream_t st) {
  int64_t ntiles = ceil_div(n, kSortTile);
  PDX_PROFILE(sizeof(V) == 8 ? "radix_scatter" : "radix_scatter_small", st);
  const int swz = sort_xcd_swizzle();
  if (write_keys) launch_radix_scatter<BITS, V, true, false, K>(ntiles, st, kin, vin, kout, vout, n, shift, hist, swz);
  else launch_radix_scatter<BITS, V, false, false, K>(ntiles, st, kin, vin, kout, vout, n, shift, hist, swz);
  PDX_LAUNCH_CHECK();
  return PDX_OK;
}
// narrowing pass (digit = the key's low BITS): the keys written are (key >> BITS) in the narrower type KO
// FLAGS: the keys' top bit is the null flag (see k_radix_scatter); `valid`/`valid_off`: where a pass over 4-byte keys reads it
template <int BITS, typename V, typename K, typename KO, bool FLAGS = false>
int radix_scatter_narrow(const K* kin, const V* vin, KO* kout, V* vout, int64_t n, const uint32_t* offsets, hipStream_t st,
                         const uint8_t* valid = nullptr, int64_t valid_off = 0) {
  int64_t ntiles = ceil_div(n, kSortTile);
  PDX_PROFILE(sizeof(V) == 8 ? "radix_scatter" : "radix_scatter_small", st);
  if (kout) launch_radix_scatter<BITS, V, true, false, K, KO, FLAGS>(ntiles, st, kin, vin, kout, vout, n, 0, offsets, sort_xcd_swizzle(), IotaSrc{valid, valid_off}, BITS);
  else launch_radix_scatter<BITS, V, false, false, K, KO, FLAGS>(ntiles, st, kin, vin, kout, vout, n, 0, offsets, sort_xcd_swizzle(), IotaSrc{valid, valid_off}, BITS);
  PDX_LAUNCH_CHECK();
  return PDX_OK;
}
// Starts of the runs of equal (digit, lower digits) after a stable scatter pass, without a pass over the rows.  The pass's INPUT is
// sorted by the lower digits: combination c of them occupies input rows [prev_start[c], prev_start[c + 1]).  In the output, run
// (d, c) starts at offsets[T][d] + (rows of tile T in front of the boundary that carry digit d), T = the tile holding input row
// prev_start[c] -- one workgroup per c reads at most one tile of the pass's input keys.  out[(d << prev_bits) | c], out[last] = n.
template <typename K>
__global__ void __launch_bounds__(256) k_level_starts(const K* __restrict__ keys_in, int64_t n, const uint32_t* __restrict__ prev_start, int64_t ncombos,
                                                      int prev_bits, int bits, const uint32_t* __restrict__ offsets /* [tiles][1 << bits] */,
                                                      uint32_t* __restrict__ out) {
  __shared__ uint32_t h[256];
  const int R = 1 << bits, tid = threadIdx.x;
  for (int64_t c = blockIdx.x; c < ncombos; c += gridDim.x) {
    h[tid] = 0;
    __syncthreads();
    const int64_t P = prev_start[c];
    if (P >= n) {  // nothing at or after this combination: its runs are empty and start where the next digit's rows begin
      for (int d = tid; d < R; d += 256) out[((int64_t)d << prev_bits) | c] = d + 1 < R ? offsets[d + 1] : (uint32_t)n;
    } else {
      const int64_t T = P / kSortTile;
      const int lim = (int)(P - T * kSortTile);
      for (int i = tid; i < lim; i += 256) atomicAdd(&h[(uint32_t)keys_in[T * kSortTile + i] & (uint32_t)(R - 1)], 1u);
      __syncthreads();
      for (int d = tid; d < R; d += 256) out[((int64_t)d << prev_bits) | c] = offsets[T * R + d] + h[d];
    }
    __syncthreads();
  }
  if (blockIdx.x == 0 && tid == 0) out[(int64_t)R * ncombos] = (uint32_t)n;
}
// the hash build's partition: 8-byte payload (the keys) + the rows' original indexes (| null flag) in one launch
template <int BITS>
int radix_scatter_with_rows(const uint8_t* digits, const uint64_t* vin, uint64_t* vout, uint32_t* rows_out, int64_t n, const uint32_t* hist,
                            const uint8_t* valid, int64_t valid_off, hipStream_t st) {
  static_assert(kSortTile <= 4096, "the local row rides in 12 bits");
  int64_t ntiles = ceil_div(n, kSortTile);
  PDX_PROFILE("radix_scatter", st);
  launch_radix_scatter<BITS, uint64_t, false, false, uint8_t, uint32_t, false, true>(ntiles, st, digits, vin, (uint32_t*)nullptr, vout, n, 0, hist,
                                                                                   sort_xcd_swizzle(), IotaSrc{valid, valid_off}, 0, rows_out);
  PDX_LAUNCH_CHECK();
  return PDX_OK;
}
// payload = row index (| null flag): no payload input stream
template <int BITS, typename K = uint32_t>
int radix_scatter_iota(const K* kin, uint32_t* kout, uint32_t* rows_out, int64_t n, int shift, bool write_keys, const uint32_t* hist,
                       const uint8_t* valid, int64_t valid_off, hipStream_t st) {
  int64_t ntiles = ceil_div(n, kSortTile);
  PDX_PROFILE("radix_scatter_rows", st);
  const int swz = sort_xcd_swizzle();
  if (write_keys)
    hipLaunchKernelGGL((k_radix_scatter<BITS, uint32_t, true, true, K>), dim3((unsigned)ntiles), dim3(kScatBlock), 0, st, kin, (const uint32_t*)nullptr,
                       kout, rows_out, n, shift, hist, swz, IotaSrc{valid, valid_off});
  else
    hipLaunchKernelGGL((k_radix_scatter<BITS, uint32_t, false, true, K>), dim3((unsigned)ntiles), dim3(kScatBlock), 0, st, kin, (const uint32_t*)nullptr,
                       kout, rows_out, n, shift, hist, swz, IotaSrc{valid, valid_off});
  PDX_LAUNCH_CHECK();
  return PDX_OK;
}

template <int BITS, typename V>
int radix_pass(const uint32_t* kin, const V* vin, uint32_t* kout, V* vout, int64_t n, int shift, bool write_keys, uint32_t* hist,
               uint32_t* chunk_sum, hipStream_t st) {
  PDX_TRY((radix_offsets<BITS>(kin, n, shift, hist, chunk_sum, sizeof(V) == 8, st)));
  return radix_scatter_only<BITS, V>(kin, vin, kout, vout, n, shift, write_keys, hist, st);
}

template <typename V>
int radix_pass_dispatch(int bits, const uint32_t* kin, const V* vin, uint32_t* kout, V* vout, int64_t n, int shift, bool write_keys,
                        uint32_t* hist, uint32_t* chunk_sum, hipStream_t st) {
  switch (bits) {
    case 4: return radix_pass<4, V>(kin, vin, kout, vout, n, shift, write_keys, hist, chunk_sum, st);
    case 5: return radix_pass<5, V>(kin, vin, kout, vout, n, shift, write_keys, hist, chunk_sum, st);
    case 6: return radix_pass<6, V>(kin, vin, kout, vout, n, shift, write_keys, hist, chunk_sum, st);
    case 7: return radix_pass<7, V>(kin, vin, kout, vout, n, shift, write_keys, hist, chunk_sum, st);
    case 8: return radix_pass<8, V>(kin, vin, kout, vout, n, shift, write_keys, hist, chunk_sum, st);
    case 9: return radix_pass<9, V>(kin, vin, kout, vout, n, shift, write_keys, hist, chunk_sum, st);
    case 10: return radix_pass<10, V>(kin, vin, kout, vout, n, shift, write_keys, hist, chunk_sum, st);
    default: return fail(PDX_INVALID, "radix sort: unsupported digit width");
  }
}

template <typename V>
int radix_scatter_dispatch(int bits, const uint32_t* kin, const V* vin, uint32_t* kout, V* vout, int64_t n, int shift, bool write_keys,
                           const uint32_t* offsets, hipStream_t st) {
  switch (bits) {
    case 4: return radix_scatter_only<4, V>(kin, vin, kout, vout, n, shift, write_keys, offsets, st);
    case 5: return radix_scatter_only<5, V>(kin, vin, kout, vout, n, shift, write_keys, offsets, st);
    case 6: return radix_scatter_only<6, V>(kin, vin, kout, vout, n, shift, write_keys, offsets, st);
    case 7: return radix_scatter_only<7, V>(kin, vin, kout, vout, n, shift, write_keys, offsets, st);
    case 8: return radix_scatter_only<8, V>(kin, vin, kout, vout, n, shift, write_keys, offsets, st);
    default: return fail(PDX_INVALID, "radix sort: unsupported digit width");
  }
}
inline int sort_max_bits() {
  int max_bits = 8;
  if (const char* e = getenv("PDX_SORT_MAX_BITS")) max_bits = atoi(e) >= 4 && atoi(e) <= 10 ? atoi(e) : 8;
  return max_bits;
}

// Sorts (keys_in, vals_in) by key bits [0, total_bits) into (*keys_sorted, *vals_sorted), which point into the
// caller's ping-pong buffers (k0,v0)/(k1,v1).  Inputs are never written.  n must be < 2^32.
template <typename V>
int radix_sort_pairs(const uint32_t* keys_in, const V* vals_in, uint32_t* k0, V* v0, uint32_t* k1, V* v1, int64_t n, int total_bits,
                     const uint32_t** keys_sorted, const V** vals_sorted, bool need_sorted_keys, Scratch& s, hipStream_t st,
                     int first_shift = 0, const uint32_t* first_pass_offsets = nullptr) {
  // first_pass_offsets: scanned offsets of pass 0 (digit width = make_sort_plan(total_bits, sort_max_bits()).bits[0]) computed by the
  // caller, e.g. fused into the kernel that produced keys_in; pass 0 then needs neither the histogram nor the scan
  int max_bits = sort_max_bits();
  SortPlan plan = make_sort_plan(total_bits, max_bits);
  int64_t ntiles = ceil_div(n, kSortTile);
  int64_t nchunks = ceil_div(ntiles, kColChunk);
  uint32_t* hist = s.get<uint32_t>((size_t)ntiles * ((size_t)1 << max_bits));
  uint32_t* chunk_sum = s.get<uint32_t>((size_t)(nchunks + 1) * ((size_t)1 << max_bits));  // + digit totals row
  PDX_SCRATCH_CHECK(s);
  const uint32_t* kin = keys_in;
  const V* vin = vals_in;
  int shift = first_shift;
  for (int p = 0; p < plan.npasses; ++p) {
    uint32_t* kout = (p & 1) ? k1 : k0;
    V* vout = (p & 1) ? v1 : v0;
    bool last = p == plan.npasses - 1;
    if (p == 0 && first_pass_offsets)
      PDX_TRY(radix_scatter_dispatch<V>(plan.bits[p], kin, vin, kout, vout, n, shift, !last || need_sorted_keys, first_pass_offsets, st));
    else
      PDX_TRY(radix_pass_dispatch<V>(plan.bits[p], kin, vin, kout, vout, n, shift, !last || need_sorted_keys, hist, chunk_sum, st));
    shift += plan.bits[p];
    kin = kout;
    vin = vout;
  }
  *keys_sorted = kin;
  *vals_sorted = vin;
  return PDX_OK;
}

}  // namespace pdx
