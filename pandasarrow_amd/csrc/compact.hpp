// compact.hpp -- ordered stream compaction of row indices for gfx950 (wave ballot + mbcnt ranks, block counts, device scan).
// Used by filter (rows selected by a boolean mask), group-by (occupied hash slots) and resample (bin boundaries).
#pragma once
#include "pdx_common.hpp"
#include "scan.hpp"

namespace pdx {

constexpr int kCompactBlock = 256;
constexpr int kCompactItems = 16;  // rows per thread, as 16 wave-steps of 64 consecutive rows
constexpr int kCompactTile = kCompactBlock * kCompactItems;

// Pred: __device__ bool operator()(int64_t i) const   -- i in [0, n)
template <typename Pred>
__global__ void __launch_bounds__(kCompactBlock) k_compact_count(int64_t n, Pred pred, int64_t* __restrict__ block_counts) {
  __shared__ int wave_tot[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t base = (int64_t)blockIdx.x * kCompactTile + wave * (64 * kCompactItems);
  int cnt = 0;
  for (int s = 0; s < kCompactItems; ++s) {
    int64_t i = base + s * 64 + lane;
    bool p = i < n && pred(i);
    cnt += __popcll(__ballot(p));
  }
  if (lane == 0) wave_tot[wave] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = (int64_t)wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
}

// Emit: __device__ void operator()(int64_t out_pos, int64_t i) const
template <typename Pred, typename Emit>
__global__ void __launch_bounds__(kCompactBlock) k_compact_write(int64_t n, Pred pred, Emit emit, const int64_t* __restrict__ block_offsets) {
  __shared__ int wave_tot[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t base = (int64_t)blockIdx.x * kCompactTile + wave * (64 * kCompactItems);
  // pass 1: wave totals (predicates are cheap; re-evaluated below)
  int cnt = 0;
  for (int s = 0; s < kCompactItems; ++s) {
    int64_t i = base + s * 64 + lane;
    bool p = i < n && pred(i);
    cnt += __popcll(__ballot(p));
  }
  if (lane == 0) wave_tot[wave] = cnt;
  __syncthreads();
  int64_t pos = block_offsets[blockIdx.x];
  for (int w = 0; w < wave; ++w) pos += wave_tot[w];
  const uint64_t lt = (1ull << lane) - 1ull;
  for (int s = 0; s < kCompactItems; ++s) {
    int64_t i = base + s * 64 + lane;
    bool p = i < n && pred(i);
    uint64_t b = __ballot(p);
    if (p) emit(pos + __popcll(b & lt), i);
    pos += __popcll(b);
  }
}

// total (host) = number of selected rows; emit is called once per selected row with its output position (ordered).
template <typename Pred, typename Emit>
int compact_indices(int64_t n, Pred pred, Emit emit, int64_t* total_host /* nullptr: the count is not read back, no host wait */, Scratch& s,
                    hipStream_t st) {
  if (total_host) *total_host = 0;
  if (n <= 0) return PDX_OK;
  int64_t nblocks = ceil_div(n, kCompactTile);
  int64_t* counts = s.get<int64_t>((size_t)nblocks);
  int64_t* total = s.get<int64_t>(1);
  PDX_SCRATCH_CHECK(s);
  hipLaunchKernelGGL((k_compact_count<Pred>), dim3((unsigned)nblocks), dim3(kCompactBlock), 0, st, n, pred, counts);
  PDX_TRY((device_exclusive_scan<int64_t, SumOp>(counts, counts, nblocks, total, s, st)));
  hipLaunchKernelGGL((k_compact_write<Pred, Emit>), dim3((unsigned)nblocks), dim3(kCompactBlock), 0, st, n, pred, emit, counts);
  PDX_LAUNCH_CHECK();
  if (!total_host) return PDX_OK;
  PDX_HIP(hipMemcpyAsync(total_host, total, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}

// count only
template <typename Pred>
int count_if(int64_t n, Pred pred, int64_t* total_host, Scratch& s, hipStream_t st) {
  *total_host = 0;
  if (n <= 0) return PDX_OK;
  int64_t nblocks = ceil_div(n, kCompactTile);
  int64_t* counts = s.get<int64_t>((size_t)nblocks);
  int64_t* total = s.get<int64_t>(1);
  PDX_SCRATCH_CHECK(s);
  hipLaunchKernelGGL((k_compact_count<Pred>), dim3((unsigned)nblocks), dim3(kCompactBlock), 0, st, n, pred, counts);
  PDX_TRY((device_exclusive_scan<int64_t, SumOp>(counts, counts, nblocks, total, s, st)));
  PDX_HIP(hipMemcpyAsync(total_host, total, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}

// bytes (0/1 per row) -> Arrow validity bitmap; one thread per output byte.  Returns nothing; null count via count kernels.
__global__ inline void k_pack_bytes(const uint8_t* __restrict__ bytes, int64_t n, uint8_t* __restrict__ bits) {
  int64_t nb = (n + 7) >> 3;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += stride) {
    uint8_t r = 0;
    for (int k = 0; k < 8; ++k) {
      int64_t i = (b << 3) + k;
      if (i < n && bytes[i]) r |= (uint8_t)(1u << k);
    }
    bits[b] = r;
  }
}

}  // namespace pdx
