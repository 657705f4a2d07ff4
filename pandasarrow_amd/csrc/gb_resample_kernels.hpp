// gb_resample_kernels.hpp -- part of groupby.hip (textually included there, in its namespace context; split by stage, kernels unchanged):
// resample: bin arithmetic on sorted timestamps, non-empty bins, row labels.
#pragma once

// ---------------------------------------------------------------- resample helpers
struct BinParams {
  const long long* ts;
  long long first, freq;
  double inv_freq;  // 1.0 / freq: quotient estimate, corrected exactly below (int64 division is ~100 instructions on CDNA)
  int closed_right;
  __device__ long long bin(int64_t i) const {
    long long x = ts[i] - first - (closed_right ? 1 : 0);  // >= 0: every timestamp is >= first (checked on the host)
    long long q = (long long)((double)x * inv_freq);
    long long r = x - q * freq;
    while (r < 0) { --q; r += freq; }
    while (r >= freq) { ++q; r -= freq; }
    return q;
  }
};
// sparse bins (many rows per bin): bin b starts at the first row whose timestamp is >= (closed-left) / > (closed-right) edge b
// (the search starts from the position a uniformly spaced axis would give and brackets the answer with growing steps before it
//  bisects: a few probes in neighbouring cache lines instead of ~30 scattered ones per edge on regular timestamps)
__global__ void k_bin_lower_bounds(BinParams p, int64_t n, int64_t nbins, long long tmin, long long tmax, uint32_t* __restrict__ lb /* nbins + 1 */) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const double scale = tmax > tmin ? (double)(n - 1) / (double)(tmax - tmin) : 0.0;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= nbins; b += stride) {
    if (b == nbins) {
      lb[b] = (uint32_t)n;
      continue;
    }
    long long edge = p.first + b * p.freq;
    auto before_at = [&](int64_t i) {
      const long long v = p.ts[i];
      return p.closed_right ? (v <= edge) : (v < edge);
    };
    int64_t g = (int64_t)((double)(edge - tmin) * scale);
    g = g < 0 ? 0 : (g > n - 1 ? n - 1 : g);
    int64_t lo = 0, hi = n;
    if (before_at(g)) {  // the answer lies behind g: step forward until a row is not before the edge
      lo = g + 1;
      for (int64_t st = 64; lo + st < n; st <<= 2) {
        if (!before_at(lo + st)) {
          hi = lo + st;
          break;
        }
        lo = lo + st + 1;
      }
    } else {  // the answer is g or in front of it
      hi = g;
      for (int64_t st = 64; hi - st > 0; st <<= 2) {
        if (before_at(hi - st)) {
          lo = hi - st + 1;
          break;
        }
        hi = hi - st;
      }
    }
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      long long v = p.ts[mid];
      bool before = p.closed_right ? (v <= edge) : (v < edge);
      if (before) lo = mid + 1;
      else hi = mid;
    }
    lb[b] = (uint32_t)lo;
  }
}
struct NonEmptyBinPred {
  const uint32_t* lb;
  __device__ bool operator()(int64_t b) const { return lb[b] < lb[b + 1]; }
};
struct NonEmptyBinEmit {
  const uint32_t* lb;
  long long label_base, freq;
  uint32_t* seg_start;
  int64_t* labels;
  int64_t* first_rows;
  __device__ void operator()(int64_t pos, int64_t b) const {
    seg_start[pos] = lb[b];
    labels[pos] = label_base + b * freq;
    first_rows[pos] = (int64_t)lb[b];
  }
};
struct BinStartPred {
  BinParams p;
  __device__ bool operator()(int64_t i) const { return i == 0 || p.bin(i) != p.bin(i - 1); }
};
struct BinStartEmit {
  BinParams p;
  long long label_base;  // first + label_right * freq
  uint32_t* seg_start;
  int64_t* labels;
  int64_t* first_rows;
  __device__ void operator()(int64_t pos, int64_t i) const {
    seg_start[pos] = (uint32_t)i;
    labels[pos] = label_base + p.bin(i) * p.freq;
    first_rows[pos] = i;
  }
};
__global__ void k_check_sorted(const long long* __restrict__ ts, int64_t n, unsigned int* __restrict__ bad) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; i < n; i += stride)
    if (ts[i] < ts[i - 1]) atomicExch(bad, 1u);
}
__global__ void k_row_labels(BinParams p, long long label_base, int64_t n, int64_t* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = label_base + p.bin(i) * p.freq;
}
__global__ void k_seg_row_ids(const uint32_t* __restrict__ seg_start, int64_t G, int64_t n, uint32_t* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int64_t lo = 0, hi = G;  // last segment with start <= i
    while (hi - lo > 1) {
      int64_t mid = (lo + hi) >> 1;
      if (seg_start[mid] <= (uint32_t)i) lo = mid;
      else hi = mid;
    }
    out[i] = (uint32_t)lo;
  }
}
__global__ void k_set_last(uint32_t* p, int64_t idx, uint32_t v) {
  if (threadIdx.x == 0 && blockIdx.x == 0) p[idx] = v;
}

// ---- keys that arrive grouped already (non-decreasing): the groups are the runs of equal keys, in place
// Sampled test (64 x 1024 evenly spaced adjacent pairs): bit 0 = a pair descends in signed order, bit 1 = in unsigned order, bit 2 = a
// null row.  Either order proves that equal keys are neighbours, so a clean bit 0 OR bit 1 (and no null) makes the input a candidate.
__global__ void __launch_bounds__(1024) k_sample_descents(const long long* __restrict__ keys, const uint8_t* __restrict__ valid, int64_t off, int64_t n,
                                                          unsigned int* __restrict__ flags) {
  const int64_t npairs = n - 1;
  const int64_t nsamp = npairs < 65536 ? npairs : 65536;
  const int64_t j = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  if (j >= nsamp) return;
  const int64_t i = (int64_t)((unsigned __int128)j * (unsigned __int128)npairs / (unsigned __int128)nsamp);
  unsigned int f = 0;
  if (valid && (!bit_get(valid, off + i) || !bit_get(valid, off + i + 1))) f |= 4u;
  const long long a = keys[i], b = keys[i + 1];
  if (b < a) f |= 1u;
  if ((unsigned long long)b < (unsigned long long)a) f |= 2u;
  if (f) atomicOr(flags, f);
}
// ---- runs of equal labels over the rows as they stand.  LabelFn: __device__ long long operator()(int64_t row) const.
// Count pass: every lane evaluates its row's label ONCE (the neighbour's comes through a shuffle; lane 0 of a wave's first step
// evaluates the row in front of the wave), the run-start flags of 64 rows go to marks[row >> 6] as one ballot word, descents are
// flagged in both integer orders (flags[0] signed, flags[1] unsigned: either order clean proves that equal labels are neighbours).
// Write pass: reads the n / 8 bytes of marks, never the rows; workgroups without a run start return at once.
constexpr int kRunBlock = 256, kRunSteps = 16, kRunTile = kRunBlock * kRunSteps;  // 4096 rows per workgroup, 1024 per wave
constexpr int kRunWaveRows = 64 * kRunSteps;                                        // the unit of the counts / offsets: one wave's rows
// BATCH: labels requested before the first of them is used -- 16 for plain keys (one dependent load per step left the kernel at
// 4.4 TB/s), 4 where the label costs registers to compute (the calendar rounding: 16 at once was slower, 5.1 vs 4.7 ms end to end)
template <typename LabelFn, int BATCH>
__global__ void __launch_bounds__(kRunBlock) k_label_run_count(int64_t n, LabelFn fn, unsigned long long* __restrict__ marks,
                                                                int64_t* __restrict__ wave_counts, unsigned int* __restrict__ flags) {
  static_assert(kRunSteps % BATCH == 0, "whole batches");
  const int lane = threadIdx.x & 63;
  const int64_t w = ((int64_t)blockIdx.x * kRunBlock + threadIdx.x) >> 6;  // wave = 1024-row slice
  const int64_t base = w * kRunWaveRows;
  if (base >= n) return;  // (whole waves leave: nothing below is a workgroup barrier)
  long long last = 0;
  if (lane == 0 && base > 0) last = fn(base - 1);
  int cnt = 0;
  bool desc_s = false, desc_u = false;
  auto step = [&](int s, long long label) {
    const int64_t i = base + (int64_t)s * 64 + lane;
    const bool act = i < n;
    long long prev = __shfl_up(label, 1, 64);
    if (lane == 0) prev = last;
    const bool start = act && (i == 0 || label != prev);
    if (act && i > 0) {
      desc_s |= label < prev;
      desc_u |= (unsigned long long)label < (unsigned long long)prev;
    }
    const unsigned long long b = __ballot(start);
    if (lane == 0 && base + (int64_t)s * 64 < n) marks[(base >> 6) + s] = b;
    cnt += __popcll(b);
    last = __shfl(label, 63, 64);
  };
  if constexpr (BATCH == kRunSteps) {
    long long label[kRunSteps];
#pragma unroll
    for (int s = 0; s < kRunSteps; ++s) {
      const int64_t i = base + (int64_t)s * 64 + lane;
      label[s] = i < n ? fn(i) : 0;
    }
#pragma unroll
    for (int s = 0; s < kRunSteps; ++s) step(s, label[s]);
  } else {
#pragma unroll BATCH
    for (int s = 0; s < kRunSteps; ++s) {
      const int64_t i = base + (int64_t)s * 64 + lane;
      step(s, i < n ? fn(i) : 0);
    }
  }
  if (desc_s) flags[0] = 1u;
  if (desc_u) flags[1] = 1u;
  if (lane == 0) wave_counts[w] = cnt;
}
// Emit: __device__ void operator()(int64_t run, int64_t row) const.  One wave per 1024-row slice, no workgroup state.
template <typename Emit>
__global__ void __launch_bounds__(kRunBlock) k_label_run_write(int64_t n, const unsigned long long* __restrict__ marks, Emit emit,
                                                                const int64_t* __restrict__ wave_offsets, int64_t nwaves,
                                                                const int64_t* __restrict__ total) {
  const int lane = threadIdx.x & 63;
  const int64_t w = ((int64_t)blockIdx.x * kRunBlock + threadIdx.x) >> 6;
  if (w >= nwaves) return;
  int64_t pos = wave_offsets[w];
  const int64_t pos1 = w + 1 < nwaves ? wave_offsets[w + 1] : *total;
  if (pos == pos1) return;  // (uniform per wave)
  const int64_t base = w * kRunWaveRows;
  unsigned long long word = 0;
  if (lane < kRunSteps && base + (int64_t)lane * 64 < n) word = marks[(base >> 6) + lane];
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int s = 0; s < kRunSteps; ++s) {
    const unsigned long long b = __shfl(word, s, 64);
    if ((b >> lane) & 1ull) emit(pos + __popcll(b & lt), base + (int64_t)s * 64 + lane);
    pos += __popcll(b);
  }
}
struct KeyLabel {
  static constexpr int kBatch = 16;
  const long long* keys;
  __device__ long long operator()(int64_t i) const { return keys[i]; }
};
// the run's label is recomputed from its first row; `shift` is added on the way out (downsample's "one day less" for M / W / Q rules)
template <typename LabelFn>
struct RunEmit {
  LabelFn fn;
  long long shift;
  uint32_t* seg_start;
  int64_t* uniques;
  int64_t* first_rows;
  uint32_t* gid_of_occ;  // identity: the runs are the groups, in first-occurrence order
  __device__ void operator()(int64_t pos, int64_t i) const {
    seg_start[pos] = (uint32_t)i;
    uniques[pos] = (long long)((unsigned long long)fn(i) + (unsigned long long)shift);
    first_rows[pos] = i;
    gid_of_occ[pos] = (uint32_t)pos;
  }
};
