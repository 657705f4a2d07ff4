// gb_more_aggs.hpp -- part of groupby.hip (textually included there, in its namespace context; split by stage, kernels unchanged):
// variance / stddev / product / first / last on the grouped layout; nullable values in very long groups.
#pragma once

// ---------------------------------------------------------------- "next" aggregations on the grouped layout
__device__ __forceinline__ bool seg_row_is_null(const uint32_t* sorted_keys, const uint8_t* row_valid, int64_t valid_off, int64_t i) {
  if (sorted_keys) return (sorted_keys[i] >> 31) != 0;          // grouped (sorted) layout: the flag travelled with the slot
  return row_valid && !bit_get(row_valid, valid_off + i);       // segments of the original order (resample)
}
// d[i] = (x[i] - mean of x's segment)^2: the second pass of Arrow's variance.  One wave per segment, coalesced.
template <typename T>
__global__ void __launch_bounds__(256) k_seg_sqdev(const T* __restrict__ vals, const uint32_t* __restrict__ seg_start, int64_t nseg,
                                                   const double* __restrict__ mean_seg, double* __restrict__ d,
                                                   const uint32_t* __restrict__ mean_index = nullptr) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); k < nseg; k += nw) {
    const int64_t s = seg_start[k], e = seg_start[k + 1];
    if (s == e) continue;
    const double mu = mean_seg[mean_index ? mean_index[k] : k];  // (mean_index: the means are in group-id order, the segments are not)
    for (int64_t i = s + lane; i < e; i += 64) {
      // NaN operands: x86 SUBSD/MULSD hand back the first NaN operand unchanged, v_add_f64 with a negated source flips its sign;
      // spell the x86 result out so the NaN bits agree too
      const double v = (double)vals[i];
      const double x = v - mu;
      d[i] = v != v ? v : (mu != mu ? mu : x * x);
    }
  }
}
// product of the valid values of every segment in row order (sequential by definition: one multiply chain per group); first /
// last row of every segment.  One wave per segment: 1024 values at a time are loaded coalesced into LDS (null rows as the
// multiplicative identity), lane 0 runs the chain -- the loads, not the chain, bound the kernel.
template <typename T>
__global__ void __launch_bounds__(256) k_seg_product_first_last(const T* __restrict__ vals, const uint32_t* __restrict__ sorted_keys,
                                                                const uint8_t* __restrict__ row_valid, int64_t valid_off,
                                                                const uint32_t* __restrict__ seg_start, int64_t nseg,
                                                                const uint32_t* __restrict__ out_index, T* __restrict__ prod,
                                                                uint8_t* __restrict__ prod_ok, T* __restrict__ first, uint8_t* __restrict__ first_ok,
                                                                T* __restrict__ last, uint8_t* __restrict__ last_ok) {
  __shared__ T stage_all[4][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* stage = stage_all[wave];
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t k = (int64_t)blockIdx.x * 4 + wave; k < nseg; k += nw) {
    const int64_t s = seg_start[k], e = seg_start[k + 1];
    const uint32_t oi = out_index ? out_index[k] : (uint32_t)k;
    if (prod) {
      T p = T(1);
      bool any = false;
      for (int64_t c0 = s; c0 < e; c0 += 1024) {
        const int cl = (int)(e - c0 < 1024 ? e - c0 : 1024);
        bool mine = false;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int idx = q * 64 + lane;
          if (idx < cl) {
            const bool isnull = seg_row_is_null(sorted_keys, row_valid, valid_off, c0 + idx);
            stage[idx] = isnull ? T(1) : vals[c0 + idx];
            mine |= !isnull;
          }
        }
        any |= __any(mine);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
          int i = 0;
          for (; i + 16 <= cl; i += 16) {  // the 16 LDS reads are issued together; only the multiplies form the chain
            T x[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) x[q] = stage[i + q];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              if constexpr (__is_same(T, double)) p = p * x[q];
              else p = (T)((unsigned long long)p * (unsigned long long)x[q]);
            }
          }
          for (; i < cl; ++i) {
            if constexpr (__is_same(T, double)) p = p * stage[i];
            else p = (T)((unsigned long long)p * (unsigned long long)stage[i]);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (lane == 0) {
        prod[oi] = p;
        if (prod_ok) prod_ok[oi] = any;
      }
    }
    if (lane == 0) {
      if (first) {
        first[oi] = e > s ? vals[s] : T(0);
        if (first_ok) first_ok[oi] = e > s && !seg_row_is_null(sorted_keys, row_valid, valid_off, s);
      }
      if (last) {
        last[oi] = e > s ? vals[e - 1] : T(0);
        if (last_ok) last_ok[oi] = e > s && !seg_row_is_null(sorted_keys, row_valid, valid_off, e - 1);
      }
    }
  }
}
// var = m2 / count (ddof = 0); stddev = sqrt(var)
__global__ void k_var_finish(const double* __restrict__ m2, const long long* __restrict__ count, int64_t G, double* __restrict__ var,
                             double* __restrict__ sd) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < G; g += stride) {
    const double v = count[g] > 0 ? m2[g] / (double)count[g] : 0.0;
    if (var) var[g] = v;
    if (sd) sd[g] = v != v ? v : sqrt(v);  // a NaN variance passes through unchanged (x86 sqrtsd keeps the operand's NaN bits)
  }
}

// ---------------------------------------------------------------- nullable values in very long groups.
// One wave per group is hopeless for a group of 1e8 rows with nulls (leaves restart at every run of valid rows, so the work cannot
// be cut into aligned sub-segments the way dense values are).  The grouped values of such a group are one contiguous slice: the
// whole-column kernels (pdx_aggregate: window scan + pairwise tree levels, all workgroups on one slice) reduce it exactly.
struct HugePred {
  const uint32_t* seg_start;
  __device__ bool operator()(int64_t k) const { return (int64_t)seg_start[k + 1] - (int64_t)seg_start[k] > kHugeNullable; }
};
struct HugeEmit {
  const uint32_t* seg_start;
  const uint32_t* out_index;
  int64_t* rec;  // [3 * pos]: start, end, output index
  __device__ void operator()(int64_t pos, int64_t k) const {
    rec[3 * pos] = seg_start[k];
    rec[3 * pos + 1] = seg_start[k + 1];
    rec[3 * pos + 2] = out_index ? out_index[k] : k;
  }
};
// validity bitmap of the grouped layout from the flag bit that travelled with the slots
__global__ void k_flags_to_bitmap(const uint32_t* __restrict__ sorted_keys, int64_t n, uint64_t* __restrict__ words) {
  const int lane = threadIdx.x & 63;
  const int64_t nwords = (n + 63) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < nwords; w += nw) {
    const int64_t i = (w << 6) + lane;
    const uint64_t bal = __ballot(i < n && !(sorted_keys[i] >> 31));
    if (lane == 0) words[w] = bal;
  }
}
static int reduce_huge_nullable_groups(const void* vals, int value_dtype, const uint32_t* sorted_flag_keys, const uint8_t* row_valid, int64_t valid_off,
                                       const uint32_t* seg_start, int64_t nseg, const uint32_t* out_index, int64_t nrows, const SegOut& o, uint8_t* ok,
                                       Scratch& s, hipStream_t st) {
  if (nrows <= kHugeNullable) return PDX_OK;
  const int64_t maxH = nrows / kHugeNullable + 1;
  int64_t* rec = s.get<int64_t>((size_t)3 * (size_t)std::min<int64_t>(nseg, maxH));
  PDX_SCRATCH_CHECK(s);
  int64_t H = 0;
  PDX_TRY(compact_indices(nseg, HugePred{seg_start}, HugeEmit{seg_start, out_index, rec}, &H, s, st));
  if (H == 0) return PDX_OK;
  std::vector<int64_t> hrec((size_t)3 * (size_t)H);
  PDX_HIP(hipMemcpyAsync(hrec.data(), rec, hrec.size() * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  const uint8_t* bitmap = row_valid;
  int64_t bitmap_off = valid_off;
  if (sorted_flag_keys) {
    uint64_t* words = s.get<uint64_t>((size_t)((nrows + 63) >> 6) + 2);
    PDX_SCRATCH_CHECK(s);
    hipLaunchKernelGGL(k_flags_to_bitmap, dim3(grid_for(nrows, 256)), dim3(256), 0, st, sorted_flag_keys, nrows, words);
    PDX_LAUNCH_CHECK();
    bitmap = reinterpret_cast<const uint8_t*>(words);
    bitmap_off = 0;
  }
  PDX_HIP(hipStreamSynchronize(st));
  for (int64_t h = 0; h < H; ++h) {
    const int64_t start = hrec[3 * h], end = hrec[3 * h + 1], oi = hrec[3 * h + 2];
    pdx_column col{};
    col.dtype = value_dtype;
    col.length = end - start;
    col.offset = bitmap_off + start;  // values and bitmap share the element offset: rebase the values pointer instead
    col.null_count = -1;
    col.validity = bitmap;
    col.values = static_cast<const uint8_t*>(vals) - (size_t)bitmap_off * 8;
    auto put = [&](void* dst, const void* src, size_t bytes) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st); };
    pdx_scalar sc{};
    uint8_t valid_group = 0;
    if (o.sum_f || o.mean || o.sum_i) {
      if (o.sum_f || o.sum_i) {
        PDX_TRY(pdx_aggregate(PDX_AGG_SUM, &col, &sc, st));
        valid_group = (uint8_t)sc.is_valid;
        if (o.sum_f) PDX_HIP(put(o.sum_f + oi, &sc.v.f64, 8));
        if (o.sum_i) PDX_HIP(put(o.sum_i + oi, &sc.v.i64, 8));
      }
      if (o.mean) {
        PDX_TRY(pdx_aggregate(PDX_AGG_MEAN, &col, &sc, st));
        valid_group = (uint8_t)sc.is_valid;
        PDX_HIP(put(o.mean + oi, &sc.v.f64, 8));
      }
    }
    if (o.vmin) {
      PDX_TRY(pdx_aggregate(PDX_AGG_MIN, &col, &sc, st));
      valid_group = (uint8_t)sc.is_valid;
      PDX_HIP(put(static_cast<uint8_t*>(o.vmin) + 8 * oi, &sc.v, 8));
    }
    if (o.vmax) {
      PDX_TRY(pdx_aggregate(PDX_AGG_MAX, &col, &sc, st));
      valid_group = (uint8_t)sc.is_valid;
      PDX_HIP(put(static_cast<uint8_t*>(o.vmax) + 8 * oi, &sc.v, 8));
    }
    PDX_TRY(pdx_aggregate(PDX_AGG_COUNT, &col, &sc, st));
    if (!(o.sum_f || o.mean || o.sum_i || o.vmin || o.vmax)) valid_group = sc.v.i64 > 0;
    if (o.count) PDX_HIP(put(o.count + oi, &sc.v.i64, 8));
    if (ok) PDX_HIP(put(ok + oi, &valid_group, 1));
    PDX_HIP(hipStreamSynchronize(st));  // (the staged host scalars above must outlive their copies)
  }
  return PDX_OK;
}
