// select.hip -- boolean-mask filter, take (gather) and row concat for gfx950.
//
// Replaces (reference file:line)
//   DataFrame::where / Series::where : CallFunction("filter"/"array_filter", {.., mask}, FilterOptions{EMIT_NULL})
//        src/dataframe.cpp:461-475, src/series.cpp:130-144, reached from NDFrame::operator[](Series) src/ndframe.cpp:347-350
//   DataFrame::take / Series::take   : CallFunction("take"/"array_take")   src/dataframe.cpp:477-492, src/series.cpp:146-159
//   pd::concat rows                  : arrow::ConcatenateTables + CombineChunksToBatch   src/concat.cpp:152-154
//
// filter = ordered compaction of the selected row ids (wave ballot + popcount ranks, compact.hpp) followed by ONE fused
// gather over all columns (the index list is read once for up to 16 columns; source reads are monotonic so every cache
// line is fetched once).  take is the same gather with caller indices plus a bounds check that reports through a device
// flag.  Output validity words are assembled with wave ballots: a wave owns 64 consecutive output rows.
// Algorithmic bytes: filter (1/8 + 8(C+1)(1+s)) B/row, take (8 + 16(C+1)) B per output row.
#include "compact.hpp"

namespace pdx {

constexpr int kMaxCols = 16;
struct GatherCols {
  const uint64_t* src[kMaxCols];      // values + offset applied
  const uint8_t* src_valid[kMaxCols]; // or nullptr
  int64_t src_off[kMaxCols];
  uint64_t* dst[kMaxCols];
  uint8_t* dst_valid[kMaxCols];       // or nullptr
  int ncols;
};

// idx[j] < 0  => emit a null row.  idx_valid (bitmap, optional) marks null indices.  Bounds are checked against n_src.
// A wave owns kGatherU consecutive 64-row output words per iteration and issues the index loads of all of them, then per
// column the kGatherU gathers, before anything is consumed (memory-level parallelism; the loop is latency bound otherwise).
constexpr int kGatherU = 4;
constexpr int kGatherCols = 4;
template <typename IDX>
__global__ void __launch_bounds__(256) k_gather(GatherCols c, const IDX* __restrict__ idx, const uint8_t* __restrict__ idx_valid,
                                                int64_t idx_off, int64_t m, int64_t n_src, int check_bounds, ErrFlag* err,
                                                unsigned long long* __restrict__ null_counts) {
  const int lane = threadIdx.x & 63;
  // null rows per column: summed per workgroup in LDS, one global add per workgroup and column at the end (an add per 64-row word
  // on neighbouring global counters serialises in one L2 channel: a nullable 1e8-row filter took 73 ms that way)
  __shared__ unsigned int snulls[kMaxCols];
  if (threadIdx.x < kMaxCols) snulls[threadIdx.x] = 0;
  __syncthreads();
  int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int64_t nwords = (m + 63) >> 6;
  int64_t ngroups = (nwords + kGatherU - 1) / kGatherU;
  for (int64_t g = wave; g < ngroups; g += nwaves) {
    long long k[kGatherU];
    bool in[kGatherU], have[kGatherU];
#pragma unroll
    for (int u = 0; u < kGatherU; ++u) {
      int64_t j = ((g * kGatherU + u) << 6) + lane;
      in[u] = j < m;
      have[u] = in[u] && (!idx_valid || bit_get(idx_valid, idx_off + j));
      k[u] = have[u] ? (long long)idx[j] : -1;
    }
    if (check_bounds) {
#pragma unroll
      for (int u = 0; u < kGatherU; ++u)
        if (have[u] && (k[u] < 0 || k[u] >= n_src)) {
          atomicMax(&err->code, 1ull);
          err->payload = k[u];
          k[u] = -1;
        }
    }
    // columns in batches of kGatherCols: kGatherCols x kGatherU independent random loads are issued before any is consumed
    for (int col0 = 0; col0 < c.ncols; col0 += kGatherCols) {
      uint64_t v[kGatherCols][kGatherU];
      bool ok[kGatherCols][kGatherU];
#pragma unroll
      for (int cc = 0; cc < kGatherCols; ++cc) {
        const int col = col0 + cc;
#pragma unroll
        for (int u = 0; u < kGatherU; ++u) {
          ok[cc][u] = col < c.ncols && k[u] >= 0 && (!c.src_valid[col] || bit_get(c.src_valid[col], c.src_off[col] + k[u]));
          v[cc][u] = ok[cc][u] ? c.src[col][k[u]] : 0ull;
        }
      }
#pragma unroll
      for (int cc = 0; cc < kGatherCols; ++cc) {
        const int col = col0 + cc;
        if (col >= c.ncols) break;
#pragma unroll
        for (int u = 0; u < kGatherU; ++u) {
          const int64_t w = g * kGatherU + u;
          if (in[u]) c.dst[col][(w << 6) + lane] = v[cc][u];
          if (c.dst_valid[col]) {
            uint64_t bal = __ballot(ok[cc][u]);
            if (lane == 0 && w < nwords) {
              int64_t remain = m - (w << 6);
              if (remain >= 64) reinterpret_cast<uint64_t*>(c.dst_valid[col])[w] = bal;
              else {
                int nbytes = (int)((remain + 7) >> 3);
                for (int q = 0; q < nbytes; ++q) c.dst_valid[col][(w << 3) + q] = (uint8_t)(bal >> (8 * q));
              }
              int valid_rows = (int)(remain >= 64 ? 64 : remain);
              unsigned long long nulls = (unsigned long long)(valid_rows - __popcll(bal));
              if (nulls) atomicAdd(&snulls[col], (unsigned int)nulls);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < c.ncols && snulls[threadIdx.x]) atomicAdd(&null_counts[threadIdx.x], (unsigned long long)snulls[threadIdx.x]);
}

// scatter: dst[idx[j]] = src[j]; validity bits are set/cleared with atomics (distinct rows may share a byte)
__global__ void __launch_bounds__(256) k_scatter(GatherCols c, const int64_t* __restrict__ idx, int64_t m, int64_t n_dst, ErrFlag* err) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
    long long k = idx[j];
    if (k < 0 || k >= n_dst) {
      atomicMax(&err->code, 1ull);
      err->payload = k;
      continue;
    }
    for (int col = 0; col < c.ncols; ++col) {
      bool ok = !c.src_valid[col] || bit_get(c.src_valid[col], c.src_off[col] + j);
      c.dst[col][k] = c.src[col][j];
      if (c.dst_valid[col]) {
        unsigned int* word = reinterpret_cast<unsigned int*>(c.dst_valid[col]) + (k >> 5);
        unsigned int bit = 1u << (k & 31);
        if (ok) atomicOr(word, bit);
        else atomicAnd(word, ~bit);
      }
    }
  }
}

struct MaskPred {
  const uint8_t* mask;
  const uint8_t* mvalid;
  int64_t off;
  int emit_null;
  __device__ bool operator()(int64_t i) const {
    bool mv = !mvalid || bit_get(mvalid, off + i);
    return mv ? bit_get(mask, off + i) : (emit_null != 0);
  }
};
struct MaskEmit {
  const uint8_t* mvalid;
  int64_t off;
  int64_t* sel;
  __device__ void operator()(int64_t pos, int64_t i) const {
    bool mv = !mvalid || bit_get(mvalid, off + i);
    sel[pos] = mv ? i : -1;  // a null mask slot emits a null row (FilterOptions::EMIT_NULL)
  }
};

// Streaming filter for columns without validity (and a mask whose null slots are dropped or absent): instead of gathering by
// the compacted row ids, every column is read ONCE with coalesced 8-byte loads and the selected values are written in order
// (within a wave step the selected lanes store to consecutive addresses; consecutive steps continue where the last one ended).
// Same tile / wave layout as k_compact_count, whose scanned block counts give every workgroup its output offset.  A wave owns
// 1024 consecutive rows: lane s < 16 loads the 64 selection bits of step s, the words are then broadcast (uniform registers).
// NULLS: columns with validity and / or an EMIT_NULL mask with null slots (a null slot selects the row and makes it null in every
// column).  The output validity bitmaps are preset to all ones by the host; only the selected rows that ARE null clear their
// bit (one atomicAnd on the 32-bit word each: nulls are the rare case), and the wave adds its null count per column.
constexpr int kFilterCols = 2;  // columns per batch: 2 x 16 independent 8-byte loads per lane in flight
template <bool NULLS>
__global__ void __launch_bounds__(kCompactBlock) k_filter_stream(GatherCols c, const uint8_t* __restrict__ mask, const uint8_t* __restrict__ mvalid,
                                                                 int64_t off, int64_t n, const int64_t* __restrict__ block_offsets, int emit_null,
                                                                 unsigned int* __restrict__ block_nulls /* [ncols][blocks] */) {
  __shared__ int wave_tot[4];
  // null rows per column of this workgroup: summed in LDS and written once (an atomicAdd per wave and column on nine neighbouring
  // global counters serialised in one L2 channel: 8 of the kernel's 10 ms)
  __shared__ unsigned int snulls[kMaxCols];
  if (NULLS && threadIdx.x < kMaxCols) snulls[threadIdx.x] = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * kCompactTile + wave * (64 * kCompactItems);
  uint64_t mine = 0, mine_forced = 0;
  if (lane < kCompactItems) {
    const int64_t i0 = base + (int64_t)lane * 64;
    if (i0 < n) {
      mine = load_bits64(mask, off + i0, off + n);
      if (mvalid) {
        const uint64_t mv = load_bits64(mvalid, off + i0, off + n);
        if (NULLS && emit_null) {
          const int64_t remain = n - i0;
          const uint64_t inr = remain >= 64 ? ~0ull : ((1ull << remain) - 1ull);
          mine_forced = ~mv & inr;            // EMIT_NULL: a null mask slot selects the row and nulls it
          mine = (mine & mv) | mine_forced;
        } else {
          mine &= mv;                          // DROP: a null mask slot selects nothing
        }
      }
    }
  }
  uint64_t sel[kCompactItems];
  int cnt = 0;
#pragma unroll
  for (int s = 0; s < kCompactItems; ++s) {
    sel[s] = __shfl(mine, s, 64);
    cnt += __popcll(sel[s]);
  }
  if (lane == 0) wave_tot[wave] = cnt;
  __syncthreads();
  int64_t pos0 = block_offsets[blockIdx.x];
  for (int w = 0; w < wave; ++w) pos0 += wave_tot[w];
  const uint64_t lt = (1ull << lane) - 1ull;
  for (int col0 = 0; col0 < c.ncols; col0 += kFilterCols) {
    uint64_t v[kFilterCols][kCompactItems];
#pragma unroll
    for (int cc = 0; cc < kFilterCols; ++cc) {
      const int col = col0 + cc;
#pragma unroll
      for (int s = 0; s < kCompactItems; ++s) {
        const bool p = ((sel[s] >> lane) & 1) && col < c.ncols;
        v[cc][s] = p ? c.src[col][base + s * 64 + lane] : 0ull;
      }
    }
#pragma unroll
    for (int cc = 0; cc < kFilterCols; ++cc) {
      const int col = col0 + cc;
      if (col >= c.ncols) break;
      int64_t pos = pos0;
      if constexpr (NULLS) {
        // validity window of step `lane` of this column (all ones without a bitmap), minus the rows the mask forces to null
        uint64_t okw = ~0ull;
        if (lane < kCompactItems) {
          const int64_t i0 = base + (int64_t)lane * 64;
          if (c.src_valid[col] && i0 < n) okw = load_bits64(c.src_valid[col], c.src_off[col] + i0, c.src_off[col] + n);
          okw &= ~mine_forced;
        }
        uint32_t* dstw = reinterpret_cast<uint32_t*>(c.dst_valid[col]);
        int wave_nulls = 0;
        // the wave's selected rows occupy <= 1024 consecutive output bits = <= 33 words: lane w keeps the clear mask of word w and
        // the wave sends ONE atomicAnd per word at the end (an atomic per null row cost 8 ms of 10 at 5 % nulls).  Which rows are null
        // is wave-uniform knowledge (64-bit words), so the bit positions come from a scalar loop over the null bits.
        uint32_t myw = 0xFFFFFFFFu;
        const int64_t wbase = pos0 & ~31ll;
#pragma unroll
        for (int s = 0; s < kCompactItems; ++s) {
          const uint64_t ok = __shfl(okw, s, 64);
          const uint64_t nulls_v = sel[s] & ~ok;
          const uint64_t nulls = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(nulls_v >> 32)) << 32) |
                                 (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)nulls_v);
          if ((sel[s] >> lane) & 1) {
            const int64_t o = pos + __popcll(sel[s] & lt);
            c.dst[col][o] = ((nulls >> lane) & 1) ? 0ull : v[cc][s];
          }
          for (uint64_t rem = nulls; rem; rem &= rem - 1) {
            const int b = __ffsll((unsigned long long)rem) - 1;
            const int orel = (int)(pos - wbase) + __popcll(sel[s] & ((1ull << b) - 1ull));
            if (lane == (orel >> 5)) myw &= ~(1u << (orel & 31));
          }
          wave_nulls += __popcll(nulls);
          pos += __popcll(sel[s]);
        }
        if (myw != 0xFFFFFFFFu && dstw) atomicAnd(&dstw[(wbase >> 5) + lane], myw);
        if (lane == 0 && wave_nulls) atomicAdd(&snulls[col], (unsigned int)wave_nulls);
      } else {
#pragma unroll
        for (int s = 0; s < kCompactItems; ++s) {
          if ((sel[s] >> lane) & 1) c.dst[col][pos + __popcll(sel[s] & lt)] = v[cc][s];
          pos += __popcll(sel[s]);
        }
      }
    }
  }
  if constexpr (NULLS) {
    __syncthreads();
    if ((int)threadIdx.x < c.ncols) block_nulls[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = snulls[threadIdx.x];
  }
}
__global__ void __launch_bounds__(256) k_sum_block_nulls(const unsigned int* __restrict__ block_nulls, int64_t nblocks, unsigned long long* __restrict__ out) {
  __shared__ unsigned long long part[4];
  unsigned long long acc = 0;
  for (int64_t i = threadIdx.x; i < nblocks; i += 256) acc += block_nulls[(int64_t)blockIdx.x * nblocks + i];
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

static int fill_cols(GatherCols& g, const pdx_column* cols, int ncols, pdx_mut_column* outs, int64_t out_len, int64_t src_len,
                     bool forced_nulls, const char* what) {
  if (ncols < 1 || ncols > kMaxCols) return fail(PDX_INVALID, std::string(what) + ": between 1 and 16 columns per call");
  g.ncols = ncols;
  for (int c = 0; c < ncols; ++c) {
    PDX_TRY(check_column(&cols[c], what));
    if (cols[c].dtype == PDX_BOOL) return fail(PDX_NOT_IMPLEMENTED, std::string(what) + ": boolean columns are not supported yet");
    if (cols[c].length != src_len) return fail(PDX_INVALID, std::string(what) + ": all columns must have the same length");
    if (outs[c].length < out_len) return fail(PDX_INVALID, std::string(what) + ": output too small");
    if (outs[c].dtype != cols[c].dtype) return fail(PDX_INVALID, std::string(what) + ": output dtype must equal the column dtype");
    if (out_len && !outs[c].values) return fail(PDX_INVALID, std::string(what) + ": null output buffer");
    const uint8_t* sv = validity_or_null(&cols[c]);
    if ((sv || forced_nulls) && !outs[c].validity) return fail(PDX_INVALID, std::string(what) + ": nulls possible but an output has no validity buffer");
    g.src[c] = static_cast<const uint64_t*>(cols[c].values) + cols[c].offset;
    g.src_valid[c] = sv;
    g.src_off[c] = cols[c].offset;
    g.dst[c] = static_cast<uint64_t*>(outs[c].values);
    g.dst_valid[c] = static_cast<uint8_t*>(outs[c].validity);
  }
  return PDX_OK;
}

template <typename IDX>
static int run_gather(GatherCols& g, const IDX* idx, const uint8_t* idx_valid, int64_t idx_off, int64_t m, int64_t n_src, int check,
                      pdx_mut_column* outs, Scratch& s, hipStream_t st, long long* bad_index) {
  ErrFlag* err = s.get<ErrFlag>(1);
  unsigned long long* nulls = s.get<unsigned long long>(kMaxCols);
  PDX_SCRATCH_CHECK(s);
  PDX_HIP(hipMemsetAsync(err, 0, sizeof(ErrFlag), st));
  PDX_HIP(hipMemsetAsync(nulls, 0, sizeof(unsigned long long) * kMaxCols, st));
  if (m > 0) {
    int64_t nwords = (m + 63) >> 6;
    // (109 VGPRs: 4 workgroups are resident per CU; 4 .. 32 per CU measured 11.0 .. 10.7 ms for the 4.5e8 gathers of the C2 take: the
    //  memory system's random-access rate, not the launch shape, bounds this kernel)
    static const int gather_wgs_per_cu = [] { const char* e = getenv("PDX_GATHER_WGS_PER_CU"); return e && atoi(e) > 0 ? atoi(e) : 16; }();
    hipLaunchKernelGGL((k_gather<IDX>), dim3(grid_for(ceil_div(nwords, kGatherU) * 64, 256, 1, kCUs * gather_wgs_per_cu)), dim3(256), 0, st, g, idx, idx_valid,
                       idx_off, m, n_src, check, err, nulls);
    PDX_LAUNCH_CHECK();
  }
  ErrFlag h;
  unsigned long long hn[kMaxCols];
  PDX_HIP(hipMemcpyAsync(&h, err, sizeof(h), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipMemcpyAsync(hn, nulls, sizeof(hn), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
  if (h.code) {
    *bad_index = h.payload;
    return PDX_INDEX_ERROR;
  }
  for (int c = 0; c < g.ncols; ++c) {
    outs[c].length = m;
    outs[c].null_count = g.dst_valid[c] ? (int64_t)hn[c] : 0;
  }
  return PDX_OK;
}

// ---------------------------------------------------------------- concat validity: one thread per output word
struct ConcatParts {
  const uint8_t* valid[64];
  int64_t off[64];
  int64_t start[65];  // output row where each part begins; start[nparts] = total
  int nparts;
};
__global__ void k_concat_validity(ConcatParts p, int64_t total, uint8_t* __restrict__ out, unsigned long long* __restrict__ nulls) {
  int64_t nwords = (total + 63) >> 6;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long nc = 0;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += stride) {
    int64_t base = w << 6;
    int64_t end = base + 64 < total ? base + 64 : total;
    uint64_t r = 0;
    for (int q = 0; q < p.nparts; ++q) {
      int64_t lo = p.start[q] > base ? p.start[q] : base;
      int64_t hi = p.start[q + 1] < end ? p.start[q + 1] : end;
      if (lo >= hi) continue;
      int cnt = (int)(hi - lo);
      uint64_t bits = ~0ull;
      if (p.valid[q]) bits = load_bits64(p.valid[q], p.off[q] + (lo - p.start[q]), p.off[q] + (p.start[q + 1] - p.start[q]));
      if (cnt < 64) bits &= (1ull << cnt) - 1ull;
      r |= bits << (lo - base);
    }
    int rows = (int)(end - base);
    nc += (unsigned long long)(rows - __popcll(r));
    if (rows == 64) reinterpret_cast<uint64_t*>(out)[w] = r;
    else {
      int nbytes = (rows + 7) >> 3;
      for (int k = 0; k < nbytes; ++k) out[(w << 3) + k] = (uint8_t)(r >> (8 * k));
    }
  }
  for (int d = 32; d > 0; d >>= 1) nc += __shfl_down(nc, d, 64);
  if ((threadIdx.x & 63) == 0 && nc) atomicAdd(nulls, nc);
}

}  // namespace pdx

using namespace pdx;

extern "C" {

int pdx_filter_count(const pdx_column* mask, int emit_null, int64_t* out_count, void* stream) {
  PDX_TRY(check_column(mask, "pdx_filter_count"));
  if (mask->dtype != PDX_BOOL) return fail(PDX_INVALID, "filter mask must be boolean");
  if (!out_count) return fail(PDX_INVALID, "pdx_filter_count: null output");
  Scratch s;
  MaskPred pred{static_cast<const uint8_t*>(mask->values), validity_or_null(mask), mask->offset, emit_null};
  return count_if(mask->length, pred, out_count, s, as_stream(stream));
}

int pdx_filter(const pdx_column* cols, int ncols, const pdx_column* mask, int emit_null, pdx_mut_column* outs, void* stream) {
  PDX_TRY(check_column(mask, "pdx_filter"));
  if (mask->dtype != PDX_BOOL) return fail(PDX_INVALID, "filter mask must be boolean");
  if (!cols || !outs) return fail(PDX_INVALID, "pdx_filter: null argument");
  if (ncols >= 1 && cols[0].length != mask->length)
    return fail(PDX_INVALID, "Filter inputs must all be the same length: " + std::to_string(cols[0].length) + " vs " + std::to_string(mask->length));
  hipStream_t st = as_stream(stream);
  Scratch s;
  const int64_t n = mask->length;
  const uint8_t* mvalid = validity_or_null(mask);
  MaskPred pred{static_cast<const uint8_t*>(mask->values), mvalid, mask->offset, emit_null};
  // selected rows per tile, scanned: the output offset of every tile (and the output length)
  int64_t m = 0;
  const int64_t nblocks = ceil_div(n, kCompactTile);
  int64_t* counts = s.get<int64_t>((size_t)std::max<int64_t>(nblocks, 1));
  int64_t* total = s.get<int64_t>(1);
  PDX_SCRATCH_CHECK(s);
  if (n > 0) {
    hipLaunchKernelGGL((k_compact_count<MaskPred>), dim3((unsigned)nblocks), dim3(kCompactBlock), 0, st, n, pred, counts);
    PDX_TRY((device_exclusive_scan<int64_t, SumOp>(counts, counts, nblocks, total, s, st)));
    PDX_HIP(hipMemcpyAsync(&m, total, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    PDX_HIP(hipStreamSynchronize(st));
  }
  GatherCols g;
  PDX_TRY(fill_cols(g, cols, ncols, outs, m, n, mvalid && emit_null, "pdx_filter"));
  // Streaming form (every column read once, selected values written in order).  With nulls in play (a column bitmap, or an EMIT_NULL
  // mask with null slots) the output bitmaps are preset to ones and null rows clear their bit with a 32-bit atomic: the bitmaps
  // must start on a 4-byte boundary (fresh buffers do); otherwise the compacted-row-id gather below takes over.
  bool any_nulls = mvalid && emit_null;
  bool streaming = true;
  for (int c = 0; c < ncols; ++c) {
    any_nulls = any_nulls || g.src_valid[c];
    if (g.dst_valid[c] && (reinterpret_cast<uintptr_t>(g.dst_valid[c]) & 3)) streaming = false;
  }
  if (const char* e = getenv("PDX_FILTER_STREAM")) streaming = streaming && e[0] != '0';
  if (streaming) {
    unsigned long long* nulls = s.get<unsigned long long>(kMaxCols);
    PDX_SCRATCH_CHECK(s);
    PDX_HIP(hipMemsetAsync(nulls, 0, sizeof(unsigned long long) * kMaxCols, st));
    for (int c = 0; c < ncols; ++c)
      if (g.dst_valid[c] && m > 0) PDX_HIP(hipMemsetAsync(g.dst_valid[c], 0xFF, (size_t)((m + 7) / 8), st));
    if (m > 0) {
      if (any_nulls) {
        unsigned int* block_nulls = s.get<unsigned int>((size_t)nblocks * ncols);
        PDX_SCRATCH_CHECK(s);
        hipLaunchKernelGGL((k_filter_stream<true>), dim3((unsigned)nblocks), dim3(kCompactBlock), 0, st, g, static_cast<const uint8_t*>(mask->values), mvalid,
                           mask->offset, n, counts, emit_null, block_nulls);
        hipLaunchKernelGGL(k_sum_block_nulls, dim3(ncols), dim3(256), 0, st, block_nulls, nblocks, nulls);
      } else {
        hipLaunchKernelGGL((k_filter_stream<false>), dim3((unsigned)nblocks), dim3(kCompactBlock), 0, st, g, static_cast<const uint8_t*>(mask->values), mvalid,
                           mask->offset, n, counts, emit_null, (unsigned int*)nullptr);
      }
      PDX_LAUNCH_CHECK();
    }
    unsigned long long hn[kMaxCols] = {0};
    if (any_nulls) PDX_HIP(hipMemcpyAsync(hn, nulls, sizeof(hn), hipMemcpyDeviceToHost, st));
    PDX_HIP(hipStreamSynchronize(st));
    for (int c = 0; c < ncols; ++c) {
      outs[c].length = m;
      outs[c].null_count = g.dst_valid[c] ? (int64_t)hn[c] : 0;
    }
    return PDX_OK;
  }
  int64_t* sel = s.get<int64_t>((size_t)std::max<int64_t>(m, 1));
  PDX_SCRATCH_CHECK(s);
  MaskEmit emit{mvalid, mask->offset, sel};
  if (n > 0) {
    hipLaunchKernelGGL((k_compact_write<MaskPred, MaskEmit>), dim3((unsigned)nblocks), dim3(kCompactBlock), 0, st, n, pred, emit, counts);
    PDX_LAUNCH_CHECK();
  }
  long long bad = 0;
  return run_gather<int64_t>(g, sel, nullptr, 0, m, n, 0, outs, s, st, &bad);
}

int pdx_take(const pdx_column* cols, int ncols, const pdx_column* indices, pdx_mut_column* outs, void* stream) {
  PDX_TRY(check_column(indices, "pdx_take"));
  if (indices->dtype == PDX_BOOL) return fail(PDX_INVALID, "take indices must be integers, not boolean");
  if (indices->dtype != PDX_INT64 && indices->dtype != PDX_UINT64) return fail(PDX_NOT_IMPLEMENTED, "pdx_take: indices must be int64");
  if (!cols || !outs || ncols < 1) return fail(PDX_INVALID, "pdx_take: null argument");
  hipStream_t st = as_stream(stream);
  Scratch s;
  const int64_t m = indices->length, n = cols[0].length;
  GatherCols g;
  const uint8_t* iv = validity_or_null(indices);
  PDX_TRY(fill_cols(g, cols, ncols, outs, m, n, iv != nullptr, "pdx_take"));
  long long bad = 0;
  int rc = run_gather<int64_t>(g, static_cast<const int64_t*>(indices->values) + indices->offset, iv, indices->offset, m, n, 1, outs, s, st, &bad);
  if (rc == PDX_INDEX_ERROR) return fail(PDX_INDEX_ERROR, "Index " + std::to_string(bad) + " out of bounds");
  return rc;
}

int pdx_scatter(const pdx_column* cols, int ncols, const pdx_column* indices, pdx_mut_column* outs, void* stream) {
  PDX_TRY(check_column(indices, "pdx_scatter"));
  if (indices->dtype != PDX_INT64) return fail(PDX_INVALID, "pdx_scatter: indices must be int64");
  if (validity_or_null(indices)) return fail(PDX_INVALID, "pdx_scatter: null indices are not allowed");
  if (!cols || !outs || ncols < 1 || ncols > kMaxCols) return fail(PDX_INVALID, "pdx_scatter: between 1 and 16 columns per call");
  hipStream_t st = as_stream(stream);
  const int64_t m = indices->length;
  GatherCols g;
  g.ncols = ncols;
  int64_t n_dst = outs[0].length;
  for (int c = 0; c < ncols; ++c) {
    PDX_TRY(check_column(&cols[c], "pdx_scatter"));
    if (cols[c].length != m) return fail(PDX_INVALID, "pdx_scatter: columns and indices must have the same length");
    if (outs[c].dtype != cols[c].dtype || cols[c].dtype == PDX_BOOL) return fail(PDX_INVALID, "pdx_scatter: dtype mismatch / boolean unsupported");
    if (outs[c].length != n_dst || (n_dst && !outs[c].values)) return fail(PDX_INVALID, "pdx_scatter: bad output column");
    const uint8_t* sv = validity_or_null(&cols[c]);
    if (sv && !outs[c].validity) return fail(PDX_INVALID, "pdx_scatter: nulls possible but an output has no validity buffer");
    g.src[c] = static_cast<const uint64_t*>(cols[c].values) + cols[c].offset;
    g.src_valid[c] = sv;
    g.src_off[c] = cols[c].offset;
    g.dst[c] = static_cast<uint64_t*>(outs[c].values);
    g.dst_valid[c] = static_cast<uint8_t*>(outs[c].validity);
    outs[c].null_count = -1;
  }
  if (m == 0) return PDX_OK;
  Scratch s;
  ErrFlag* err = s.get<ErrFlag>(1);
  PDX_SCRATCH_CHECK(s);
  PDX_HIP(hipMemsetAsync(err, 0, sizeof(ErrFlag), st));
  hipLaunchKernelGGL(k_scatter, dim3(grid_for(m, 256, 4)), dim3(256), 0, st, g, static_cast<const int64_t*>(indices->values) + indices->offset, m, n_dst, err);
  PDX_LAUNCH_CHECK();
  ErrFlag h;
  PDX_HIP(hipMemcpyAsync(&h, err, sizeof(h), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
  if (h.code) return fail(PDX_INDEX_ERROR, "Index " + std::to_string(h.payload) + " out of bounds");
  return PDX_OK;
}

int pdx_concat(const pdx_column* parts, int nparts, pdx_mut_column* out, void* stream) {
  if (!parts || !out || nparts < 1 || nparts > 64) return fail(PDX_INVALID, "pdx_concat: between 1 and 64 parts per call");
  hipStream_t st = as_stream(stream);
  ConcatParts cp;
  cp.nparts = nparts;
  int64_t total = 0;
  bool any_valid = false;
  for (int q = 0; q < nparts; ++q) {
    PDX_TRY(check_column(&parts[q], "pdx_concat"));
    if (parts[q].dtype != parts[0].dtype) return fail(PDX_INVALID, "pdx_concat: parts must share one dtype (promote first)");
    if (parts[q].dtype == PDX_BOOL) return fail(PDX_NOT_IMPLEMENTED, "pdx_concat: boolean columns are not supported yet");
    cp.valid[q] = validity_or_null(&parts[q]);
    cp.off[q] = parts[q].offset;
    cp.start[q] = total;
    any_valid = any_valid || cp.valid[q];
    total += parts[q].length;
  }
  cp.start[nparts] = total;
  if (out->length < total) return fail(PDX_INVALID, "pdx_concat: output too small");
  if (out->dtype != parts[0].dtype) return fail(PDX_INVALID, "pdx_concat: output dtype must equal the parts' dtype");
  if (any_valid && !out->validity) return fail(PDX_INVALID, "pdx_concat: parts carry nulls but output has no validity buffer");
  out->length = total;
  out->null_count = 0;
  if (total == 0) return PDX_OK;
  for (int q = 0; q < nparts; ++q)
    if (parts[q].length)
      PDX_HIP(hipMemcpyAsync(static_cast<uint64_t*>(out->values) + cp.start[q], static_cast<const uint64_t*>(parts[q].values) + parts[q].offset,
                             (size_t)parts[q].length * 8, hipMemcpyDeviceToDevice, st));
  if (out->validity) {
    Scratch s;
    unsigned long long* nulls = s.get<unsigned long long>(1);
    PDX_SCRATCH_CHECK(s);
    PDX_HIP(hipMemsetAsync(nulls, 0, sizeof(*nulls), st));
    hipLaunchKernelGGL(k_concat_validity, dim3(grid_for((total + 63) >> 6, 256)), dim3(256), 0, st, cp, total, static_cast<uint8_t*>(out->validity), nulls);
    PDX_LAUNCH_CHECK();
    unsigned long long h = 0;
    PDX_HIP(hipMemcpyAsync(&h, nulls, sizeof(h), hipMemcpyDeviceToHost, st));
    PDX_HIP(hipStreamSynchronize(st));
    out->null_count = (int64_t)h;
  }
  return PDX_OK;
}

}  // extern "C"
