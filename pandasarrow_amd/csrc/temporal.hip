// temporal.hip -- floor_temporal / ceil_temporal on timestamp[ns] columns for gfx950.
//
// Replaces arrow::compute::FloorTemporal / CeilTemporal(m_index, RoundTemporalOptions(multiple, unit, week_starts_monday,
// ceil_is_strictly_greater = false, calendar_based_origin)) in DataFrame::downsample (reference src/dataframe.cpp:1265-1290):
// the binned index that the Resampler's GroupBy is then keyed on.  One HBM-bound stream: 8 B read + 8 B write per row
// (16 B/row), grid-stride over coalesced 8-byte lanes, 4 rows in flight per thread; all the calendar work is integer
// arithmetic in registers.  Semantics restate Arrow C++ 25.0.0 (pinned by the vectors in tests/golden/arrow_golden_r2.npz): floors go toward -inf, a calendar origin is the floor to the next larger unit (day: the
// 1st of the month; week: the Monday/Sunday after the last Thursday/Wednesday of the previous December), month / quarter are
// counted in calendar months, ceil = floor when floor >= t, else floor + multiple x unit (month / quarter: always floor +
// multiple -- Arrow ignores ceil_is_strictly_greater there).
#include "pdx_common.hpp"
#include "temporal_round.hpp"

namespace pdx {

int launch_validity_and(const pdx_column* a, const pdx_column* b, int b_is_scalar, int64_t n, uint8_t* out, hipStream_t st);  // elementwise.hip

namespace {

template <int MODE, bool CEIL>
__global__ void __launch_bounds__(256) k_round_temporal(const long long* __restrict__ ts, long long* __restrict__ out, int64_t n, RoundParams q) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    long long t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = ts[i + k * stride];
#pragma unroll
    for (int k = 0; k < 4; ++k) out[i + k * stride] = round_one<MODE, CEIL>(t[k], q);
  }
  for (; i < n; i += stride) out[i] = round_one<MODE, CEIL>(ts[i], q);
}

template <int MODE>
void launch_mode(bool ceil_mode, const long long* ts, long long* out, int64_t n, const RoundParams& q, hipStream_t st) {
  dim3 grid(grid_for(n, 256, 4)), block(256);
  if (ceil_mode) hipLaunchKernelGGL((k_round_temporal<MODE, true>), grid, block, 0, st, ts, out, n, q);
  else hipLaunchKernelGGL((k_round_temporal<MODE, false>), grid, block, 0, st, ts, out, n, q);
}

}  // namespace
}  // namespace pdx

using namespace pdx;

extern "C" {

int pdx_round_temporal(int ceil_mode, const pdx_column* ts, int64_t multiple, int unit, int week_starts_monday, int calendar_based_origin,
                       pdx_mut_column* out, void* stream) {
  PDX_TRY(check_column(ts, "pdx_round_temporal"));
  if (ts->dtype != PDX_TIMESTAMP_NS) return fail(PDX_INVALID, "pdx_round_temporal: input must be PDX_TIMESTAMP_NS");
  RoundParams q{};
  int mode = 0;
  PDX_TRY(make_round_params(multiple, unit, week_starts_monday, calendar_based_origin, &q, &mode, "pdx_round_temporal"));
  if (!out || out->length < ts->length || out->dtype != PDX_TIMESTAMP_NS)
    return fail(PDX_INVALID, "pdx_round_temporal: output must be PDX_TIMESTAMP_NS of the input length");
  const bool has_nulls = validity_or_null(ts) != nullptr;
  if (has_nulls && !out->validity) return fail(PDX_INVALID, "pdx_round_temporal: input carries nulls but output has no validity buffer");
  hipStream_t st = as_stream(stream);
  const int64_t n = ts->length;
  out->length = n;
  out->null_count = has_nulls ? -1 : 0;
  if (n == 0) return PDX_OK;
  if (!out->values) return fail(PDX_INVALID, "pdx_round_temporal: null output buffer");
  const long long* in = static_cast<const long long*>(ts->values) + ts->offset;
  long long* o = static_cast<long long*>(out->values);
  {
    PDX_PROFILE("round_temporal", st);
    switch (mode) {
      case 0: launch_mode<0>(ceil_mode, in, o, n, q, st); break;
      case 1: launch_mode<1>(ceil_mode, in, o, n, q, st); break;
      case 2: launch_mode<2>(ceil_mode, in, o, n, q, st); break;
      case 3: launch_mode<3>(ceil_mode, in, o, n, q, st); break;
      case 4: launch_mode<4>(ceil_mode, in, o, n, q, st); break;
      case 5: launch_mode<5>(ceil_mode, in, o, n, q, st); break;
      case 6: launch_mode<6>(ceil_mode, in, o, n, q, st); break;
      case 7: launch_mode<7>(ceil_mode, in, o, n, q, st); break;
      case 8: launch_mode<8>(ceil_mode, in, o, n, q, st); break;
      default: launch_mode<9>(ceil_mode, in, o, n, q, st); break;
    }
    PDX_LAUNCH_CHECK();
  }
  if (out->validity) PDX_TRY(launch_validity_and(ts, nullptr, 0, n, static_cast<uint8_t*>(out->validity), st));
  return PDX_OK;
}

}  // extern "C"
