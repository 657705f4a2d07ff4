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

namespace pdx {

int launch_validity_and(const pdx_column* a, const pdx_column* b, int b_is_scalar, int64_t n, uint8_t* out, hipStream_t st);  // elementwise.hip

namespace {

constexpr long long kNsPerDay = 86400000000000LL;

struct RoundParams {
  long long u;       // unit in ns (fixed units), 7 days for week
  long long up;      // next larger unit in ns (calendar origin of the fixed units below day)
  long long p;       // multiple * u
  long long mult;    // multiple (months for month / quarter: multiple * 3 for quarters)
  long long week_org;  // 3 days (weeks start Monday) or 4 days (Sunday)
  int week_target;   // weekday of the anchor in the previous December: 4 = Thursday, 3 = Wednesday (0 = Sunday)
};

// floor division / remainder toward -inf for b > 0
__device__ __forceinline__ long long fdiv(long long a, long long b) {
  long long q = a / b;
  return (a % b < 0) ? q - 1 : q;
}

// days since 1970-01-01 <-> proleptic Gregorian civil date (H. Hinnant's algorithms, the ones Arrow's vendored date.h uses)
__device__ __forceinline__ void civil_from_days(long long z, long long* y, int* m) {
  z += 719468;
  const long long era = (z >= 0 ? z : z - 146096) / 146097;
  const long long doe = z - era * 146097;
  const long long yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  const long long doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  const long long mp = (5 * doy + 2) / 153;
  *m = (int)(mp < 10 ? mp + 3 : mp - 9);
  *y = yoe + era * 400 + (*m <= 2);
}
__device__ __forceinline__ long long days_from_civil(long long y, int m, int d) {
  y -= m <= 2;
  const long long era = (y >= 0 ? y : y - 399) / 400;
  const long long yoe = y - era * 400;
  const long long doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
  const long long doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  return era * 146097 + doe - 719468;
}

// MODE: 0 = fixed unit, multiple == 1      1 = fixed unit, multiples since the epoch      2 = fixed unit below day, calendar origin
//       3 = day, calendar origin (1st of the month)    4 = week, multiple == 1    5 = week, multiples since the epoch
//       6 = week, calendar origin          7 = month / quarter, multiple == 1 month       8 = months since 1970-01
//       9 = months since January of the year
template <int MODE>
__device__ __forceinline__ long long floor_one(long long t, const RoundParams& q) {
  if constexpr (MODE == 0) {
    return fdiv(t, q.u) * q.u;
  } else if constexpr (MODE == 1) {
    return fdiv(t, q.p) * q.p;  // floor(floor(t / u) / mult) == floor(t / (u * mult))
  } else if constexpr (MODE == 2) {
    const long long origin = fdiv(t, q.up) * q.up;
    return (t - origin) / q.p * q.p + origin;
  } else if constexpr (MODE == 3) {
    long long y;
    int m;
    civil_from_days(fdiv(t, kNsPerDay), &y, &m);
    const long long origin = days_from_civil(y, m, 1) * kNsPerDay;
    return (t - origin) / q.p * q.p + origin;
  } else if constexpr (MODE == 4) {
    return fdiv(t + q.week_org, q.u) * q.u - q.week_org;
  } else if constexpr (MODE == 5) {
    return fdiv(t + q.week_org, q.p) * q.p - q.week_org;
  } else if constexpr (MODE == 6) {
    const long long tt = t + q.week_org;
    long long y;
    int m;
    civil_from_days(fdiv(tt, kNsPerDay), &y, &m);
    const long long dec31 = days_from_civil(y - 1, 12, 31);
    const long long wd = ((dec31 + 4) % 7 + 7) % 7;  // 0 = Sunday (1970-01-01 was a Thursday)
    const long long last = dec31 - (((wd - q.week_target) % 7 + 7) % 7);
    const long long start = (last + 4) * kNsPerDay;  // date.h: (mon - thu) counts 4 days modulo 7
    return (tt - start) / q.p * q.p + start;         // truncating, like the C++ it restates: tt may precede start by a few days
  } else {
    long long y;
    int m;
    civil_from_days(fdiv(t, kNsPerDay), &y, &m);
    if constexpr (MODE == 7) return days_from_civil(y, m, 1) * kNsPerDay;
    if constexpr (MODE == 9) return days_from_civil(y, 1 + (int)((m - 1) / q.mult * q.mult), 1) * kNsPerDay;
    const long long tm = fdiv((y - 1970) * 12 + m - 1, q.mult) * q.mult;
    const long long yy = fdiv(tm, 12);
    return days_from_civil(1970 + yy, (int)(tm - yy * 12) + 1, 1) * kNsPerDay;
  }
}

template <int MODE, bool CEIL>
__device__ __forceinline__ long long round_one(long long t, const RoundParams& q) {
  const long long f = floor_one<MODE>(t, q);
  if constexpr (!CEIL) return f;
  if constexpr (MODE <= 6) return f >= t ? f : f + q.p;
  long long y;
  int m;
  civil_from_days(fdiv(f, kNsPerDay), &y, &m);
  const long long tm = y * 12 + m - 1 + q.mult;
  const long long yy = fdiv(tm, 12);
  return days_from_civil(yy, (int)(tm - yy * 12) + 1, 1) * kNsPerDay;
}

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
  if (unit < PDX_UNIT_NANOSECOND || unit > PDX_UNIT_QUARTER)
    return fail(unit == PDX_UNIT_QUARTER + 1 ? PDX_NOT_IMPLEMENTED : PDX_INVALID, "pdx_round_temporal: unit must be nanosecond .. quarter");
  if (multiple < 1) return fail(PDX_INVALID, "pdx_round_temporal: multiple must be >= 1");
  if (!out || out->length < ts->length || out->dtype != PDX_TIMESTAMP_NS)
    return fail(PDX_INVALID, "pdx_round_temporal: output must be PDX_TIMESTAMP_NS of the input length");
  const bool has_nulls = validity_or_null(ts) != nullptr;
  if (has_nulls && !out->validity) return fail(PDX_INVALID, "pdx_round_temporal: input carries nulls but output has no validity buffer");
  static const long long unit_ns[8] = {1LL, 1000LL, 1000000LL, 1000000000LL, 60000000000LL, 3600000000000LL, kNsPerDay, 7 * kNsPerDay};
  RoundParams q{};
  int mode;
  if (unit <= PDX_UNIT_WEEK) {
    q.u = unit_ns[unit];
    if (multiple > INT64_MAX / q.u) return fail(PDX_INVALID, "pdx_round_temporal: multiple x unit overflows int64 nanoseconds");
    q.p = multiple * q.u;
    q.mult = multiple;
    if (unit <= PDX_UNIT_DAY) {
      q.up = unit < PDX_UNIT_DAY ? unit_ns[unit + 1] : 0;
      mode = multiple == 1 ? 0 : !calendar_based_origin ? 1 : unit == PDX_UNIT_DAY ? 3 : 2;
    } else {
      q.week_org = (week_starts_monday ? 3 : 4) * kNsPerDay;
      q.week_target = week_starts_monday ? 4 : 3;
      mode = multiple == 1 ? 4 : !calendar_based_origin ? 5 : 6;
    }
  } else {
    if (multiple > (1 << 24)) return fail(PDX_INVALID, "pdx_round_temporal: multiple too large for a calendar unit");
    q.mult = multiple * (unit == PDX_UNIT_QUARTER ? 3 : 1);
    mode = q.mult == 1 ? 7 : calendar_based_origin ? 9 : 8;
  }
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
