// temporal_round.hpp -- Arrow's floor_temporal / ceil_temporal on one int64 nanosecond timestamp, as device functions shared by
// pdx_round_temporal (temporal.hip: the rounded column) and pdx_downsample_create (groupby.hip: runs of equal rounded labels without
// materialising the column).  Semantics and their pinning: see temporal.hip.
#pragma once
#include "pdx_common.hpp"

namespace pdx {

constexpr long long kNsPerDay = 86400000000000LL;

struct RoundParams {
  long long u;       // unit in ns (fixed units), 7 days for week
  long long up;      // next larger unit in ns (calendar origin of the fixed units below day)
  long long p;       // multiple * u
  long long mult;    // multiple (months for month / quarter: multiple * 3 for quarters)
  long long week_org;  // 3 days (weeks start Monday) or 4 days (Sunday)
  int week_target;   // weekday of the anchor in the previous December: 4 = Thursday, 3 = Wednesday (0 = Sunday)
  double inv_u, inv_up, inv_p;  // reciprocals of u / up / p (0 where the divisor is unused)
};

// floor division toward -inf for b > 0 (generic form: a 64-bit hardware-less division, ~100 instructions)
__device__ __forceinline__ long long fdiv(long long a, long long b) {
  long long q = a / b;
  return (a % b < 0) ? q - 1 : q;
}
// The same for a divisor known on the host (inv = 1.0 / b): estimate the quotient in double precision, then correct it with the
// exact int64 remainder.  |estimate - a / b| <= |a / b| * 2^-51 + 1: at most one unit off for real timestamps and b >= 1 us (the
// common rules), so ONE conditional step settles it; a small divisor (rules like "7n": b < 1000 against |a| ~ 1e18) can leave the
// estimate hundreds of units off -- that case takes one exact 64-bit division of the (small) remainder instead of a loop of unit steps.
// The 64-bit division this replaces made the rounding kernels ALU bound (3.6 ms per 1e9 rows at 16 B/row).
__device__ __forceinline__ long long fdiv_c(long long a, long long b, double inv) {
  if (b == 1) return a;
  long long q = (long long)__builtin_floor((double)a * inv);
  long long r = (long long)((unsigned long long)a - (unsigned long long)q * (unsigned long long)b);
  if (r < 0) {
    --q;
    r += b;
  } else if (r >= b) {
    ++q;
    r -= b;
  }
  if (r < 0 || r >= b) {  // rare: still outside [0, b) -- floor-divide the remainder exactly
    long long adj = r / b;
    if (r % b < 0) --adj;
    q += adj;
  }
  return q;
}
// truncating division (C++ operator/) by a host-known divisor
__device__ __forceinline__ long long tdiv_c(long long a, long long b, double inv) {
  const long long q = fdiv_c(a, b, inv);
  return (a < 0 && q * b != a) ? q + 1 : q;
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
    return fdiv_c(t, q.u, q.inv_u) * q.u;
  } else if constexpr (MODE == 1) {
    return fdiv_c(t, q.p, q.inv_p) * q.p;  // floor(floor(t / u) / mult) == floor(t / (u * mult))
  } else if constexpr (MODE == 2) {
    const long long origin = fdiv_c(t, q.up, q.inv_up) * q.up;
    return fdiv_c(t - origin, q.p, q.inv_p) * q.p + origin;  // (t >= origin: truncation == floor)
  } else if constexpr (MODE == 3) {
    long long y;
    int m;
    civil_from_days(fdiv(t, kNsPerDay), &y, &m);
    const long long origin = days_from_civil(y, m, 1) * kNsPerDay;
    return fdiv_c(t - origin, q.p, q.inv_p) * q.p + origin;  // (t >= origin)
  } else if constexpr (MODE == 4) {
    return fdiv_c(t + q.week_org, q.u, q.inv_u) * q.u - q.week_org;
  } else if constexpr (MODE == 5) {
    return fdiv_c(t + q.week_org, q.p, q.inv_p) * q.p - q.week_org;
  } else if constexpr (MODE == 6) {
    const long long tt = t + q.week_org;
    long long y;
    int m;
    civil_from_days(fdiv(tt, kNsPerDay), &y, &m);
    const long long dec31 = days_from_civil(y - 1, 12, 31);
    const long long wd = ((dec31 + 4) % 7 + 7) % 7;  // 0 = Sunday (1970-01-01 was a Thursday)
    const long long last = dec31 - (((wd - q.week_target) % 7 + 7) % 7);
    const long long start = (last + 4) * kNsPerDay;  // date.h: (mon - thu) counts 4 days modulo 7
    return tdiv_c(tt - start, q.p, q.inv_p) * q.p + start;  // truncating, like the C++ it restates: tt may precede start by a few days
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

// RoundTemporalOptions -> kernel parameters and MODE; returns a PDX_* code (message through fail()).
inline int make_round_params(int64_t multiple, int unit, int week_starts_monday, int calendar_based_origin, RoundParams* out, int* mode_out,
                             const char* who) {
  if (unit < PDX_UNIT_NANOSECOND || unit > PDX_UNIT_QUARTER)
    return fail(unit == PDX_UNIT_QUARTER + 1 ? PDX_NOT_IMPLEMENTED : PDX_INVALID, std::string(who) + ": unit must be nanosecond .. quarter");
  if (multiple < 1) return fail(PDX_INVALID, std::string(who) + ": multiple must be >= 1");
  static const long long unit_ns[8] = {1LL, 1000LL, 1000000LL, 1000000000LL, 60000000000LL, 3600000000000LL, kNsPerDay, 7 * kNsPerDay};
  RoundParams q{};
  int mode;
  if (unit <= PDX_UNIT_WEEK) {
    q.u = unit_ns[unit];
    if (multiple > INT64_MAX / q.u) return fail(PDX_INVALID, std::string(who) + ": multiple x unit overflows int64 nanoseconds");
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
    q.inv_u = 1.0 / (double)q.u;
    q.inv_p = 1.0 / (double)q.p;
    q.inv_up = q.up ? 1.0 / (double)q.up : 0.0;
  } else {
    if (multiple > (1 << 24)) return fail(PDX_INVALID, std::string(who) + ": multiple too large for a calendar unit");
    q.mult = multiple * (unit == PDX_UNIT_QUARTER ? 3 : 1);
    mode = q.mult == 1 ? 7 : calendar_based_origin ? 9 : 8;
  }
  *out = q;
  *mode_out = mode;
  return PDX_OK;
}

}  // namespace pdx
