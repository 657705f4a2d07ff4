/*
 * pdx_oracle.c -- CPU ORACLE (test infrastructure only; see pdx_oracle.h).
 *
 * Plain-C restatement of the Arrow C++ 25.0.0 behaviour that EPOCHDevs/PandasArrow
 * forwards to on its vectorized operator path.  Every function cites the reference
 * call site it follows (file:line relative to /root/reference).  No code is copied
 * from the reference or from Arrow: algorithms are restated from their observable
 * behaviour and pinned by tests/golden/.
 */
#include "pdx_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline int bit_get(const uint8_t* bits, int64_t i) { return (bits[i >> 3] >> (i & 7)) & 1; }
static inline void bit_set_to(uint8_t* bits, int64_t i, int v) {
  if (v) bits[i >> 3] |= (uint8_t)(1u << (i & 7));
  else bits[i >> 3] &= (uint8_t)~(1u << (i & 7));
}
static inline int is_valid(const uint8_t* valid, int64_t off, int64_t i) { return valid ? bit_get(valid, off + i) : 1; }

/* ------------------------------------------------------------------ synthetic inputs */
uint64_t orc_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
/* key[i] = mix(i ^ 0x5EED0001) % num_keys   */
void orc_synth_keys(int64_t start, int64_t n, int64_t num_keys, int64_t* out) {
  for (int64_t k = 0; k < n; ++k) {
    uint64_t i = (uint64_t)(start + k);
    out[k] = (int64_t)(orc_splitmix64(i ^ 0x5EED0001ull) % (uint64_t)num_keys);
  }
}
/* val[i] = (mix(i + 0x5EED0002 + seed_off) >> 11) * 2^-53 in [0,1) */
void orc_synth_vals(int64_t start, int64_t n, uint64_t seed_off, double* out) {
  for (int64_t k = 0; k < n; ++k) {
    uint64_t i = (uint64_t)(start + k);
    out[k] = (double)(orc_splitmix64(i + 0x5EED0002ull + seed_off) >> 11) * 0x1.0p-53;
  }
}
void orc_synth_ts(int64_t start, int64_t n, int64_t t0_ns, int64_t step_ns, int64_t* out) {
  for (int64_t k = 0; k < n; ++k) out[k] = t0_ns + (start + k) * step_ns;
}

/* ------------------------------------------------------------------ pairwise fp64 sum
 * Reference call sites: NDFrame::sum (src/ndframe.cpp:220 via macro 26-31) and every
 * per-group CallFunction("sum"/"mean") in src/pd_core_macros.h:31,66,103,132.
 * Arrow's floating-point scalar `sum`: sequential 16-value leaves per run of valid
 * values, binary-counter merge of leaf sums, final low->high fold. */
typedef struct {
  double sum[64];
  uint64_t mask;
  int root;
} pw_state;

/* NaN results, bit for bit as Arrow C++ 25 on x86-64 produces them (measured: tests/test_oracle_golden_r4.py, oracle/gen_golden_nanbits.py):
 * an SSE add returns its FIRST NaN operand quieted (the second if only that one is NaN) and the negative default NaN for inf + -inf.
 * Inside a 16-value leaf the accumulator -- the EARLIER rows -- is the first operand; in every merge of the tree (counter pushes, the
 * final fold) the compiled code adds the LATER operand first.  Spelled out here so that the restatement does not depend on which
 * operand order THIS compiler happens to pick for `a += b`. */
static double pw_nan_of(double first, double second) {
  uint64_t b;
  if (first != first) memcpy(&b, &first, 8), b |= 0x0008000000000000ull;
  else if (second != second) memcpy(&b, &second, 8), b |= 0x0008000000000000ull;
  else b = 0xFFF8000000000000ull;
  double r;
  memcpy(&r, &b, 8);
  return r;
}
static double pw_leaf_add(double earlier, double later) {
  const double r = earlier + later;
  return r == r ? r : pw_nan_of(earlier, later);
}
static double pw_merge(double earlier, double later) {
  const double r = earlier + later;
  return r == r ? r : pw_nan_of(later, earlier);
}
static void pw_init(pw_state* s) {
  for (int i = 0; i < 64; ++i) s->sum[i] = 0.0;
  s->mask = 0;
  s->root = 0;
}
static void pw_reduce(pw_state* s, double block) {
  int cur = 0;
  uint64_t m = 1;
  s->sum[0] = pw_merge(s->sum[0], block);
  s->mask ^= m;
  while ((s->mask & m) == 0) {
    block = s->sum[cur];
    s->sum[cur] = 0.0;
    ++cur;
    m <<= 1;
    s->sum[cur] = pw_merge(s->sum[cur], block);
    s->mask ^= m;
  }
  if (cur > s->root) s->root = cur;
}
static double pw_finish(pw_state* s) {
  for (int i = 1; i <= s->root; ++i) s->sum[i] = pw_merge(s->sum[i], s->sum[i - 1]);
  return s->sum[s->root];
}
/* feed one run of consecutive valid values */
static void pw_run(pw_state* s, const double* v, int64_t len) {
  int64_t blocks = len / 16, rem = len % 16;
  for (int64_t b = 0; b < blocks; ++b) {
    double acc = 0.0;
    for (int j = 0; j < 16; ++j) acc = pw_leaf_add(acc, v[j]);
    pw_reduce(s, acc);
    v += 16;
  }
  if (rem > 0) {
    double acc = 0.0;
    for (int64_t j = 0; j < rem; ++j) acc = pw_leaf_add(acc, v[j]);
    pw_reduce(s, acc);
  }
}

int orc_sum_f64(const double* v, const uint8_t* valid, int64_t off, int64_t n, double* out, int64_t* count) {
  pw_state s;
  pw_init(&s);
  int64_t cnt = 0;
  if (!valid) {
    cnt = n;
    if (n > 0) pw_run(&s, v + off, n);
  } else {
    int64_t i = 0;
    while (i < n) {
      while (i < n && !bit_get(valid, off + i)) ++i;
      int64_t st = i;
      while (i < n && bit_get(valid, off + i)) ++i;
      if (i > st) {
        pw_run(&s, v + off + st, i - st);
        cnt += i - st;
      }
    }
  }
  *count = cnt;
  *out = cnt ? pw_finish(&s) : 0.0; /* min_count=1: count==0 -> null (caller checks count) */
  return ORC_OK;
}

/* integer sum wraps (two's complement); int64 -> int64 */
int orc_sum_i64(const int64_t* v, const uint8_t* valid, int64_t off, int64_t n, int64_t* out, int64_t* count) {
  uint64_t acc = 0;
  int64_t cnt = 0;
  for (int64_t i = 0; i < n; ++i)
    if (is_valid(valid, off, i)) {
      acc += (uint64_t)v[off + i];
      ++cnt;
    }
  *out = (int64_t)acc;
  *count = cnt;
  return ORC_OK;
}
/* mean = pairwise sum / count (NDFrame::mean src/ndframe.cpp:162) */
int orc_mean_f64(const double* v, const uint8_t* valid, int64_t off, int64_t n, double* out, int64_t* count) {
  double s;
  orc_sum_f64(v, valid, off, n, &s, count);
  *out = *count ? s / (double)*count : 0.0;
  return ORC_OK;
}
/* integer mean (Arrow 25.0.0): values are converted to double one by one and summed with the SAME pairwise
 * algorithm as fp64 (no int64 wrap-around), then divided by the valid count.  Pinned by tests/golden (agg_i64_*). */
int orc_mean_i64(const int64_t* v, const uint8_t* valid, int64_t off, int64_t n, double* out, int64_t* count) {
  double* d = (double*)malloc((size_t)(n ? n : 1) * sizeof(double));
  for (int64_t i = 0; i < n; ++i) d[i] = (double)v[off + i];
  /* validity bitmap is addressed with `off`, the converted values start at 0: shift by re-walking runs */
  pw_state s;
  pw_init(&s);
  int64_t cnt = 0, i = 0;
  while (i < n) {
    while (i < n && !is_valid(valid, off, i)) ++i;
    int64_t st = i;
    while (i < n && is_valid(valid, off, i)) ++i;
    if (i > st) {
      pw_run(&s, d + st, i - st);
      cnt += i - st;
    }
  }
  free(d);
  *count = cnt;
  *out = cnt ? pw_finish(&s) / (double)cnt : 0.0;
  return ORC_OK;
}
/* variance / stddev with default VarianceOptions (ddof = 0, skip_nulls, min_count = 0), reached per group through
 * GROUPBY_NUMERIC_AGG(variance|stddev) src/dataframe.cpp:1516-1520 with null options.  Arrow 25.0.0 (VarStdImpl, two passes):
 * mean = pairwise sum / count, m2 = pairwise sum of (x - mean)^2 over the same valid runs, var = m2 / (count - ddof);
 * int64 values are converted to double first.  Pinned against pyarrow (tests/golden: *_variance, *_stddev). */
int orc_var_f64(const double* v, const uint8_t* valid, int64_t off, int64_t n, int want_std, double* out, int64_t* count) {
  double s = 0.0;
  orc_sum_f64(v, valid, off, n, &s, count);
  if (*count == 0) {
    *out = 0.0;
    return ORC_OK;
  }
  const double mean = s / (double)*count;
  double* d = (double*)malloc((size_t)(n + off + 1) * sizeof(double));
  for (int64_t i = 0; i < n; ++i) {
    double x = v[off + i] - mean;
    d[off + i] = x * x;
  }
  double m2 = 0.0;
  int64_t c2 = 0;
  orc_sum_f64(d, valid, off, n, &m2, &c2);
  free(d);
  double var = m2 / (double)*count;
  *out = want_std ? sqrt(var) : var;
  return ORC_OK;
}
/* product (ScalarAggregateOptions default: skip_nulls, min_count = 1): one multiply per valid value in row order;
 * int64 wraps (GROUPBY_AGG(product) src/dataframe.cpp:1536) */
int orc_product_f64(const double* v, const uint8_t* valid, int64_t off, int64_t n, double* out, int64_t* count) {
  double p = 1.0;
  int64_t c = 0;
  for (int64_t i = 0; i < n; ++i)
    if (is_valid(valid, off, i)) {
      p = p * v[off + i];
      ++c;
    }
  *out = p;
  *count = c;
  return ORC_OK;
}
int orc_product_i64(const int64_t* v, const uint8_t* valid, int64_t off, int64_t n, int64_t* out, int64_t* count) {
  uint64_t p = 1;
  int64_t c = 0;
  for (int64_t i = 0; i < n; ++i)
    if (is_valid(valid, off, i)) {
      p = p * (uint64_t)v[off + i];
      ++c;
    }
  *out = (int64_t)p;
  *count = c;
  return ORC_OK;
}
/* min/max: nulls skipped; NaN skipped unless every valid value is NaN; first value wins ties
 * (NDFrame::min/max src/ndframe.cpp:163-166; MinMax in src/resample.cpp:223) */
int orc_minmax_f64(const double* v, const uint8_t* valid, int64_t off, int64_t n, double* mn, double* mx, int64_t* count) {
  int64_t cnt = 0;
  int have = 0;
  double lo = 0, hi = 0;
  /* Ties between values that compare equal but differ in bits (0.0 / -0.0), pinned against Arrow 25.0.0: min keeps the FIRST;
   * max keeps the FIRST when the array has no nulls and the LAST when it has at least one (Arrow's null-aware loop is a different
   * instantiation of the same fmax step, compiled with the operands the other way round). */
  int has_nulls = 0;
  for (int64_t i = 0; i < n && !has_nulls; ++i) has_nulls = !is_valid(valid, off, i);
  for (int64_t i = 0; i < n; ++i) {
    if (!is_valid(valid, off, i)) continue;
    ++cnt;
    double x = v[off + i];
    if (x != x) continue;
    if (!have) {
      lo = hi = x;
      have = 1;
    } else {
      if (x < lo) lo = x;
      if (x > hi || (has_nulls && x == hi)) hi = x;
    }
  }
  *count = cnt;
  if (cnt && !have) lo = hi = NAN;
  *mn = lo;
  *mx = hi;
  return ORC_OK;
}
int orc_minmax_i64(const int64_t* v, const uint8_t* valid, int64_t off, int64_t n, int64_t* mn, int64_t* mx, int64_t* count) {
  int64_t cnt = 0, lo = 0, hi = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (!is_valid(valid, off, i)) continue;
    int64_t x = v[off + i];
    if (!cnt) lo = hi = x;
    else {
      if (x < lo) lo = x;
      if (x > hi) hi = x;
    }
    ++cnt;
  }
  *count = cnt;
  *mn = lo;
  *mx = hi;
  return ORC_OK;
}
int64_t orc_count(const uint8_t* valid, int64_t off, int64_t n) {
  if (!valid) return n;
  int64_t c = 0;
  for (int64_t i = 0; i < n; ++i) c += bit_get(valid, off + i);
  return c;
}

/* ------------------------------------------------------------------ element-wise */
void orc_validity_and(const uint8_t* va, int64_t aoff, const uint8_t* vb, int64_t boff, int64_t n, uint8_t* out_valid) {
  if (!out_valid) return;
  memset(out_valid, 0, (size_t)((n + 7) / 8));
  for (int64_t i = 0; i < n; ++i) bit_set_to(out_valid, i, is_valid(va, aoff, i) && is_valid(vb, boff, i));
}
static void valid_and_scalar(const uint8_t* va, int64_t aoff, const uint8_t* vb, int64_t boff, int b_is_scalar, int64_t n,
                             uint8_t* out_valid) {
  if (!out_valid) return;
  memset(out_valid, 0, (size_t)((n + 7) / 8));
  /* b_is_scalar: 0 = arrays, 1 = b is ONE value (Series op Scalar), 2 = a is ONE value (Scalar op Series, src/scalar.cpp:24-36) */
  for (int64_t i = 0; i < n; ++i)
    bit_set_to(out_valid, i, (b_is_scalar == 2 ? is_valid(va, aoff, 0) : is_valid(va, aoff, i)) &&
                                 (b_is_scalar == 1 ? is_valid(vb, boff, 0) : is_valid(vb, boff, i)));
}

/* Series::operator+,-,*,/ (src/series.cpp:19-33,229-235): "add/subtract/multiply/divide", null if either side null. */
int orc_binary_f64(int op, const double* a, const uint8_t* va, int64_t aoff, const double* b, const uint8_t* vb, int64_t boff,
                   int b_is_scalar, int64_t n, double* out, uint8_t* out_valid) {
  for (int64_t i = 0; i < n; ++i) {
    double x = b_is_scalar == 2 ? a[aoff] : a[aoff + i], y = b_is_scalar == 1 ? b[boff] : b[boff + i];
    double r;
    switch (op) {
      case ORC_ADD: r = x + y; break;
      case ORC_SUB: r = x - y; break;
      case ORC_MUL: r = x * y; break;
      case ORC_DIV: r = x / y; break; /* IEEE: 1/0=inf, 0/0=NaN */
      default: return ORC_INVALID;
    }
    out[i] = r;
  }
  valid_and_scalar(va, aoff, vb, boff, b_is_scalar, n, out_valid);
  return ORC_OK;
}
/* unchecked integer arithmetic: wraps; divide truncates toward zero, INT64_MIN/-1 -> 0,
 * divisor 0 at a VALID slot -> ArrowInvalid "divide by zero" for the whole call */
int orc_binary_i64(int op, const int64_t* a, const uint8_t* va, int64_t aoff, const int64_t* b, const uint8_t* vb, int64_t boff,
                   int b_is_scalar, int64_t n, int64_t* out, uint8_t* out_valid) {
  for (int64_t i = 0; i < n; ++i) {
    int64_t x = b_is_scalar == 2 ? a[aoff] : a[aoff + i], y = b_is_scalar == 1 ? b[boff] : b[boff + i];
    int ok = (b_is_scalar == 2 ? is_valid(va, aoff, 0) : is_valid(va, aoff, i)) && (b_is_scalar == 1 ? is_valid(vb, boff, 0) : is_valid(vb, boff, i));
    int64_t r;
    switch (op) {
      case ORC_ADD: r = (int64_t)((uint64_t)x + (uint64_t)y); break;
      case ORC_SUB: r = (int64_t)((uint64_t)x - (uint64_t)y); break;
      case ORC_MUL: r = (int64_t)((uint64_t)x * (uint64_t)y); break;
      case ORC_DIV:
        if (!ok) { r = 0; break; }
        if (y == 0) return ORC_INVALID;
        if (x == INT64_MIN && y == -1) r = 0;
        else r = x / y;
        break;
      /* bit_wise_or / and / xor, shift_left / shift_right (src/series.cpp:237-245): Arrow's unchecked shifts return the left
       * operand when the amount is negative or >= std::numeric_limits<int64_t>::digits (63); shift_right is arithmetic */
      case 4: r = (int64_t)((uint64_t)x | (uint64_t)y); break;
      case 5: r = (int64_t)((uint64_t)x & (uint64_t)y); break;
      case 6: r = (int64_t)((uint64_t)x ^ (uint64_t)y); break;
      case 7: r = (y < 0 || y >= 63) ? x : (int64_t)((uint64_t)x << y); break;
      case 8: r = (y < 0 || y >= 63) ? x : (x >> y); break;
      default: return ORC_INVALID;
    }
    out[i] = r;
  }
  valid_and_scalar(va, aoff, vb, boff, b_is_scalar, n, out_valid);
  return ORC_OK;
}

#define CMP_BODY(T)                                                                          \
  memset(out_bits, 0, (size_t)((n + 7) / 8));                                                \
  for (int64_t i = 0; i < n; ++i) {                                                          \
    T x = b_is_scalar == 2 ? a[aoff] : a[aoff + i], y = b_is_scalar == 1 ? b[boff] : b[boff + i]; \
    int r;                                                                                   \
    switch (op) {                                                                            \
      case ORC_EQ: r = x == y; break;                                                        \
      case ORC_NE: r = x != y; break;                                                        \
      case ORC_LT: r = x < y; break;                                                         \
      case ORC_LE: r = x <= y; break;                                                        \
      case ORC_GT: r = x > y; break;                                                         \
      case ORC_GE: r = x >= y; break;                                                        \
      default: return ORC_INVALID;                                                           \
    }                                                                                        \
    bit_set_to(out_bits, i, r);                                                              \
  }                                                                                          \
  valid_and_scalar(va, aoff, vb, boff, b_is_scalar, n, out_valid);                           \
  return ORC_OK;

/* Series::operator{>,>=,<,<=,==,!=} (src/series.cpp:247-257): bit-packed LSB-first; NaN false except != */
int orc_compare_f64(int op, const double* a, const uint8_t* va, int64_t aoff, const double* b, const uint8_t* vb, int64_t boff,
                    int b_is_scalar, int64_t n, uint8_t* out_bits, uint8_t* out_valid) {
  CMP_BODY(double)
}
int orc_compare_i64(int op, const int64_t* a, const uint8_t* va, int64_t aoff, const int64_t* b, const uint8_t* vb, int64_t boff,
                    int b_is_scalar, int64_t n, uint8_t* out_bits, uint8_t* out_valid) {
  CMP_BODY(int64_t)
}
/* "and"/"or" are the non-Kleene kernels (src/series.cpp:259-260): null if either side null */
int orc_logical(int op, const uint8_t* a, const uint8_t* va, int64_t aoff, const uint8_t* b, const uint8_t* vb, int64_t boff,
                int64_t n, uint8_t* out_bits, uint8_t* out_valid) {
  memset(out_bits, 0, (size_t)((n + 7) / 8));
  for (int64_t i = 0; i < n; ++i) {
    int x = bit_get(a, aoff + i), y = bit_get(b, boff + i);
    bit_set_to(out_bits, i, op == ORC_AND ? (x & y) : (x | y));
  }
  orc_validity_and(va, aoff, vb, boff, n, out_valid);
  return ORC_OK;
}
void orc_invert(const uint8_t* a, int64_t aoff, int64_t n, uint8_t* out_bits) {
  memset(out_bits, 0, (size_t)((n + 7) / 8));
  for (int64_t i = 0; i < n; ++i) bit_set_to(out_bits, i, !bit_get(a, aoff + i));
}

/* ------------------------------------------------------------------ filter / take */
/* DataFrame::where (src/dataframe.cpp:461-475): "filter" with EMIT_NULL; Series::where index uses DROP (src/series.cpp:130-144) */
int64_t orc_filter_count(const uint8_t* mask, const uint8_t* mask_valid, int64_t moff, int64_t n, int emit_null) {
  int64_t c = 0;
  for (int64_t i = 0; i < n; ++i) {
    int mv = is_valid(mask_valid, moff, i);
    if (mv ? bit_get(mask, moff + i) : emit_null) ++c;
  }
  return c;
}
int orc_filter_64(const uint64_t* v, const uint8_t* valid, int64_t off, const uint8_t* mask, const uint8_t* mask_valid,
                  int64_t moff, int64_t n, int emit_null, uint64_t* out, uint8_t* out_valid, int64_t* out_nulls) {
  int64_t o = 0, nulls = 0;
  for (int64_t i = 0; i < n; ++i) {
    int mv = is_valid(mask_valid, moff, i);
    if (mv) {
      if (!bit_get(mask, moff + i)) continue;
      int ok = is_valid(valid, off, i);
      out[o] = ok ? v[off + i] : 0;
      if (out_valid) bit_set_to(out_valid, o, ok);
      nulls += !ok;
      ++o;
    } else if (emit_null) {
      out[o] = 0;
      if (out_valid) bit_set_to(out_valid, o, 0);
      ++nulls;
      ++o;
    }
  }
  if (out_nulls) *out_nulls = nulls;
  return ORC_OK;
}
/* DataFrame::take (src/dataframe.cpp:477-492): bounds-checked gather; null index -> null row */
int orc_take_64(const uint64_t* v, const uint8_t* valid, int64_t off, int64_t n, const int64_t* idx, const uint8_t* idx_valid,
                int64_t ioff, int64_t m, uint64_t* out, uint8_t* out_valid, int64_t* out_nulls, int64_t* bad_index) {
  int64_t nulls = 0;
  for (int64_t j = 0; j < m; ++j) {
    if (!is_valid(idx_valid, ioff, j)) continue;
    int64_t k = idx[ioff + j];
    if (k < 0 || k >= n) {
      if (bad_index) *bad_index = k;
      return ORC_INDEX_ERROR;
    }
  }
  for (int64_t j = 0; j < m; ++j) {
    if (!is_valid(idx_valid, ioff, j)) {
      out[j] = 0;
      if (out_valid) bit_set_to(out_valid, j, 0);
      ++nulls;
      continue;
    }
    int64_t k = idx[ioff + j];
    int ok = is_valid(valid, off, k);
    out[j] = ok ? v[off + k] : 0;
    if (out_valid) bit_set_to(out_valid, j, ok);
    nulls += !ok;
  }
  if (out_nulls) *out_nulls = nulls;
  return ORC_OK;
}

/* ------------------------------------------------------------------ group-by */
/* GroupBy::makeGroups -> Grouper::Consume (src/dataframe.cpp:1580-1584): ids dense, first-occurrence order. */
int64_t orc_group_ids_i64(const int64_t* keys, const uint8_t* valid, int64_t off, int64_t n, uint32_t* ids, int64_t* uniques,
                          uint8_t* unique_is_null, int64_t* first_row) {
  uint64_t cap = 16;
  while (cap < (uint64_t)n * 2 + 2) cap <<= 1;
  int64_t* slot_gid = (int64_t*)malloc(cap * sizeof(int64_t));
  for (uint64_t i = 0; i < cap; ++i) slot_gid[i] = -1;
  int64_t G = 0, null_gid = -1;
  for (int64_t i = 0; i < n; ++i) {
    if (!is_valid(valid, off, i)) {
      if (null_gid < 0) {
        null_gid = G;
        uniques[G] = 0;
        if (unique_is_null) unique_is_null[G] = 1;
        if (first_row) first_row[G] = i;
        ++G;
      }
      ids[i] = (uint32_t)null_gid;
      continue;
    }
    int64_t k = keys[off + i];
    uint64_t h = orc_splitmix64((uint64_t)k) & (cap - 1);
    for (;;) {
      int64_t g = slot_gid[h];
      if (g < 0) {
        slot_gid[h] = G;
        uniques[G] = k;
        if (unique_is_null) unique_is_null[G] = 0;
        if (first_row) first_row[G] = i;
        ids[i] = (uint32_t)G;
        ++G;
        break;
      }
      if (uniques[g] == k && !(unique_is_null && unique_is_null[g])) {
        ids[i] = (uint32_t)g;
        break;
      }
      h = (h + 1) & (cap - 1);
    }
  }
  free(slot_gid);
  return G;
}
/* Grouper::MakeGroupings (src/dataframe.cpp:1586-1588): row ids ascending within each group */
void orc_make_groupings(const uint32_t* ids, int64_t n, int64_t G, int64_t* offsets, int64_t* rows) {
  for (int64_t g = 0; g <= G; ++g) offsets[g] = 0;
  for (int64_t i = 0; i < n; ++i) offsets[ids[i] + 1]++;
  for (int64_t g = 0; g < G; ++g) offsets[g + 1] += offsets[g];
  int64_t* cur = (int64_t*)malloc((size_t)(G + 1) * sizeof(int64_t));
  memcpy(cur, offsets, (size_t)(G + 1) * sizeof(int64_t));
  for (int64_t i = 0; i < n; ++i) rows[cur[ids[i]]++] = i;
  free(cur);
}

/* GROUPBY_AGG / GROUPBY_NUMERIC_AGG (src/pd_core_macros.h:5-147): for every group, gather the group's rows
 * (Grouper::ApplyGroupings, src/dataframe.cpp:1546) then CallFunction(kind) on that array. */
int orc_groupby_agg_f64(int kind, const int64_t* offsets, const int64_t* rows, int64_t G, const double* v, const uint8_t* valid,
                        int64_t off, double* out_f64, int64_t* out_i64, uint8_t* out_valid, int nthreads) {
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int64_t g = 0; g < G; ++g) {
    int64_t len = offsets[g + 1] - offsets[g];
    double* gv = (double*)malloc((size_t)(len ? len : 1) * sizeof(double));
    uint8_t* gb = valid ? (uint8_t*)calloc((size_t)(len / 8 + 1), 1) : NULL;
    for (int64_t j = 0; j < len; ++j) {
      int64_t r = rows[offsets[g] + j];
      gv[j] = v[off + r];
      if (gb) bit_set_to(gb, j, bit_get(valid, off + r));
    }
    int64_t cnt = 0;
    double s = 0, lo = 0, hi = 0;
    switch (kind) {
      case ORC_AGG_SUM: orc_sum_f64(gv, gb, 0, len, &s, &cnt); out_f64[g] = s; break;
      case ORC_AGG_MEAN: orc_mean_f64(gv, gb, 0, len, &s, &cnt); out_f64[g] = s; break;
      case ORC_AGG_MIN: orc_minmax_f64(gv, gb, 0, len, &lo, &hi, &cnt); out_f64[g] = lo; break;
      case ORC_AGG_MAX: orc_minmax_f64(gv, gb, 0, len, &lo, &hi, &cnt); out_f64[g] = hi; break;
      case ORC_AGG_COUNT: cnt = orc_count(gb, 0, len); out_i64[g] = cnt; break;
      case ORC_AGG_VARIANCE: orc_var_f64(gv, gb, 0, len, 0, &s, &cnt); out_f64[g] = s; break;
      case ORC_AGG_STDDEV: orc_var_f64(gv, gb, 0, len, 1, &s, &cnt); out_f64[g] = s; break;
      case ORC_AGG_PRODUCT: orc_product_f64(gv, gb, 0, len, &s, &cnt); out_f64[g] = s; break;
      /* GroupBy::first / last (src/dataframe.cpp:1698-1810): the group's first / last ROW, null if that row is null */
      case ORC_AGG_FIRST: cnt = len > 0 && (!gb || bit_get(gb, 0)); out_f64[g] = len > 0 ? gv[0] : 0.0; break;
      case ORC_AGG_LAST: cnt = len > 0 && (!gb || bit_get(gb, len - 1)); out_f64[g] = len > 0 ? gv[len - 1] : 0.0; break;
    }
    if (out_valid) out_valid[g] = (kind == ORC_AGG_COUNT) ? 1 : (cnt > 0);
    free(gv);
    free(gb);
  }
  return ORC_OK;
}
int orc_groupby_agg_i64(int kind, const int64_t* offsets, const int64_t* rows, int64_t G, const int64_t* v, const uint8_t* valid,
                        int64_t off, double* out_f64, int64_t* out_i64, uint8_t* out_valid, int nthreads) {
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int64_t g = 0; g < G; ++g) {
    int64_t len = offsets[g + 1] - offsets[g];
    int64_t* gv = (int64_t*)malloc((size_t)(len ? len : 1) * sizeof(int64_t));
    uint8_t* gb = valid ? (uint8_t*)calloc((size_t)(len / 8 + 1), 1) : NULL;
    for (int64_t j = 0; j < len; ++j) {
      int64_t r = rows[offsets[g] + j];
      gv[j] = v[off + r];
      if (gb) bit_set_to(gb, j, bit_get(valid, off + r));
    }
    int64_t cnt = 0, s = 0, lo = 0, hi = 0;
    double m = 0;
    switch (kind) {
      case ORC_AGG_SUM: orc_sum_i64(gv, gb, 0, len, &s, &cnt); out_i64[g] = s; break;
      case ORC_AGG_MEAN: orc_mean_i64(gv, gb, 0, len, &m, &cnt); out_f64[g] = m; break;
      case ORC_AGG_MIN: orc_minmax_i64(gv, gb, 0, len, &lo, &hi, &cnt); out_i64[g] = lo; break;
      case ORC_AGG_MAX: orc_minmax_i64(gv, gb, 0, len, &lo, &hi, &cnt); out_i64[g] = hi; break;
      case ORC_AGG_COUNT: cnt = orc_count(gb, 0, len); out_i64[g] = cnt; break;
      case ORC_AGG_VARIANCE:
      case ORC_AGG_STDDEV: {
        double* d = (double*)malloc((size_t)(len ? len : 1) * sizeof(double));
        for (int64_t j = 0; j < len; ++j) d[j] = (double)gv[j];
        orc_var_f64(d, gb, 0, len, kind == ORC_AGG_STDDEV, &m, &cnt);
        free(d);
        out_f64[g] = m;
        break;
      }
      case ORC_AGG_PRODUCT: orc_product_i64(gv, gb, 0, len, &s, &cnt); out_i64[g] = s; break;
      case ORC_AGG_FIRST: cnt = len > 0 && (!gb || bit_get(gb, 0)); out_i64[g] = len > 0 ? gv[0] : 0; break;
      case ORC_AGG_LAST: cnt = len > 0 && (!gb || bit_get(gb, len - 1)); out_i64[g] = len > 0 ? gv[len - 1] : 0; break;
    }
    if (out_valid) out_valid[g] = (kind == ORC_AGG_COUNT) ? 1 : (cnt > 0);
    free(gv);
    free(gb);
  }
  return ORC_OK;
}

/* The reference's whole group-by call sequence for df.group_by("k").{sum,mean,count}("v"), no nulls:
 *   Grouper::Consume -> MakeGroupings -> ApplyGroupings(index, key col, value col) (src/dataframe.cpp:1571-1600)
 *   -> per group CallFunction("sum"), ("mean"), ("count") (src/pd_core_macros.h:5-147; TBB over groups = OpenMP here).
 * It omits the reference's unordered_map<ScalarPtr,...> bookkeeping, so it flatters the reference. */
int64_t orc_groupby_sum_mean_count(const int64_t* keys, const double* vals, int64_t n, int64_t* out_keys, double* out_sum,
                                   double* out_mean, int64_t* out_count, int nthreads) {
  uint32_t* ids = (uint32_t*)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
  int64_t* uniq = (int64_t*)malloc((size_t)(n ? n : 1) * sizeof(int64_t));
  int64_t G = orc_group_ids_i64(keys, NULL, 0, n, ids, uniq, NULL, NULL);
  int64_t* offsets = (int64_t*)malloc((size_t)(G + 1) * sizeof(int64_t));
  int64_t* rows = (int64_t*)malloc((size_t)(n ? n : 1) * sizeof(int64_t));
  orc_make_groupings(ids, n, G, offsets, rows);
  /* ApplyGroupings of the index (uint64 range), the key column and the value column: three gathers */
  uint64_t* g_index = (uint64_t*)malloc((size_t)(n ? n : 1) * sizeof(uint64_t));
  int64_t* g_keys = (int64_t*)malloc((size_t)(n ? n : 1) * sizeof(int64_t));
  double* g_vals = (double*)malloc((size_t)(n ? n : 1) * sizeof(double));
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int64_t j = 0; j < n; ++j) {
    int64_t r = rows[j];
    g_index[j] = (uint64_t)r;
    g_keys[j] = keys[r];
    g_vals[j] = vals[r];
  }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int64_t g = 0; g < G; ++g) {
    int64_t len = offsets[g + 1] - offsets[g], c1, c2;
    double s, m;
    orc_sum_f64(g_vals + offsets[g], NULL, 0, len, &s, &c1);  /* sum   */
    orc_mean_f64(g_vals + offsets[g], NULL, 0, len, &m, &c2); /* mean  */
    out_sum[g] = s;
    out_mean[g] = m;
    out_count[g] = len; /* count */
    out_keys[g] = uniq[g];
  }
  free(ids); free(uniq); free(offsets); free(rows); free(g_index); free(g_keys); free(g_vals);
  return G;
}

/* ------------------------------------------------------------------ resample */
static int64_t floor_div(int64_t a, int64_t b) {
  int64_t q = a / b, r = a % b;
  return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}
#define NS_PER_DAY 86400000000000LL

/* adjustDatesAnchored (src/resample.cpp:85-178), tz == "" path */
int orc_adjust_dates_anchored(int64_t min_ns, int64_t max_ns, int64_t freq_ns, int closed_right, int origin_type,
                              int64_t origin_custom_ns, int64_t offset_ns, int64_t* first_out, int64_t* last_out) {
  int64_t first = min_ns, last = max_ns, origin = 0;
  switch (origin_type) {
    case ORC_ORIGIN_EPOCH: origin = 0; break;
    case ORC_ORIGIN_START_DAY: origin = floor_div(first, NS_PER_DAY) * NS_PER_DAY; break;
    case ORC_ORIGIN_START: origin = first; break;
    case ORC_ORIGIN_END: origin = last; break;
    case ORC_ORIGIN_END_DAY: origin = floor_div(last, NS_PER_DAY) * NS_PER_DAY; break;
    default: origin = origin_custom_ns; break;
  }
  origin += offset_ns;
  int64_t foffset = (first - origin) % freq_ns; /* C++ % : sign follows the dividend */
  int64_t loffset = (last - origin) % freq_ns;
  if (closed_right) {
    if (foffset > 0) first -= foffset;
    else first -= freq_ns;
    if (loffset > 0) last += freq_ns - loffset;
  } else {
    if (foffset > 0) first -= foffset;
    if (loffset > 0) last += freq_ns - loffset;
    else last += freq_ns;
  }
  *first_out = first;
  *last_out = last;
  return ORC_OK;
}

/* makeGroupInfo for a fixed-duration rule (src/resample.cpp:202-295):
 *   MinMax -> adjustDatesAnchored -> date_range edges (src/core.cpp:308-331) -> generate_bins_dt64 (src/resample.cpp:11-83)
 *   -> label slicing (269-292). Requires sorted, null-free timestamps. Returns #bins, or -1 invalid length, -2 value before
 *   first bin, -3 value after last bin, -4 cap too small, -5 start >= end. */
int64_t orc_resample_group_info(const int64_t* ts, int64_t n, int64_t freq_ns, int closed_right, int label_right, int origin_type,
                                int64_t origin_custom_ns, int64_t offset_ns, int64_t* bins, int64_t* labels, int64_t cap) {
  if (n == 0) return 0;
  int64_t mn = ts[0], mx = ts[0];
  for (int64_t i = 1; i < n; ++i) {
    if (ts[i] < mn) mn = ts[i];
    if (ts[i] > mx) mx = ts[i];
  }
  int64_t first, last;
  orc_adjust_dates_anchored(mn, mx, freq_ns, closed_right, origin_type, origin_custom_ns, offset_ns, &first, &last);
  if (first >= last || freq_ns <= 0) return -5;
  int64_t nedges = (last - first) / freq_ns + 1; /* edges first + k*freq <= last */
  if (nedges - 1 > cap) return -4;
  if (nedges <= 0) return -1;
  if (ts[0] < first) return -2;
  if (ts[n - 1] > first + (nedges - 1) * freq_ns) return -3;
  int64_t j = 0, nb = nedges - 1;
  for (int64_t i = 0; i < nb; ++i) {
    int64_t r_bin = first + (i + 1) * freq_ns;
    if (closed_right) while (j < n && ts[j] <= r_bin) ++j;
    else while (j < n && ts[j] < r_bin) ++j;
    bins[i] = j;
  }
  /* labels = edges, sliced by one when label_right; truncated to bins.size() */
  for (int64_t i = 0; i < nb; ++i) labels[i] = first + (i + (label_right ? 1 : 0)) * freq_ns;
  return nb;
}
/* GroupInfo::downsample (src/resample.h:19-43) */
void orc_resample_expand(const int64_t* bins, const int64_t* labels, int64_t nb, int64_t* row_labels) {
  int64_t last = 0;
  for (int64_t b = 0; b < nb; ++b) {
    for (int64_t i = last; i < bins[b]; ++i) row_labels[i] = labels[b];
    last = bins[b];
  }
}

/* ------------------------------------------------------------------ concat rows (src/concat.cpp:135-189) */
void orc_concat_64(const uint64_t* const* parts, const uint8_t* const* valids, const int64_t* offs, const int64_t* lens,
                   int nparts, uint64_t* out, uint8_t* out_valid, int64_t* out_nulls) {
  int64_t o = 0, nulls = 0;
  for (int p = 0; p < nparts; ++p) {
    for (int64_t i = 0; i < lens[p]; ++i) {
      int ok = valids && valids[p] ? bit_get(valids[p], offs[p] + i) : 1;
      out[o] = parts[p][offs[p] + i];
      if (out_valid) bit_set_to(out_valid, o, ok);
      nulls += !ok;
      ++o;
    }
  }
  if (out_nulls) *out_nulls = nulls;
}

/* ------------------------------------------------------------------ temporal rounding: DataFrame::downsample (src/dataframe.cpp:1265-1290)
 * arrow::compute::FloorTemporal / CeilTemporal(index, RoundTemporalOptions(multiple, unit, week_starts_monday,
 * ceil_is_strictly_greater = false, calendar_based_origin)) on timestamp[ns] without a time zone.  Restates Arrow C++ 25.0.0
 * (scalar_temporal_unary.cc: FloorTimePoint / FloorWeekTimePoint / GetFlooredYmd / Ceil*), pinned by oracle/gen_golden_r2.py:
 *   - ns..day, multiple == 1                : floor to the unit since the epoch (toward -inf)
 *   - ns..day, calendar_based_origin        : origin = floor to the next larger unit (day: the 1st of the month), then whole
 *                                             multiples of the unit since that origin
 *   - ns..day otherwise                     : whole multiples of (multiple x unit) since the epoch (toward -inf)
 *   - week                                  : weeks start Monday (or Sunday); with a calendar origin and multiple > 1 the origin is
 *                                             the Monday (Sunday) after the last Thursday (Wednesday) of the previous December,
 *                                             the year taken from t + 3 (4) days, and the result is NOT shifted back (Arrow's own
 *                                             behaviour, so a floor may lie after t)
 *   - month / quarter                       : first day of the month, multiples counted from 1970-01 (or from January of the year)
 *   - ceil                                  : floor if floor >= t, else floor + multiple x unit; month / quarter: ALWAYS floor + multiple
 *                                             (Arrow ignores ceil_is_strictly_greater there) */
static void civil_from_days(int64_t z, int64_t* y, int* m, int* d) {
  z += 719468;
  const int64_t era = (z >= 0 ? z : z - 146096) / 146097;
  const int64_t doe = z - era * 146097;
  const int64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  const int64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  const int64_t mp = (5 * doy + 2) / 153;
  *d = (int)(doy - (153 * mp + 2) / 5 + 1);
  *m = (int)(mp < 10 ? mp + 3 : mp - 9);
  *y = yoe + era * 400 + (*m <= 2);
}
static int64_t days_from_civil(int64_t y, int m, int d) {
  y -= m <= 2;
  const int64_t era = (y >= 0 ? y : y - 399) / 400;
  const int64_t yoe = y - era * 400;
  const int64_t doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
  const int64_t doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  return era * 146097 + doe - 719468;
}
static const int64_t kUnitNs[7] = {1LL, 1000LL, 1000000LL, 1000000000LL, 60000000000LL, 3600000000000LL, 86400000000000LL};

static int64_t floor_temporal_1(int64_t t, int64_t mult, int unit, int wsm, int cbo) {
  if (unit <= ORC_UNIT_DAY) {
    const int64_t u = kUnitNs[unit];
    if (mult == 1) return floor_div(t, u) * u;
    if (cbo) {
      int64_t origin;
      if (unit == ORC_UNIT_DAY) {
        int64_t y;
        int m, d;
        civil_from_days(floor_div(t, NS_PER_DAY), &y, &m, &d);
        origin = days_from_civil(y, m, 1) * NS_PER_DAY;
      } else {
        origin = floor_div(t, kUnitNs[unit + 1]) * kUnitNs[unit + 1];
      }
      const int64_t p = mult * u;
      return (t - origin) / p * p + origin; /* t >= origin */
    }
    return floor_div(floor_div(t, u), mult) * mult * u;
  }
  if (unit == ORC_UNIT_WEEK) {
    const int64_t org = (wsm ? 3 : 4) * NS_PER_DAY, W = 7 * NS_PER_DAY, tt = t + org;
    if (mult == 1) return floor_div(tt, W) * W - org;
    if (cbo) {
      int64_t y;
      int m, d;
      civil_from_days(floor_div(tt, NS_PER_DAY), &y, &m, &d);
      const int64_t dec31 = days_from_civil(y - 1, 12, 31);
      const int64_t wd = ((dec31 + 4) % 7 + 7) % 7; /* 0 = Sunday; 1970-01-01 was a Thursday */
      const int64_t target = wsm ? 4 : 3;             /* Thursday / Wednesday */
      const int64_t last = dec31 - (((wd - target) % 7 + 7) % 7);
      const int64_t start = (last + 4) * NS_PER_DAY;  /* date.h: (mon - thu) is 4 days modulo 7 */
      const int64_t p = mult * W;
      return (tt - start) / p * p + start;            /* C++ truncating division: tt may precede start by a few days */
    }
    return floor_div(floor_div(tt, W), mult) * mult * W - org;
  }
  /* month / quarter */
  const int64_t mm = mult * (unit == ORC_UNIT_QUARTER ? 3 : 1);
  int64_t y;
  int m, d;
  civil_from_days(floor_div(t, NS_PER_DAY), &y, &m, &d);
  if (mm == 1) return days_from_civil(y, m, 1) * NS_PER_DAY;
  if (cbo) return days_from_civil(y, 1 + (int)((m - 1) / mm * mm), 1) * NS_PER_DAY;
  int64_t tm = floor_div((y - 1970) * 12 + m - 1, mm) * mm;
  return days_from_civil(1970 + floor_div(tm, 12), (int)(tm - floor_div(tm, 12) * 12) + 1, 1) * NS_PER_DAY;
}
static int64_t ceil_temporal_1(int64_t t, int64_t mult, int unit, int wsm, int cbo) {
  const int64_t f = floor_temporal_1(t, mult, unit, wsm, cbo);
  if (unit <= ORC_UNIT_DAY) return f >= t ? f : f + mult * kUnitNs[unit];
  if (unit == ORC_UNIT_WEEK) return f >= t ? f : f + mult * 7 * NS_PER_DAY;
  const int64_t mm = mult * (unit == ORC_UNIT_QUARTER ? 3 : 1);
  int64_t y;
  int m, d;
  civil_from_days(floor_div(f, NS_PER_DAY), &y, &m, &d);
  const int64_t tm = y * 12 + m - 1 + mm;
  return days_from_civil(floor_div(tm, 12), (int)(tm - floor_div(tm, 12) * 12) + 1, 1) * NS_PER_DAY;
}
int orc_round_temporal(int ceil_mode, const int64_t* ts, const uint8_t* valid, int64_t off, int64_t n, int64_t multiple, int unit,
                       int week_starts_monday, int calendar_based_origin, int64_t* out, uint8_t* out_valid) {
  if (multiple < 1 || unit < ORC_UNIT_NANOSECOND || unit > ORC_UNIT_QUARTER) return ORC_INVALID;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t t = ts[off + i];
    out[i] = ceil_mode ? ceil_temporal_1(t, multiple, unit, week_starts_monday, calendar_based_origin)
                       : floor_temporal_1(t, multiple, unit, week_starts_monday, calendar_based_origin);
  }
  if (out_valid && n > 0) {
    memset(out_valid, 0, (size_t)((n + 7) / 8));
    for (int64_t i = 0; i < n; ++i) bit_set_to(out_valid, i, is_valid(valid, off, i));
  }
  return ORC_OK;
}

/* ------------------------------------------------------------------ group-by all / any / count_distinct
 * GROUPBY_NUMERIC_AGG(all | any, bool), GROUPBY_NUMERIC_AGG(count_distinct, int64_t) (src/dataframe.cpp:1520-1526): one
 * CallFunction(name, {group}, nullptr) per group -> Arrow defaults: all / any skip nulls and are null without a valid value
 * (ScalarAggregateOptions{skip_nulls = true, min_count = 1}); count_distinct counts distinct VALID values (CountOptions
 * ONLY_VALID), two doubles being the same value iff their bit patterns are (memo table hashed on the bytes). */
int orc_groupby_all_any(const uint32_t* ids, int64_t n, int64_t G, const uint8_t* bits, const uint8_t* valid, int64_t off,
                        uint8_t* out_all, uint8_t* out_any, uint8_t* out_valid) {
  for (int64_t g = 0; g < G; ++g) {
    out_all[g] = 1;
    out_any[g] = 0;
    out_valid[g] = 0;
  }
  for (int64_t i = 0; i < n; ++i) {
    if (!is_valid(valid, off, i)) continue;
    const uint32_t g = ids[i];
    const int b = bit_get(bits, off + i);
    out_valid[g] = 1;
    if (b) out_any[g] = 1;
    else out_all[g] = 0;
  }
  return ORC_OK;
}
typedef struct { uint32_t g; uint64_t v; } gv_pair;
static int gv_cmp(const void* a, const void* b) {
  const gv_pair *x = (const gv_pair*)a, *y = (const gv_pair*)b;
  if (x->g != y->g) return x->g < y->g ? -1 : 1;
  if (x->v != y->v) return x->v < y->v ? -1 : 1;
  return 0;
}
int orc_groupby_count_distinct(const uint32_t* ids, int64_t n, int64_t G, const uint64_t* v, const uint8_t* valid, int64_t off,
                               int64_t* out) {
  for (int64_t g = 0; g < G; ++g) out[g] = 0;
  gv_pair* p = (gv_pair*)malloc((size_t)(n > 0 ? n : 1) * sizeof(gv_pair));
  if (!p) return ORC_INVALID;
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i)
    if (is_valid(valid, off, i)) {
      p[m].g = ids[i];
      p[m].v = v[off + i];
      ++m;
    }
  qsort(p, (size_t)m, sizeof(gv_pair), gv_cmp);
  for (int64_t i = 0; i < m; ++i)
    if (i == 0 || p[i].g != p[i - 1].g || p[i].v != p[i - 1].v) out[p[i].g] += 1;
  free(p);
  return ORC_OK;
}

/* ---------------------------------------------------------------- functions of one column
 * Arrow 25.0.0 scalar arithmetic kernels (unchecked variants, as CallFunction(name) without options selects them):
 * negate / abs wrap for integers; sign = isnan(x) ? x : (x == 0 ? 0 : (signbit(x) ? -1 : 1)) for floats and (x > 0) - (x < 0)
 * for integers; sqrt / exp / power run in float64 after the implicit SAFE cast of integer input (values outside +-2^53 fail);
 * sqrt / exp / power call the host libm (std::sqrt / std::exp / std::pow), so exp and power are only as reproducible as it. */
int orc_unary(int op, int dtype, const uint64_t* in_bits, const uint8_t* valid, int64_t off, int64_t n, double expo, uint64_t* out_bits,
              uint64_t* bad) {
  const int to_f64 = op == 3 || op == 4 || op == 100;
  if (op == 5 && dtype == 2) return -1;
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t u = in_bits[i];
    uint64_t r = 0;
    if (to_f64) {
      double d;
      if (dtype == 2) {
        memcpy(&d, &u, 8);
      } else {
        int out_of_range = dtype == 1 ? u > (1ull << 53) : ((int64_t)u > (1ll << 53) || (int64_t)u < -(1ll << 53));
        if (out_of_range && (!valid || ((valid[(off + i) >> 3] >> ((off + i) & 7)) & 1))) {
          *bad = u;
          return 1;
        }
        d = dtype == 1 ? (double)u : (double)(int64_t)u;
      }
      /* SquareRoot::Call: `if (arg < 0.0) return quiet_NaN()` -- the POSITIVE quiet NaN, not libm's -nan for a negative operand */
      const double y = op == 3 ? (d < 0.0 ? (double)NAN : sqrt(d)) : op == 4 ? exp(d) : pow(d, expo);
      memcpy(&r, &y, 8);
    } else if (dtype == 2) {
      double d, y;
      memcpy(&d, &u, 8);
      if (op == 0) y = -d;
      else if (op == 1) y = fabs(d);
      else y = d != d ? d : (d == 0.0 ? 0.0 : (signbit(d) ? -1.0 : 1.0));
      memcpy(&r, &y, 8);
    } else if (op == 0) {
      r = 0ull - u;
    } else if (op == 1) {
      r = (dtype == 0 && (int64_t)u < 0) ? 0ull - u : u;
    } else if (op == 2) {
      r = dtype == 1 ? (uint64_t)(u != 0) : (uint64_t)(int64_t)(((int64_t)u > 0) - ((int64_t)u < 0));
    } else {
      r = ~u;
    }
    out_bits[i] = r;
  }
  return 0;
}

void orc_if_else(const uint8_t* cond_bits, const uint8_t* cond_valid, const uint64_t* a, const uint8_t* va, int a_scalar, const uint64_t* b,
                 const uint8_t* vb, int b_scalar, int64_t n, uint64_t* out, uint8_t* out_valid) {
  memset(out_valid, 0, (size_t)((n + 7) / 8));
  for (int64_t i = 0; i < n; ++i) {
    const int c = (cond_bits[i >> 3] >> (i & 7)) & 1;
    const int cv = !cond_valid || ((cond_valid[i >> 3] >> (i & 7)) & 1);
    const int64_t ia = a_scalar ? 0 : i, ib = b_scalar ? 0 : i;
    const int av = !va || ((va[ia >> 3] >> (ia & 7)) & 1), bv = !vb || ((vb[ib >> 3] >> (ib & 7)) & 1);
    out[i] = c ? a[ia] : b[ib];
    if (cv && (c ? av : bv)) out_valid[i >> 3] |= (uint8_t)(1u << (i & 7));
  }
}
