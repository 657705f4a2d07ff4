/*
 * pdx_oracle.h -- CPU ORACLE for the PandasArrow vectorized-operator hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the shipped path (libpdx_hip.so) never
 * links, imports or falls back to anything in oracle/.
 *
 * It is a plain-C restatement of what the reference computes on this path.  The
 * reference (EPOCHDevs/PandasArrow) owns no arithmetic loops here: every numeric op is
 * a forward to Apache Arrow C++ compute (third-party dependency, version NOT pinned by
 * the reference -- CMakeLists.txt:36 `find_package(Arrow CONFIG REQUIRED)`).  The
 * algorithms below therefore restate Arrow C++ 25.0.0 behaviour (the version available
 * in this image through the pyarrow wheel), anchored on the reference call sites cited
 * next to each function (file:line relative to /root/reference).
 *
 * Parity pinning: oracle/gen_golden.py runs Arrow 25.0.0 (pyarrow) over seeded inputs
 * and the reference's own known-answer test vectors and freezes inputs + expected
 * outputs under tests/golden/; tests/test_oracle_golden.py checks every function here
 * against those fixtures bit-for-bit.
 */
#ifndef PDX_ORACLE_H
#define PDX_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes shared with include/pdx/abi.h */
#define ORC_OK 0
#define ORC_INVALID 1      /* ArrowInvalid: type/length mismatch, integer divide by zero */
#define ORC_INDEX_ERROR 2  /* ArrowIndexError: take out of bounds */

/* op codes (same numbering as include/pdx/abi.h) */
enum { ORC_ADD = 0, ORC_SUB = 1, ORC_MUL = 2, ORC_DIV = 3 };
enum { ORC_EQ = 0, ORC_NE = 1, ORC_LT = 2, ORC_LE = 3, ORC_GT = 4, ORC_GE = 5 };
enum { ORC_AND = 0, ORC_OR = 1 };
enum { ORC_AGG_SUM = 0, ORC_AGG_MEAN = 1, ORC_AGG_MIN = 2, ORC_AGG_MAX = 3, ORC_AGG_COUNT = 4,
       ORC_AGG_VARIANCE = 5, ORC_AGG_STDDEV = 6, ORC_AGG_PRODUCT = 7, ORC_AGG_FIRST = 8, ORC_AGG_LAST = 9 };

/* ---- synthetic inputs (counter based) ---- */
uint64_t orc_splitmix64(uint64_t x);
void orc_synth_keys(int64_t start, int64_t n, int64_t num_keys, int64_t* out);
void orc_synth_vals(int64_t start, int64_t n, uint64_t seed_off, double* out);
void orc_synth_ts(int64_t start, int64_t n, int64_t t0_ns, int64_t step_ns, int64_t* out);

/* ---- whole-array aggregates: NDFrame::sum/mean/min/max/count (src/ndframe.cpp:26-31,119,162-166,220) ---- */
/* validity may be NULL (all valid); bit i of the array is validity bit (off + i), values[off + i]. */
int orc_sum_f64(const double* v, const uint8_t* valid, int64_t off, int64_t n, double* out, int64_t* count);
int orc_sum_i64(const int64_t* v, const uint8_t* valid, int64_t off, int64_t n, int64_t* out, int64_t* count);
int orc_mean_f64(const double* v, const uint8_t* valid, int64_t off, int64_t n, double* out, int64_t* count);
int orc_mean_i64(const int64_t* v, const uint8_t* valid, int64_t off, int64_t n, double* out, int64_t* count);
int orc_minmax_f64(const double* v, const uint8_t* valid, int64_t off, int64_t n, double* mn, double* mx, int64_t* count);
int orc_minmax_i64(const int64_t* v, const uint8_t* valid, int64_t off, int64_t n, int64_t* mn, int64_t* mx, int64_t* count);
int64_t orc_count(const uint8_t* valid, int64_t off, int64_t n);

/* ---- element-wise: Series::operator{+,-,*,/} (src/series.cpp:19-33,229-235), DataFrame (src/dataframe.cpp:233-275) ---- */
/* b_is_scalar == 1: b points to ONE value (Scalar rhs, series.cpp:25-28); == 2: a points to ONE value (Scalar lhs:
 * Scalar::operator op(Series) -> CallFunction(name, {scalar, array}), src/scalar.cpp:24-56; n is b's length).
 * out_valid may be NULL when both inputs have no validity. */
int orc_binary_f64(int op, const double* a, const uint8_t* va, int64_t aoff, const double* b, const uint8_t* vb, int64_t boff,
                   int b_is_scalar, int64_t n, double* out, uint8_t* out_valid);
int orc_binary_i64(int op, const int64_t* a, const uint8_t* va, int64_t aoff, const int64_t* b, const uint8_t* vb, int64_t boff,
                   int b_is_scalar, int64_t n, int64_t* out, uint8_t* out_valid);
/* comparisons -> bit-packed bools (src/series.cpp:247-257) */
int orc_compare_f64(int op, const double* a, const uint8_t* va, int64_t aoff, const double* b, const uint8_t* vb, int64_t boff,
                    int b_is_scalar, int64_t n, uint8_t* out_bits, uint8_t* out_valid);
int orc_compare_i64(int op, const int64_t* a, const uint8_t* va, int64_t aoff, const int64_t* b, const uint8_t* vb, int64_t boff,
                    int b_is_scalar, int64_t n, uint8_t* out_bits, uint8_t* out_valid);
/* non-Kleene and/or, invert on bit-packed bools (src/series.cpp:259-261,319) */
int orc_logical(int op, const uint8_t* a, const uint8_t* va, int64_t aoff, const uint8_t* b, const uint8_t* vb, int64_t boff,
                int64_t n, uint8_t* out_bits, uint8_t* out_valid);
void orc_invert(const uint8_t* a, int64_t aoff, int64_t n, uint8_t* out_bits);
/* arrow::compute::IfElse(cond, a, b) (src/series.cpp:1203-1209): 8-byte patterns; a_scalar / b_scalar: that operand has length 1.
 * out_valid (bit-packed, n bits) = cond_valid & (cond ? a_valid : b_valid). */
void orc_if_else(const uint8_t* cond_bits, const uint8_t* cond_valid, const uint64_t* a, const uint8_t* va, int a_scalar, const uint64_t* b,
                 const uint8_t* vb, int b_scalar, int64_t n, uint64_t* out, uint8_t* out_valid);
/* CallFunction("negate" | "abs" | "sign" | "sqrt" | "exp" | "bit_wise_not" | "power") on one array (src/dataframe.cpp:251-275,
 * 919-935): op 0..5 as pdx_unary_op, 100 = power(a, expo).  dtype: 0 int64, 1 uint64, 2 float64 (in_bits: the 8-byte patterns).
 * out_bits: 8-byte patterns of the result (its type follows Arrow, integers' sign as int64).  Validity passes through (the
 * caller copies it); `valid` is only consulted by the integer -> float64 cast check.  Returns 0, or 1 when a valid integer
 * lies outside +-2^53 (*bad = that value), or -1 for a combination Arrow has no kernel for. */
int orc_unary(int op, int dtype, const uint64_t* in_bits, const uint8_t* valid, int64_t off, int64_t n, double expo, uint64_t* out_bits,
              uint64_t* bad);
void orc_validity_and(const uint8_t* va, int64_t aoff, const uint8_t* vb, int64_t boff, int64_t n, uint8_t* out_valid);

/* ---- filter / take on 8-byte columns (src/dataframe.cpp:461-492, src/series.cpp:130-159) ---- */
/* mask: bit-packed bools + optional validity; emit_null!=0 => FilterOptions::EMIT_NULL. returns output length. */
int64_t orc_filter_count(const uint8_t* mask, const uint8_t* mask_valid, int64_t moff, int64_t n, int emit_null);
int orc_filter_64(const uint64_t* v, const uint8_t* valid, int64_t off, const uint8_t* mask, const uint8_t* mask_valid,
                  int64_t moff, int64_t n, int emit_null, uint64_t* out, uint8_t* out_valid, int64_t* out_nulls);
int orc_take_64(const uint64_t* v, const uint8_t* valid, int64_t off, int64_t n, const int64_t* idx, const uint8_t* idx_valid,
                int64_t ioff, int64_t m, uint64_t* out, uint8_t* out_valid, int64_t* out_nulls, int64_t* bad_index);

/* ---- group-by (src/group_by.h:24-31, src/dataframe.cpp:1539-1600, src/pd_core_macros.h:5-147) ---- */
/* Grouper::Consume: dense uint32 ids in first-occurrence order; a null key is its own group.
 * uniques/unique_is_null/first_row sized >= n; returns number of groups. */
int64_t orc_group_ids_i64(const int64_t* keys, const uint8_t* valid, int64_t off, int64_t n, uint32_t* ids,
                          int64_t* uniques, uint8_t* unique_is_null, int64_t* first_row);
/* Grouper::MakeGroupings: stable counting sort of row ids by group. offsets has G+1 entries. */
void orc_make_groupings(const uint32_t* ids, int64_t n, int64_t G, int64_t* offsets, int64_t* rows);
/* per-group scalar aggregate over the group's rows in row order (ApplyGroupings + CallFunction(kind)).
 * out_f64 used for SUM(f64)/MEAN/MIN/MAX(f64); out_i64 for SUM(i64)/COUNT/MIN/MAX(i64). out_valid: 1 byte per group. */
int orc_groupby_agg_f64(int kind, const int64_t* offsets, const int64_t* rows, int64_t G, const double* v, const uint8_t* valid,
                        int64_t off, double* out_f64, int64_t* out_i64, uint8_t* out_valid, int nthreads);
int orc_groupby_agg_i64(int kind, const int64_t* offsets, const int64_t* rows, int64_t G, const int64_t* v, const uint8_t* valid,
                        int64_t off, double* out_f64, int64_t* out_i64, uint8_t* out_valid, int nthreads);
/* the whole reference sequence for one int64 key column + one f64 value column, returning sum/mean/count:
 * used as the timed CPU baseline ("port" of the reference's Arrow-CPU path). Returns G. */
int64_t orc_groupby_sum_mean_count(const int64_t* keys, const double* vals, int64_t n, int64_t* out_keys, double* out_sum,
                                   double* out_mean, int64_t* out_count, int nthreads);

/* ---- resample (src/resample.cpp:11-83,85-178,202-295; src/resample.h:19-43; src/core.cpp:308-331) ---- */
enum { ORC_ORIGIN_EPOCH = 0, ORC_ORIGIN_START_DAY = 1, ORC_ORIGIN_START = 2, ORC_ORIGIN_END = 3, ORC_ORIGIN_END_DAY = 4, ORC_ORIGIN_CUSTOM = 5 };
int orc_adjust_dates_anchored(int64_t min_ns, int64_t max_ns, int64_t freq_ns, int closed_right, int origin_type,
                              int64_t origin_custom_ns, int64_t offset_ns, int64_t* first, int64_t* last);
/* sorted ts (no nulls) -> bins (cumulative counts, one per edge interval) + labels; returns number of bins kept (or <0 on error).
 * bins/labels sized >= (last-first)/freq + 2. */
int64_t orc_resample_group_info(const int64_t* ts, int64_t n, int64_t freq_ns, int closed_right, int label_right, int origin_type,
                                int64_t origin_custom_ns, int64_t offset_ns, int64_t* bins, int64_t* labels, int64_t cap);
/* GroupInfo::downsample: one label per row */
void orc_resample_expand(const int64_t* bins, const int64_t* labels, int64_t nb, int64_t* row_labels);

/* ---- concat rows (src/concat.cpp:116-190): values + validity stitching ---- */
void orc_concat_64(const uint64_t* const* parts, const uint8_t* const* valids, const int64_t* offs, const int64_t* lens,
                   int nparts, uint64_t* out, uint8_t* out_valid, int64_t* out_nulls);

/* ---- group-by all / any (bit-packed bool values) and count_distinct (8-byte values as bit patterns): src/dataframe.cpp:1520-1526.
 * outputs: one byte per group; out_valid[g] == 0 <=> the group has no valid value (all / any are null there). */
int orc_groupby_all_any(const uint32_t* ids, int64_t n, int64_t G, const uint8_t* bits, const uint8_t* valid, int64_t off,
                        uint8_t* out_all, uint8_t* out_any, uint8_t* out_valid);
int orc_groupby_count_distinct(const uint32_t* ids, int64_t n, int64_t G, const uint64_t* v, const uint8_t* valid, int64_t off,
                               int64_t* out);

/* ---- temporal rounding: DataFrame::downsample (src/dataframe.cpp:1265-1290) = Arrow floor_temporal / ceil_temporal ---- */
enum { ORC_UNIT_NANOSECOND = 0, ORC_UNIT_MICROSECOND = 1, ORC_UNIT_MILLISECOND = 2, ORC_UNIT_SECOND = 3, ORC_UNIT_MINUTE = 4,
       ORC_UNIT_HOUR = 5, ORC_UNIT_DAY = 6, ORC_UNIT_WEEK = 7, ORC_UNIT_MONTH = 8, ORC_UNIT_QUARTER = 9 };
/* ceil_mode != 0: CeilTemporal (ceil_is_strictly_greater = false, as the reference passes), else FloorTemporal. timestamp[ns], no tz. */
int orc_round_temporal(int ceil_mode, const int64_t* ts, const uint8_t* valid, int64_t off, int64_t n, int64_t multiple, int unit,
                       int week_starts_monday, int calendar_based_origin, int64_t* out, uint8_t* out_valid);

#ifdef __cplusplus
}
#endif
#endif
