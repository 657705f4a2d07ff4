/*
 * pdx/abi.h -- C ABI of libpdx_hip.so: the MI355X (gfx950) execution backend that replaces the
 * Arrow-CPU compute underneath PandasArrow's vectorized operator path.
 *
 * Drop-in boundary.  In the reference every numeric operation on this
 * path is a forward to
 *     arrow::compute::CallFunction("<kernel-name>", {Datum...}, FunctionOptions*)
 * or to arrow::compute::Grouper::{Make,Consume,MakeGroupings,ApplyGroupings,GetUniques}.
 * Each entry point below cites the reference call site(s) it replaces (file:line relative to the
 * reference repository).  INTEGRATION.md shows the binding a maintainer adds at those sites.
 *
 * Conventions
 *   - Columns are Arrow-layout raw buffers that live in DEVICE memory (HBM): a contiguous values
 *     buffer plus an optional validity bitmap (1 bit/row, LSB first), both addressed with an element
 *     `offset` exactly like a sliced arrow::ArrayData.  PDX_BOOL values are bit-packed.
 *   - Inputs are borrowed and never mutated.  Outputs whose size is known up front are written into
 *     caller-provided buffers (pdx_mut_column); data-dependent results (group-by, resample) live in
 *     an opaque handle owned by the library until pdx_*_destroy.
 *   - Every function returns a pdx_status.  No exception crosses this boundary; the C++ facade
 *     (pandasarrow_amd/cpp) rethrows std::runtime_error(pdx_last_error()) the way the reference's
 *     ReturnOrThrowOnFailure does (src/core.h:181-194).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are ordered on
 *     that stream; entry points that return host-visible results (scalars, counts) synchronise it.
 *   - Threading (the reference calls this boundary from tbb::parallel_for workers, src/pd_core_macros.h:21,56,94,122): every
 *     entry point may be called concurrently from several host threads, each on its own stream (or all on the same one).  The
 *     only shared state is the scratch pool, which is kept per device and is stream ordered: a block freed by a call on stream S
 *     is reused at once by later calls on S, and by calls on other streams / threads only after the work queued on S at the time
 *     of the free has completed.  A handle (pdx_groupby, pdx_grouped) must not be used by two threads at the same time, and calls
 *     that use one handle on different streams must be ordered by the caller (the handle's buffers are not stream-synchronised).
 *     pdx_last_error is thread local.  Tested by tests/test_gpu_threads.py.
 */
#ifndef PDX_ABI_H
#define PDX_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDX_ABI_VERSION 1

typedef enum pdx_status {
  PDX_OK = 0,
  PDX_INVALID = 1,      /* arrow::Status::Invalid: type/length mismatch, integer divide by zero, bad argument */
  PDX_INDEX_ERROR = 2,  /* arrow::Status::IndexError: take index out of bounds */
  PDX_OOM = 3,          /* device allocation failed */
  PDX_DEVICE = 4,       /* HIP runtime error */
  PDX_NOT_IMPLEMENTED = 5
} pdx_status;

typedef enum pdx_dtype {
  PDX_INT64 = 0,
  PDX_FLOAT64 = 1,
  PDX_BOOL = 2,         /* bit-packed, LSB first */
  PDX_UINT64 = 3,
  PDX_TIMESTAMP_NS = 4  /* int64 nanoseconds since epoch */
} pdx_dtype;

/* immutable input column (mirrors the fields of arrow::ArrayData the kernels read) */
typedef struct pdx_column {
  int32_t dtype;        /* pdx_dtype */
  int32_t reserved;
  int64_t length;       /* rows */
  int64_t offset;       /* element offset into values; bit offset into validity (and bool values) */
  int64_t null_count;   /* 0 => validity may be NULL; -1 => unknown */
  const void* validity; /* device pointer or NULL */
  const void* values;   /* device pointer */
} pdx_column;

/* caller-allocated output column; offset is always 0.  validity may be NULL when the caller knows the
 * result cannot contain nulls (then a would-be null is an error PDX_INVALID). */
typedef struct pdx_mut_column {
  int32_t dtype;
  int32_t reserved;
  int64_t length;       /* capacity in rows on input; rows written on output */
  int64_t null_count;   /* written by the library (exact), -1 if not computed */
  void* validity;       /* device pointer to (length+7)/8 bytes (+8 bytes slack), or NULL */
  void* values;         /* device pointer */
} pdx_mut_column;

/* host-visible scalar result (arrow::Scalar analogue) */
typedef struct pdx_scalar {
  int32_t dtype;
  int32_t is_valid;     /* 0 => null (e.g. sum of an empty / all-null array, min_count = 1) */
  union {
    int64_t i64;
    uint64_t u64;
    double f64;
  } v;
  int64_t count;        /* number of valid inputs that produced it */
} pdx_scalar;

typedef enum pdx_binary_op {
  PDX_ADD = 0, PDX_SUB = 1, PDX_MUL = 2, PDX_DIV = 3,
  /* integer operands only: BINARY_OPERATOR(| & ^ << >>) src/series.cpp:237-245, BINARY_OPERATOR_DF src/dataframe.cpp:553-561.
   * Shifts are Arrow's unchecked kernels: an amount that is negative or >= 63 (the precision of int64) returns the left operand
   * unchanged; shift_right is arithmetic; shift_left wraps. */
  PDX_BIT_OR = 4, PDX_BIT_AND = 5, PDX_BIT_XOR = 6, PDX_SHIFT_LEFT = 7, PDX_SHIFT_RIGHT = 8
} pdx_binary_op;
/* element-wise functions of one column (pdx_unary) */
typedef enum pdx_unary_op { PDX_NEGATE = 0, PDX_ABS = 1, PDX_SIGN = 2, PDX_SQRT = 3, PDX_EXP = 4, PDX_BIT_NOT = 5 } pdx_unary_op;
typedef enum pdx_compare_op { PDX_EQ = 0, PDX_NE = 1, PDX_LT = 2, PDX_LE = 3, PDX_GT = 4, PDX_GE = 5 } pdx_compare_op;
typedef enum pdx_logical_op { PDX_AND = 0, PDX_OR = 1 } pdx_logical_op;
/* which operand of pdx_binary / pdx_compare is a length-1 column that is broadcast (the `b_is_scalar` argument) */
typedef enum pdx_scalar_side { PDX_SCALAR_NONE = 0, PDX_SCALAR_RHS = 1, PDX_SCALAR_LHS = 2 } pdx_scalar_side;
typedef enum pdx_agg_kind {
  PDX_AGG_SUM = 0, PDX_AGG_MEAN = 1, PDX_AGG_MIN = 2, PDX_AGG_MAX = 3, PDX_AGG_COUNT = 4,
  /* group-by only (pdx_groupby_agg; SURVEY 8(f)-3, the reference's GROUPBY_NUMERIC_AGG(variance|stddev), GROUPBY_AGG(product),
   * GroupBy::first/last, src/dataframe.cpp:1516-1536, 1698-1810).  variance/stddev: Arrow defaults (ddof = 0, nulls skipped,
   * null for a group without valid values), two pairwise passes -> float64.  product: one multiply per valid value in row
   * order, int64 wraps, null without valid values -> value dtype.  first/last: the group's first / last ROW (null if that row
   * is null) -> value dtype. */
  PDX_AGG_VARIANCE = 5, PDX_AGG_STDDEV = 6, PDX_AGG_PRODUCT = 7, PDX_AGG_FIRST = 8, PDX_AGG_LAST = 9,
  /* GROUPBY_NUMERIC_AGG(all | any, bool) and GROUPBY_NUMERIC_AGG(count_distinct, int64_t), src/dataframe.cpp:1520-1526.
   * all / any: PDX_BOOL values -> PDX_BOOL result (nulls skipped; a group without a valid value is null, so the output needs a
   * validity buffer).  count_distinct: the number of distinct VALID values per group -> PDX_INT64, never null; float64 values
   * are distinct when their bit patterns are (0.0 / -0.0 and NaN payloads count separately, like Arrow's memo table).
   * GroupBy::min_max (src/dataframe.cpp:1602-1696) is {PDX_AGG_MIN, PDX_AGG_MAX} in one call. */
  PDX_AGG_ALL = 10, PDX_AGG_ANY = 11, PDX_AGG_COUNT_DISTINCT = 12
} pdx_agg_kind;
typedef enum pdx_origin {
  PDX_ORIGIN_EPOCH = 0, PDX_ORIGIN_START_DAY = 1, PDX_ORIGIN_START = 2, PDX_ORIGIN_END = 3, PDX_ORIGIN_END_DAY = 4, PDX_ORIGIN_CUSTOM = 5,
  /* OR-ed into origin_type by a multi-GPU caller whose `ts` is one row-range shard of a longer axis (bins made whole by the caller,
   * origin passed as PDX_ORIGIN_CUSTOM from the whole axis): the rows < bins test ("upSampling") belongs to the whole axis, which
   * the caller has already checked, so it is not applied to the shard */
  PDX_ORIGIN_SHARD = 0x100
} pdx_origin;

/* ---------------------------------------------------------------- runtime */
int pdx_abi_version(void);
/* "pdx-hip abi <version> gfx950 sources <digest>": the digest (sha256 prefix over the .hip / .hpp files under csrc and this header, computed by
 * __graft_entry__.build()) names the source tree the loaded library was compiled from; tests/test_abi_symbols.py compares it with the
 * tree, so a stale in-tree .so cannot pass for a build of the current sources. */
const char* pdx_build_info(void);
/* Select the HIP device for the calling thread and create the scratch pool.  Idempotent. */
int pdx_init(int device);
int pdx_shutdown(void);
/* Thread-local message of the last non-OK status returned on this thread ("" if none). */
const char* pdx_last_error(void);
/* Device memory helpers for callers that hand over HOST buffers (the reference builds everything from
 * host std::vector / arrow builders; cudf precedent: src/cudf/cudf_utils.h:118-141). */
int pdx_malloc(void** dptr, size_t bytes);
int pdx_free(void* dptr);
int pdx_to_device(void* dst_device, const void* src_host, size_t bytes, void* stream);
int pdx_to_host(void* dst_host, const void* src_device, size_t bytes, void* stream);
int pdx_stream_synchronize(void* stream);
/* Release cached scratch memory back to the driver. */
int pdx_trim_pool(void);

/* Optional per-kernel timing: when enabled, HIP events bracket each kernel family on its launch stream.
 * pdx_profile_report writes one "tag count total_ms" line per kernel family. Used by bench.py for the roofline line. */
int pdx_profile_enable(int on);
int pdx_profile_reset(void);
int pdx_profile_report(char* buf, size_t buf_len);

/* ---------------------------------------------------------------- synthetic inputs
 * Counter-based generators, bit-identical to oracle/pdx_oracle.c orc_synth_*; used by bench.py/tests. */
int pdx_synth_keys(int64_t start, int64_t n, int64_t num_keys, int64_t* out, void* stream);
int pdx_synth_vals(int64_t start, int64_t n, uint64_t seed_off, double* out, void* stream);
int pdx_synth_ts(int64_t start, int64_t n, int64_t t0_ns, int64_t step_ns, int64_t* out, void* stream);

/* ---------------------------------------------------------------- element-wise
 * Replaces CallFunction("add"|"subtract"|"multiply"|"divide", {lhs, rhs}) at src/series.cpp:19-33 (macro
 * BINARY_OPERATOR, instantiated 229-235), src/scalar.cpp:24-36 (Scalar rhs/lhs) and the DataFrame form
 * src/dataframe.cpp:233-275.  a,b: PDX_INT64 or PDX_FLOAT64 (mixed => float64 result, like Arrow's implicit
 * promotion).  b_is_scalar is a pdx_scalar_side: PDX_SCALAR_RHS (1): b has length 1 and is broadcast (`series - 2`,
 * src/series.cpp:25-28); PDX_SCALAR_LHS (2): a has length 1 and is broadcast (`2 - series`, `2 / series`:
 * Scalar::operator op(Series) -> BinaryImpl -> CallFunction(name, {scalar, s.array()}), src/scalar.cpp:24-36; the result has
 * b's length).  Integer arithmetic wraps; integer divide truncates toward zero, INT64_MIN / -1 == 0, a zero divisor at a valid
 * slot fails the whole call with PDX_INVALID "divide by zero".  out null where either input is null.
 * PDX_BIT_OR .. PDX_SHIFT_RIGHT take int64 operands only (PDX_NOT_IMPLEMENTED otherwise: Arrow has no floating-point kernels). */
int pdx_binary(int op, const pdx_column* a, const pdx_column* b, int b_is_scalar, pdx_mut_column* out, void* stream);
/* Replaces CallFunction("equal"|"not_equal"|"less"|"less_equal"|"greater"|"greater_equal") at
 * src/series.cpp:247-257 and, with PDX_SCALAR_LHS, Scalar::operator{>,>=,<,<=,==,!=}(Series) at src/scalar.cpp:48-56.
 * out: PDX_BOOL (bit-packed). */
int pdx_compare(int op, const pdx_column* a, const pdx_column* b, int b_is_scalar, pdx_mut_column* out, void* stream);
/* Replaces CallFunction("and"|"or") (non-Kleene) at src/series.cpp:259-260 and "invert" at src/series.cpp:319. */
int pdx_logical(int op, const pdx_column* a, const pdx_column* b, pdx_mut_column* out, void* stream);
int pdx_invert(const pdx_column* a, pdx_mut_column* out, void* stream);
/* Replaces arrow::compute::IfElse(cond, a, b): Series::if_else / Series::where(cond, other) with a Series or Scalar `other`
 * (src/series.cpp:1203-1209, 1247-1253; pinned by tests/series_indexing_test.cpp:36-52).  cond: PDX_BOOL of the result's length;
 * a, b: PDX_INT64 or PDX_FLOAT64 (mixed => float64); scalar_side (pdx_scalar_side): RHS = b has length 1, LHS = a has length 1.
 * out[i] = cond[i] ? a[i] : b[i]; null where cond is null or the chosen operand is null (the other operand's nulls do not count). */
int pdx_if_else(const pdx_column* cond, const pdx_column* a, const pdx_column* b, int scalar_side, pdx_mut_column* out, void* stream);
/* Replaces CallFunction("negate" | "abs" | "sign" | "sqrt" | "exp" | "bit_wise_not", {array}): DataFrame::unary (operator-,
 * operator~) and the UNARY_FUNCTION macros src/dataframe.cpp:251-275, 919-935, src/dataframe.h:494-502; Series::abs / exp /
 * sign / sqrt src/series.h:89-109.  a: PDX_INT64, PDX_UINT64 or PDX_FLOAT64; out null where a is null.
 *   NEGATE, ABS : the input type; integers wrap (negate(INT64_MIN) == abs(INT64_MIN) == INT64_MIN; uint64: negate wraps, abs is
 *                 the identity)
 *   SIGN        : float64 -> float64 (-1, 0, 1; NaN stays NaN); integers -> PDX_INT64 values -1 / 0 / 1 (Arrow returns int8:
 *                 this backend's integer columns are 64 bits wide)
 *   BIT_NOT     : integers only (PDX_NOT_IMPLEMENTED for float64, as Arrow has no such kernel)
 *   SQRT, EXP   : -> PDX_FLOAT64.  Integer input is cast first the way Arrow's implicit cast does: a valid value outside
 *                 +-2^53 fails the whole call with PDX_INVALID "Integer value ... not in range".  sqrt of a negative number
 *                 is NaN.  SQRT is correctly rounded (bit-identical to the reference); EXP follows the device's libm and can
 *                 differ from the reference's host libm in the last place (the reference's own result depends on its glibc
 *                 build): the only element-wise result on this path that is not bit-reproducible. */
int pdx_unary(int op, const pdx_column* a, pdx_mut_column* out, void* stream);
/* Replaces arrow::compute::Cast(column, float64()): the promotion of same-named int64 / float64 columns in pd::concat (src/concat.cpp:116-132:
 * default CastOptions = the SAFE cast: checked = 1, a valid int64 value outside +-2^53 fails the call with PDX_INVALID "Integer value ... not in
 * range: -9007199254740992 to 9007199254740992" -- the same check Arrow's DispatchBest applies when pdx_binary / pdx_compare / pdx_if_else
 * meet an int64 and a float64 operand) and the value-by-value static_cast<double> of Arrow's `mean` over int64 input (checked = 0: the
 * frame-level DataFrame::mean sums int64 chunks as doubles, src/ndframe.h:329-335).  a: PDX_INT64 (or PDX_FLOAT64: a copy); nulls carried over. */
int pdx_cast_f64(const pdx_column* a, int checked, pdx_mut_column* out, void* stream);
/* Replaces CallFunction("power", {array, Datum(double)}): Series::pow / DataFrame::pow (src/dataframe.cpp:267-270).  Integer input
 * is cast to float64 as in pdx_unary(SQRT); out: PDX_FLOAT64 = pow(a, exponent), last-place caveat as for EXP. */
int pdx_power(const pdx_column* a, double exponent, pdx_mut_column* out, void* stream);

/* ---------------------------------------------------------------- whole-array aggregates
 * Replaces CallFunction("sum"|"mean"|"min"|"max"|"count", {array}, ScalarAggregateOptions{skip_nulls=true,
 * min_count=1}) at src/ndframe.cpp:26-31 (macro), 119, 162-166, 220, and MinMax at src/resample.cpp:223.
 * fp64 sum reproduces Arrow's pairwise tree bit-for-bit (16-value leaves per valid run, binary-counter merge).
 * Synchronises `stream`. */
int pdx_aggregate(int kind, const pdx_column* a, pdx_scalar* out, void* stream);

/* ---------------------------------------------------------------- filter / take
 * Replaces CallFunction("filter", {RecordBatch, mask}, FilterOptions{EMIT_NULL}) + CallFunction("array_filter",
 * {index, mask}) at src/dataframe.cpp:461-475 and src/series.cpp:130-144.  mask: PDX_BOOL of the same length as
 * every column (else PDX_INVALID).  Two calls: count (host-visible) then fill; the facade hides the pair.
 * Output validity bitmaps: give them a capacity rounded up to a multiple of 4 bytes and a 4-byte aligned start (any device
 * allocation has both): null rows clear their bit with a 32-bit atomic on the word that holds it.  A bitmap that does not start on
 * a 4-byte boundary is still served (row-id gather path), only slower. */
int pdx_filter_count(const pdx_column* mask, int emit_null, int64_t* out_count, void* stream);
int pdx_filter(const pdx_column* cols, int ncols, const pdx_column* mask, int emit_null, pdx_mut_column* outs, void* stream);
/* Replaces CallFunction("take", {RecordBatch, indices}) + "array_take" at src/dataframe.cpp:477-492 and
 * src/series.cpp:146-159.  indices: PDX_INT64.  Out-of-range index => PDX_INDEX_ERROR "Index k out of bounds". */
int pdx_take(const pdx_column* cols, int ncols, const pdx_column* indices, pdx_mut_column* outs, void* stream);

/* Inverse of take: outs[c][indices[j]] = cols[c][j] (indices must be distinct and in [0, outs[c].length)); rows of outs
 * that no index names are left untouched.  Used to place per-owner group results by global group id. */
int pdx_scatter(const pdx_column* cols, int ncols, const pdx_column* indices, pdx_mut_column* outs, void* stream);

/* ---------------------------------------------------------------- group-by
 * pdx_groupby_create replaces GroupBy::makeGroups (src/dataframe.cpp:1571-1600): Grouper::Make + Consume
 * (dense group ids in FIRST-OCCURRENCE order, a null key is its own group) + GetUniques.  The reference's eager
 * MakeGroupings/ApplyGroupings of every column (src/dataframe.cpp:1539-1569) is deferred to pdx_groupby_agg.
 * GROUP ORDER: exactly first occurrence (group g's first row precedes group g+1's).  Arrow's Grouper::Consume (Arrow C++ 25.0.0, one
 * call over the column, as src/dataframe.cpp:1580 makes it) differs from that by a BOUNDED permutation, and only so: it walks the batch
 * in mini-batches of 128, 256, 512 and then 1024 rows (row boundaries 128, 384, 896, 1920, 2944, ...); the keys first seen in a
 * mini-batch receive ONE CONTIGUOUS BLOCK of ids -- the same block first-occurrence numbering gives them -- permuted inside the block
 * (the insertion rounds of its swiss table).  So: the reference's own tests (<= 17 rows) and any input with at most one new key per
 * mini-batch are exactly first occurrence; group g here and group g there first appear in the same mini-batch, always; per key every
 * aggregate is bit-identical; DataFrame::sort_index (src/dataframe.cpp:1062-1071; both facades) makes two results row-identical.
 * Held against live Arrow on inputs spanning thousands of mini-batches by tests/test_oracle_golden_r4.py (oracle/arrow_order.cpp) and,
 * with this library's own ids, by tests/cpp/arrow_bridge_test.cpp; measured sizes of the permutation: 51 of 97 groups at 1e5 rows / 97
 * keys, 10 145 of 43 107 at 1e5 / 5e4, 23 825 of 993 353 at 5e6 / 1e6 (tests/golden/group_order_arrow25.npz freezes two such cases).
 * key: PDX_INT64 / PDX_TIMESTAMP_NS / PDX_UINT64.  Limits: length < 2^31 rows per call (row ids are 32 bits wide inside the handle;
 * pdx_groupby_sum_mean_count_chunked serves longer inputs for the headline query by an exact merge of chunks); keys that do not span a dense integer
 * range go through a hash table of at most 2^30 slots (about 7e8 distinct keys; the LDS-resident build covers 2.7e8). */
typedef struct pdx_groupby pdx_groupby;
int pdx_groupby_create(const pdx_column* key, void* stream, pdx_groupby** out);
int pdx_groupby_destroy(pdx_groupby* gb);
int64_t pdx_groupby_num_groups(const pdx_groupby* gb);
int64_t pdx_groupby_num_rows(const pdx_groupby* gb);
/* uniqueKeys (GroupBy::unique(), src/group_by.h:52-55): G rows, first-occurrence order; validity marks the null key. */
int pdx_groupby_unique_keys(const pdx_groupby* gb, pdx_mut_column* out, void* stream);
/* Grouper::Consume output: uint32 group id per input row (out_ids: device pointer to num_rows uint32). */
int pdx_groupby_group_ids(pdx_groupby* gb, uint32_t* out_ids, void* stream);
/* out[i] = map[group_id(i)] for every input row: routes rows by a per-group attribute (global id, owner rank) in the
 * multi-GPU merge.  map: device pointer to G int64; out: device pointer to num_rows int64. */
int pdx_groupby_map_ids(pdx_groupby* gb, const int64_t* map, int64_t* out, void* stream);
/* first_row[g] = index of the first row of group g (device pointer to G int64) */
int pdx_groupby_first_rows(const pdx_groupby* gb, int64_t* out_rows, void* stream);
/* Grouper::MakeGroupings (src/dataframe.cpp:1546, 1562; what GroupBy::group / MakeSubDataFrame / apply walk, src/group_by.h:39-77):
 * out_rows (n int64, device) = the rows of group 0, then of group 1, ... each ascending; out_offsets (G + 1 int64, device): group g
 * owns out_rows[out_offsets[g] .. out_offsets[g + 1]).  Not part of pdx_groupby_create: only callers that walk groups pay for it. */
int pdx_groupby_groupings(pdx_groupby* gb, int64_t* out_rows, int64_t* out_offsets, void* stream);
/* Replaces GROUPBY_AGG(sum|min|max) and GROUPBY_NUMERIC_AGG(mean|count) (src/pd_core_macros.h:5-147, instantiated
 * src/dataframe.cpp:1512-1534): for every group, the scalar aggregate over the group's rows IN ROW ORDER.
 * `kinds`/`outs` have nk entries and are all computed from one grouped pass over `values` (sum/mean/count of the
 * headline query share one pass).  Output dtypes: SUM -> dtype of values (int64 wraps), MEAN -> FLOAT64, MIN/MAX ->
 * dtype of values, COUNT -> INT64.  outs[k].length must be >= G.  A group with no valid value yields a null
 * (validity required in that case) except COUNT. */
int pdx_groupby_agg(pdx_groupby* gb, const pdx_column* values, const int* kinds, int nk, pdx_mut_column* outs, void* stream);
/* The grouped layout of one column, built once and reused: what GroupBy's constructor does for every column of the frame
 * (processEach, src/dataframe.cpp:1539-1554: MakeGroupings + ApplyGroupings), so that gb.sum(c); gb.mean(c); gb.count(c)
 * (src/group_by.h:85-139 -- the reference has no multi-kind call) cost ONE value sort and ONE reduce instead of three of each.
 * pdx_groupby_bind registers `values` with the handle: the first pdx_groupby_agg of the column sorts it by group and keeps the
 * result (plus the per-group sum / count / min / max it computes) in the handle; every later pdx_groupby_agg whose `values` has
 * the same values pointer, offset, dtype and validity pointer is served from it.  Contract: the column's buffers stay alive and UNCHANGED until pdx_groupby_unbind /
 * pdx_groupby_destroy (Arrow buffers are immutable; the facades' GroupBy holds the frame).  Nothing is cached for columns that
 * were not bound.  values == NULL in unbind: every bound column.  The bound layouts of one handle are limited to
 * pdx_groupby_bind_limit bytes (default: a quarter of the device's memory, or PDX_BIND_MAX_BYTES): the least recently used
 * one is dropped first -- its next aggregation simply sorts again.  pdx_groupby_bound_bytes: device bytes held right now. */
int pdx_groupby_bind(pdx_groupby* gb, const pdx_column* values, void* stream);
int pdx_groupby_unbind(pdx_groupby* gb, const pdx_column* values);
int pdx_groupby_bind_limit(pdx_groupby* gb, size_t max_bytes);
int64_t pdx_groupby_bound_bytes(const pdx_groupby* gb);
/* The path the last pdx_groupby_agg on this handle took, as "key=value" words (diagnostic; tests assert it so that a moved
 * threshold fails a test instead of silently changing which kernels run):
 *   slots=dense|hash_lds|hash_part|hash_part2|hash_global|runs|bins   how keys became slots (pdx_groupby_create)
 *   sort=narrow:7+7|narrow_part:8+6|lsd:skip6|lsd|none[+finish]       the value sort (narrowing 4->2->1 byte keys, ...)
 *   layout=fused [side=<runs>] | full [skew=1]                        fused last digit vs fully sorted.  side: that many runs longer than
 *                                                                     2^19 rows (hot keys) were reduced from a side copy of their rows;
 *                                                                     skew: too many / too heavy long runs, the whole column was sorted on
 *   reducer=flr_reduce_dense|flr_reduce|flr_wave|seg_reduce|seg_reduce_nullable|none
 *   bound=0|1 [cache=fill|hit] */
int pdx_groupby_last_plan(const pdx_groupby* gb, char* buf, size_t buf_len);

/* ---------------------------------------------------------------- exact multi-GPU fp64 sum (partial-tree exchange)
 * The reference's per-group sum is Arrow's pairwise tree over the group's rows in GLOBAL row order; with row-range shards a
 * rank holds the global ranks [P, P+c) of a group.  Instead of shipping its rows to the group's owner, a rank ships
 *   - the raw values of the (at most two) 16-value leaves it shares with its neighbours ("fragments"), and
 *   - one node per maximal ALIGNED power-of-two block of the leaves that lie completely inside its range,
 * as records (key = global_gid * 64 + code; code 0 = one row of a leaf begun on a lower rank, 1..28 = a tree node of level code - 1,
 * 32 + k = the first k rows of a leaf as their sequential sum).  The owner replays the records of a
 * group in (rank, emission) order through Arrow's binary counter: bit-identical to the single-process result, with
 * O(32 + 2 log c) instead of c values per group and rank.
 *
 * pdx_groupby_group_values : stable sort of a non-null float64 column by group; the handle caches the grouped values.
 * pdx_grouped_counts       : rows per group (G int64, group-id order).
 * pdx_grouped_partial_plan : number of records given prefix[g] = rows of group g held by lower ranks (G int64, group-id order);
 *                            order (optional, G int64) = local group ids in record EMISSION order -- pass the ids sorted by global
 *                            group id and the records come out already partitioned by owner rank (no routing pass).
 * pdx_grouped_partial_fill : writes the records; gid_map[g] = global group id of local group g.
 * pdx_replay_partials      : owner side: records for global ids [gid_lo, gid_lo + n_own) in (source rank, emission) order ->
 *                            out_sum[gid - gid_lo].  Every owned id must have at least one record. */
typedef struct pdx_grouped pdx_grouped;
int pdx_groupby_group_values(pdx_groupby* gb, const pdx_column* values, void* stream, pdx_grouped** out);
int pdx_grouped_destroy(pdx_grouped* g);
int pdx_grouped_counts(pdx_grouped* g, int64_t* out_counts, void* stream);
int pdx_grouped_partial_plan(pdx_grouped* g, const int64_t* prefix, const int64_t* order, int64_t* out_total, void* stream);
int pdx_grouped_partial_fill(pdx_grouped* g, const int64_t* gid_map, int64_t* rec_key /* may be null: values only */, double* rec_val, void* stream);
/* cuts[d], d = 0..world (device): where the records of owner d's groups -- global ids [G * d / world, G * (d + 1) / world) -- begin in this
 * rank's record stream (emitted in global-id order: pass `order` to the plan).  With the rows every rank holds of every group known to all
 * ranks, the codes of the records follow from (rows on lower ranks, own rows): only the values need to travel. */
int pdx_grouped_record_cuts(pdx_grouped* g, const int64_t* gid_map, int64_t num_global_groups, int world, int64_t* cuts, void* stream);
int pdx_replay_partials(const int64_t* rec_key, const double* rec_val, int64_t m, int64_t gid_lo, int64_t n_own, double* out_sum, void* stream);

/* ---------------------------------------------------------------- resample
 * Replaces pd::resample<> / makeGroupInfo / generate_bins_dt64 / GroupInfo::downsample + the Resampler's GroupBy on
 * the per-row labels (src/resample.h:19-43,91-122; src/resample.cpp:11-83,85-178,202-295; src/core.cpp:308-331;
 * src/group_by.h:255-299).  ts: sorted PDX_TIMESTAMP_NS without nulls.  The handle behaves like a pdx_groupby whose
 * unique keys are the labels of the NON-EMPTY bins (empty bins vanish, as in the reference).
 * Errors (PDX_INVALID): "Values falls before first bin", "Values falls after last bin",
 * "upSampling is not implemented.", unsorted input.  Row-range shards of one axis: see PDX_ORIGIN_SHARD.
 * Lifetime: the handle keeps the POINTER to ts' values (pdx_resample_row_labels reads them again): the caller's timestamp
 * buffer must stay alive and unchanged until pdx_groupby_destroy. */
int pdx_resample_create(const pdx_column* ts, int64_t freq_ns, int closed_right, int label_right, int origin_type,
                        int64_t origin_custom_ns, int64_t offset_ns, void* stream, pdx_groupby** out);
/* The grid alone (host arithmetic, no GPU): first bin edge and number of bins of an axis with extremes tmin / tmax -- adjustDatesAnchored
 * + date_range (src/resample.cpp:85-178, src/core.cpp:308-331), with the reference's errors.  pdx_resample_create uses it; so does
 * the sharded resample, where every rank must bin on the WHOLE axis' grid. */
int pdx_resample_grid(int64_t tmin, int64_t tmax, int64_t freq_ns, int closed_right, int origin_type, int64_t origin_custom_ns, int64_t offset_ns,
                      int64_t* first_edge, int64_t* num_bins);
/* per-row labels (GroupInfo::downsample): device pointer to num_rows int64 */
int pdx_resample_row_labels(pdx_groupby* gb, int64_t* out_labels, void* stream);

/* DataFrame::downsample's grouping in one call (reference src/dataframe.cpp:1265-1290: arrow::compute::CeilTemporal /
 * FloorTemporal(m_index, RoundTemporalOptions(multiple, unit, week_starts_monday, false, calendar_based_origin)), for the
 * M / W / Y / Q and *E rules Subtract(one day), then Resampler(GroupBy) keyed on the rounded index).  The handle behaves like
 * pdx_groupby_create(pdx_round_temporal(ts) + label_shift_ns): unique keys = the distinct rounded labels (+ shift) in
 * first-occurrence order, null timestamps form one null group.  ceil_mode / unit / multiple / week_starts_monday /
 * calendar_based_origin as in pdx_round_temporal; label_shift_ns is added to every label (e.g. -86400e9).
 * When ts has no nulls and its rounded labels never descend (a sorted axis) the groups are found as runs of equal labels in
 * ONE pass over ts -- the rounded column is never materialised or hashed; any other input takes pdx_round_temporal +
 * pdx_groupby_create internally.  Same results either way.  The handle keeps no pointer into ts. */
int pdx_downsample_create(const pdx_column* ts, int64_t multiple, int unit, int ceil_mode, int week_starts_monday,
                          int calendar_based_origin, int64_t label_shift_ns, void* stream, pdx_groupby** out);

/* ---------------------------------------------------------------- temporal rounding: DataFrame::downsample
 * Replaces arrow::compute::FloorTemporal / CeilTemporal(m_index, RoundTemporalOptions(freq_value, getCalendarUnit(freq_unit[0]),
 * weekStartsMonday, ceil_is_strictly_greater = false, calendar_based_origin = startEpoch)) at src/dataframe.cpp:1271-1276
 * (unit letters: src/core.cpp:135-172).  ts: PDX_TIMESTAMP_NS (no time zone); out: PDX_TIMESTAMP_NS of the same length, null where
 * ts is null.  ceil_mode != 0: CeilTemporal (closed_label_right), else FloorTemporal.  Semantics follow Arrow C++ 25.0.0:
 * floors go toward -inf; calendar_based_origin counts the multiples from the floor to the next larger unit (day: the 1st of the
 * month, week: the first week of the year) instead of from the epoch; month / quarter ceil is always floor + multiple.
 * The reference then subtracts one day for M / W / Q rules (src/dataframe.cpp:1277-1285: pdx_binary(PDX_SUB) on the int64 view)
 * and keys a Resampler's GroupBy on the result (pdx_groupby_create; the labels need not be sorted). */
typedef enum pdx_calendar_unit {
  PDX_UNIT_NANOSECOND = 0, PDX_UNIT_MICROSECOND = 1, PDX_UNIT_MILLISECOND = 2, PDX_UNIT_SECOND = 3, PDX_UNIT_MINUTE = 4,
  PDX_UNIT_HOUR = 5, PDX_UNIT_DAY = 6, PDX_UNIT_WEEK = 7, PDX_UNIT_MONTH = 8, PDX_UNIT_QUARTER = 9
} pdx_calendar_unit;
int pdx_round_temporal(int ceil_mode, const pdx_column* ts, int64_t multiple, int unit, int week_starts_monday, int calendar_based_origin,
                       pdx_mut_column* out, void* stream);

/* ---------------------------------------------------------------- index alignment (the callers' slow path)
 * Binary operators on Series with UNEQUAL indexes go through Series::broadcast (src/series.cpp:212-227):
 * Concatenate(index_a, index_b) -> Unique -> array_sort_indices(ascending) -> Take, then Series::reindex of both operands
 * (src/series.cpp:1255-1309: std::unordered_map label -> LAST position, absent labels -> null).
 * pdx_index_union: the distinct labels of a and b (same dtype: int64 / uint64 / timestamp[ns], no nulls), sorted ascending
 *   (sort != 0: Series::broadcast) or in first-occurrence order (sort == 0: Series::union_, src/series.cpp:782-798, used by the
 *   column-wise concat, src/concat.cpp:78-88, 192-244).  out: same dtype, capacity a.length + b.length; out->length is set.
 * pdx_index_intersection: Series::intersection (src/series.cpp:763-780): the labels of a that occur in b, one per distinct
 *   label, ordered by their (last) position in a.  out: capacity a.length.
 * pdx_reindex_indices: for every label of new_index its LAST position in old_index as int64 take indices with a validity
 *   bitmap (absent label -> null index); feed it to pdx_take, whose null indices produce null rows (== AppendNull). */
int pdx_index_union(const pdx_column* a, const pdx_column* b, int sort, pdx_mut_column* out, void* stream);
int pdx_index_intersection(const pdx_column* a, const pdx_column* b, pdx_mut_column* out, void* stream);
int pdx_reindex_indices(const pdx_column* old_index, const pdx_column* new_index, pdx_mut_column* out_idx, void* stream);

/* ---------------------------------------------------------------- sort (sort / argsort / n_largest / n_smallest)
 * Series::argsort (src/series.cpp:864-868) and Series::sort (978-992: Take of values and index by the same indices; n_largest /
 * n_smallest, 1211-1229, are sort + Slice) call CallFunction("array_sort_indices", ArraySortOptions{order}).  Semantics pinned
 * against Arrow 25.0.0: STABLE in both orders (equal values keep their row order; -0.0 == 0.0), NaN behind every number and
 * nulls behind the NaNs in BOTH orders.  col: int64 / uint64 / float64 / timestamp[ns], <= 2^31-1 rows; out_indices: PDX_UINT64,
 * capacity col.length (the take indices; feed them to pdx_take). */
int pdx_argsort(const pdx_column* col, int ascending, pdx_mut_column* out_indices, void* stream);

/* ---------------------------------------------------------------- concat (rows)
 * Replaces arrow::ConcatenateTables + CombineChunksToBatch at src/concat.cpp:152-154 for same-dtype parts
 * (the all-gatherv merge of sharded results).  out->length must be >= sum of part lengths. */
int pdx_concat(const pdx_column* parts, int nparts, pdx_mut_column* out, void* stream);

/* ---------------------------------------------------------------- Arrow IPC streams <-> device columns
 * Replaces, for the column types of this path, DataFrame::readBinary (src/dataframe.cpp:754-791: ipc::RecordBatchStreamReader::Open
 * + ToRecordBatches, exactly ONE record batch) and DataFrame::toBinary (src/dataframe.cpp:726-752: ipc::MakeStreamWriter +
 * WriteRecordBatch(batch, custom metadata)).  A record batch body is the Arrow layout the kernels read, so reading is a parse of
 * the two flatbuffer messages on the host plus ONE host->device copy of the body; the columns are pointers into it.
 *   pdx_ipc_open     parse a stream held in HOST memory (no GPU needed); the blob must stay alive until pdx_ipc_load returns.
 *                    Column types: bool, (u)int8..64, float32/64, timestamp (any unit, time zone ignored), date64; narrow
 *                    numerics are widened to int64 / uint64 / float64 and timestamps scaled to nanoseconds on the device.
 *                    Anything else (strings, nested, dictionaries, compressed bodies): PDX_NOT_IMPLEMENTED naming the field.
 *   pdx_ipc_load     upload (one copy + the widening kernels); synchronises `stream`.
 *   pdx_ipc_column   the i-th column (device pointers owned by the frame until pdx_ipc_destroy; before pdx_ipc_load only dtype,
 *                    length and null_count are filled in).
 *   pdx_ipc_write    serialise equally long columns as one schema + one record batch (+ custom metadata key/value pairs:
 *                    metadata_kv holds 2 x nmeta strings) + end-of-stream.  columns_on_host != 0: the pdx_column pointers are
 *                    host memory.  *out_blob is malloc'ed by the library: release it with pdx_ipc_free_blob.
 * The index column convention (readBinary's `index` argument: pull the named column out, int64 -> timestamp[ns]) lives in the
 * facades. */
typedef struct pdx_ipc_frame pdx_ipc_frame;
int pdx_ipc_open(const void* blob, size_t size, pdx_ipc_frame** out);
int pdx_ipc_destroy(pdx_ipc_frame* frame);
int pdx_ipc_num_columns(const pdx_ipc_frame* frame);
int64_t pdx_ipc_num_rows(const pdx_ipc_frame* frame);
const char* pdx_ipc_column_name(const pdx_ipc_frame* frame, int i);
int pdx_ipc_num_metadata(const pdx_ipc_frame* frame);
const char* pdx_ipc_metadata_key(const pdx_ipc_frame* frame, int i);
const char* pdx_ipc_metadata_value(const pdx_ipc_frame* frame, int i);
int pdx_ipc_load(pdx_ipc_frame* frame, void* stream);
int pdx_ipc_column(const pdx_ipc_frame* frame, int i, pdx_column* out);
int pdx_ipc_write(const pdx_column* cols, const char* const* names, int ncols, const char* const* metadata_kv, int nmeta, int columns_on_host,
                  void* stream, void** out_blob, size_t* out_size);
int pdx_ipc_free_blob(void* blob);

/* ---------------------------------------------------------------- row-range shards across the GPUs of one node
 * One process per GPU; every rank holds a row range of the frame, ranks in row order.  The reference has no distributed code: the
 * contract is the SINGLE-process result of df.group_by(key).sum / mean / count (src/group_by.h:85-139, src/pd_core_macros.h:5-147)
 * and of pd::concat (src/concat.cpp:116-190), bit for bit.  librccl is opened at run time and called directly (ncclAllGather,
 * grouped ncclSend / ncclRecv over xGMI): no python, no torch in the loop -- a C++ host shards through this same header.
 *   pdx_dist_unique_id   rank 0: 128 bytes (an ncclUniqueId) to hand to the other ranks by whatever channel the host has.
 *   pdx_dist_init        ncclCommInitRank on the calling thread's current device (after pdx_init(device)).
 *   pdx_dist_init_custom the same orchestration over the caller's own transport: two primitives on DEVICE buffers, ordered on the
 *                        given stream -- all_gather (every rank contributes `bytes` bytes, received in rank order) and
 *                        all_to_all_v (byte ranges send_off/send_bytes[peer] of `send` go to peer, recv_off/recv_bytes[peer] of `recv`
 *                        come from peer).  Return 0 for success.  (tests: three processes sharing one GPU, where RCCL refuses
 *                        duplicate devices.)
 *   pdx_dist_groupby_sum_mean_count   keys / values: this rank's shard (values: float64 without nulls); row_offset: index of the
 *                        shard's first row in the whole frame.  Partial-tree exchange: the global first-occurrence dictionary from
 *                        an all-gather(v) of the local uniques, the per-group row counts all-gathered, then per group and rank only the
 *                        boundary-leaf fragments + aligned subtree nodes travel (ONE all-to-all(v)) and the owners replay them through
 *                        Arrow's binary counter.  Every rank ends with the FULL result (G groups in first-occurrence order).
 *   pdx_dist_groupby_fetch   copies of the result columns (any pointer may be NULL; capacity >= num_groups rows).
 *   pdx_dist_concat      all-gather(v) of one column's shards in rank order (values + validity).
 * PDX_DIST_FORCE_COLLECTIVES=1 keeps every collective on the wire at world size 1 (tests on a one-GPU box). */
typedef struct pdx_dist_transport {
  void* ctx;
  int (*all_gather)(void* ctx, const void* send, void* recv, size_t bytes, void* stream);
  int (*all_to_all_v)(void* ctx, const void* send, const size_t* send_off, const size_t* send_bytes, void* recv, const size_t* recv_off,
                      const size_t* recv_bytes, void* stream);
} pdx_dist_transport;
typedef struct pdx_dist pdx_dist;
typedef struct pdx_dist_groupby pdx_dist_groupby;
int pdx_dist_unique_id(void* out_id128);
int pdx_dist_init(const void* id128, int world, int rank, pdx_dist** out);
int pdx_dist_init_custom(const pdx_dist_transport* transport, int world, int rank, pdx_dist** out);
int pdx_dist_destroy(pdx_dist* d);
int pdx_dist_world(const pdx_dist* d);
int pdx_dist_rank(const pdx_dist* d);
int pdx_dist_groupby_sum_mean_count(pdx_dist* d, const pdx_column* keys, const pdx_column* values, int64_t row_offset, void* stream,
                                    pdx_dist_groupby** out);
int64_t pdx_dist_groupby_num_groups(const pdx_dist_groupby* g);
int64_t pdx_dist_groupby_num_records(const pdx_dist_groupby* g);
int pdx_dist_groupby_fetch(const pdx_dist_groupby* g, pdx_mut_column* keys, int64_t* first_rows, double* sums, double* means, int64_t* counts,
                           void* stream);
int pdx_dist_groupby_destroy(pdx_dist_groupby* g);
int pdx_dist_concat(pdx_dist* d, const pdx_column* part, pdx_mut_column* out, void* stream);
/* The order-free kinds over row-range shards -- df.group_by(key).{min, max, count}(col) and the sum of an int64 column
 * (src/group_by.h:85-139, GROUPBY_AGG / GROUPBY_NUMERIC_AGG src/pd_core_macros.h:5-147): every rank reduces its shard, the results go into
 * dense per-group partial arrays indexed by GLOBAL group id, ONE all-gather and a fold in rank order finish (SURVEY 8e 3a's
 * reduce-by-key; int64 sums wrap, min keeps the first of ties, max the first -- the last when the group holds a null on any rank).
 * Values may carry nulls.  kinds: PDX_AGG_MIN / PDX_AGG_MAX / PDX_AGG_COUNT, PDX_AGG_SUM for int64 values; float64 sums / means are order
 * dependent and stay with pdx_dist_groupby_sum_mean_count.  One corner is not Arrow's: a maximum that is a tie of +0.0 and -0.0 ACROSS
 * ranks in a group that has a null on one rank and its last zeros on a rank whose share has none takes that share's first zero.
 * Collective like the other pdx_dist_* calls: a rank whose shard fails its local checks makes EVERY rank return an error before any
 * data exchange.  Fetch: keys / first_rows may be NULL; outs[k] = the k-th requested kind, dtypes as pdx_groupby_agg. */
typedef struct pdx_dist_agg pdx_dist_agg;
int pdx_dist_groupby_order_free(pdx_dist* d, const pdx_column* keys, const pdx_column* values, const int* kinds, int nk, int64_t row_offset, void* stream,
                                pdx_dist_agg** out);
int64_t pdx_dist_agg_num_groups(const pdx_dist_agg* g);
int pdx_dist_agg_fetch(const pdx_dist_agg* g, pdx_mut_column* keys, int64_t* first_rows, pdx_mut_column* outs, void* stream);
int pdx_dist_agg_destroy(pdx_dist_agg* g);
/* pd::resample(df, rule).{kinds}(col) over a sorted axis sharded by row ranges (src/resample.h:91-122, src/group_by.h:255-299): the
 * leading rows of a shard whose bin opened on an earlier rank move there (one all-to-all(v)), every rank bins on the whole axis'
 * grid (pdx_resample_grid of the all-gathered extremes), results are all-gathered in rank order == label order.  kinds:
 * PDX_AGG_SUM .. PDX_AGG_LAST.  Every rank ends with the full result; the reference's whole-axis errors are raised on every rank. */
typedef struct pdx_dist_resampled pdx_dist_resampled;
int pdx_dist_resample(pdx_dist* d, const pdx_column* ts, const pdx_column* values, const int* kinds, int nk, int64_t freq_ns, int closed_right,
                      int label_right, int origin_type, int64_t origin_custom_ns, int64_t offset_ns, void* stream, pdx_dist_resampled** out);
int64_t pdx_dist_resampled_num_bins(const pdx_dist_resampled* g);
int pdx_dist_resampled_fetch(const pdx_dist_resampled* g, pdx_mut_column* labels, pdx_mut_column* outs, void* stream);
int pdx_dist_resampled_destroy(pdx_dist_resampled* g);
/* The headline query on ONE GPU for inputs beyond the per-call limit of pdx_groupby_create (2^31 - 1 rows: the reference's own
 * Grouper::MakeGroupings breaks there): the rows are cut into chunks of chunk_rows (0 = the largest the limit allows; tests
 * pass small values), every chunk plays one rank of the exchange above on a host thread of its own, and the partial-tree records merge
 * the chunks' sums bit-identically to ONE pairwise tree over the whole column.  Result handle as pdx_dist_groupby_sum_mean_count. */
int pdx_groupby_sum_mean_count_chunked(const pdx_column* keys, const pdx_column* values, int64_t chunk_rows, void* stream, pdx_dist_groupby** out);
/* The same for the order-free kinds (PDX_AGG_MIN / MAX / COUNT, PDX_AGG_SUM of int64 values; values may carry nulls): chunks play the ranks of
 * pdx_dist_groupby_order_free, dense per-group partials folded in chunk order.  Result handle as pdx_dist_groupby_order_free (pdx_dist_agg_fetch). */
int pdx_groupby_order_free_chunked(const pdx_column* keys, const pdx_column* values, const int* kinds, int nk, int64_t chunk_rows, void* stream, pdx_dist_agg** out);

/* ---------------------------------------------------------------- Parquet files -> device columns
 * Replaces, for the column types of this path, DataFrame::readParquet (src/dataframe.cpp:646-683: parquet::arrow::OpenFile ->
 * FileReader::ReadTable -> TableBatchReader::ToRecordBatches, exactly ONE record batch).  The host parses the metadata only (the
 * Thrift-compact footer in pdx_parquet_open, the page headers in pdx_parquet_load); the column chunks go to the device in ONE copy
 * and every page payload is decoded there: Snappy blocks, RLE / bit-packed definition levels -> validity bitmap, PLAIN and
 * dictionary-encoded (PLAIN_DICTIONARY / RLE_DICTIONARY) values -> 8-byte values, data pages v1 and v2.
 *   pdx_parquet_open    parse a file held in HOST memory (no GPU needed); the bytes must stay alive until pdx_parquet_load returns.
 *                       Columns: flat BOOLEAN / INT32 / INT64 / FLOAT / DOUBLE leaves, REQUIRED or OPTIONAL; INT annotations widen to
 *                       int64 / uint64, FLOAT to float64, TIMESTAMP(ms | us | ns) becomes timestamp[ns].  One row group, as the
 *                       reference: more give PDX_INVALID "DataFrame Only supports Parquet Table with single record batch", an empty
 *                       table "Cannot Initialize DataFrame with empty parquet table".  PDX_NOT_IMPLEMENTED, naming the column and the
 *                       feature, for strings / BYTE_ARRAY, nested or repeated columns, INT96, DATE / TIME / DECIMAL annotations,
 *                       codecs other than UNCOMPRESSED / SNAPPY, DELTA_* / BYTE_STREAM_SPLIT encodings, encryption.
 *   pdx_parquet_load    upload + decode; synchronises `stream`.  Malformed pages (bad Snappy blocks, short value sections,
 *                       dictionary indices out of range) are detected on the device and fail the call with PDX_INVALID.
 *   pdx_parquet_column  the i-th column (device pointers owned by the file object until pdx_parquet_destroy; before the load only
 *                       dtype, length and the chunk statistics' null_count, -1 when the writer left it out).
 * The file's key_value_metadata (pandas / ARROW:schema entries) is available as strings. */
typedef struct pdx_parquet_file pdx_parquet_file;
int pdx_parquet_open(const void* blob, size_t size, pdx_parquet_file** out);
int pdx_parquet_destroy(pdx_parquet_file* file);
int pdx_parquet_num_columns(const pdx_parquet_file* file);
int64_t pdx_parquet_num_rows(const pdx_parquet_file* file);
const char* pdx_parquet_column_name(const pdx_parquet_file* file, int i);
int pdx_parquet_num_metadata(const pdx_parquet_file* file);
const char* pdx_parquet_metadata_key(const pdx_parquet_file* file, int i);
const char* pdx_parquet_metadata_value(const pdx_parquet_file* file, int i);
int pdx_parquet_load(pdx_parquet_file* file, void* stream);
int pdx_parquet_column(const pdx_parquet_file* file, int i, pdx_column* out);
/* DataFrame::toParquet (src/dataframe.cpp:685-724: one record batch through parquet::arrow::WriteTable with default properties): the
 * equally long columns as ONE row group of uncompressed v1 data pages with PLAIN values; an OPTIONAL column's definition levels are one
 * bit-packed run per page whose payload IS the Arrow validity bitmap of the page's rows.  Columns without a validity bitmap are
 * REQUIRED; uint64 / timestamp[ns] carry their logical types.  Assembled on the host (columns_on_host != 0: the pdx_column pointers
 * are host memory); *out_blob is malloc'ed by the library: release it with pdx_parquet_free_blob.  Readable by Arrow / pyarrow and by
 * pdx_parquet_open. */
int pdx_parquet_write(const pdx_column* cols, const char* const* names, int ncols, int columns_on_host, void* stream, void** out_blob, size_t* out_size);
int pdx_parquet_free_blob(void* blob);

#ifdef __cplusplus
}
#endif
#endif /* PDX_ABI_H */
