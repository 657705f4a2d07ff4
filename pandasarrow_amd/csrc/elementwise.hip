// elementwise.hip -- element-wise arithmetic, comparison and logical kernels for gfx950.
//
// Replaces the Arrow kernels reached from Series::operator{+,-,*,/,<,<=,...,&&,||,!}
// (reference src/series.cpp:19-33,229-261,319; src/scalar.cpp:24-36) and the DataFrame
// forms (src/dataframe.cpp:233-275).  All kernels are HBM-bound streams:
//   binary  : 2 x 8 B read + 8 B write per row  (24 B/row, + 3/8 B with validity)
//   compare : 2 x 8 B read + 1/8 B write per row
// Layout: grid-stride over coalesced 8-byte lanes (a wave moves 512 B per instruction, 4 independent
// instructions in flight per thread); bit-packed outputs are assembled with wave-wide ballots so every
// wave stores whole 64-bit words, 64 words (512 B) at a time.
#include "pdx_common.hpp"

namespace pdx {

// ---------------------------------------------------------------- validity: out = va & vb (bit offsets honoured)
// one thread per output 64-bit word
__global__ void k_validity_and(const uint8_t* __restrict__ va, int64_t aoff, int64_t alimit, const uint8_t* __restrict__ vb,
                               int64_t boff, int64_t blimit, int b_broadcast, int64_t n, uint8_t* __restrict__ out) {
  int64_t nwords = (n + 63) >> 6;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool b_all = true;
  if (vb && b_broadcast) b_all = bit_get(vb, boff);
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += stride) {
    int64_t base = w << 6;
    uint64_t x = va ? load_bits64(va, aoff + base, alimit) : ~0ull;
    uint64_t y = ~0ull;
    if (vb) y = b_broadcast ? (b_all ? ~0ull : 0ull) : load_bits64(vb, boff + base, blimit);
    uint64_t r = x & y;
    int64_t remain = n - base;
    if (remain >= 64) {
      reinterpret_cast<uint64_t*>(out)[w] = r;
    } else {
      r &= (1ull << remain) - 1ull;
      int nbytes = (int)((remain + 7) >> 3);
      for (int k = 0; k < nbytes; ++k) out[(w << 3) + k] = (uint8_t)(r >> (8 * k));
    }
  }
}

int launch_validity_and(const pdx_column* a, const pdx_column* b, int b_is_scalar, int64_t n, uint8_t* out, hipStream_t st) {
  const uint8_t* va = validity_or_null(a);
  const uint8_t* vb = b ? validity_or_null(b) : nullptr;
  if (n == 0) return PDX_OK;
  int64_t nwords = (n + 63) >> 6;
  hipLaunchKernelGGL(k_validity_and, dim3(grid_for(nwords, 256)), dim3(256), 0, st, va, a->offset, a->offset + a->length, vb,
                     b ? b->offset : 0, b ? b->offset + b->length : 0, b_is_scalar, n, out);
  PDX_LAUNCH_CHECK();
  return PDX_OK;
}

// ---------------------------------------------------------------- binary arithmetic
template <typename T>
struct Conv;
template <>
struct Conv<double> {
  template <typename S>
  __device__ static double from(S x) { return (double)x; }
};
template <>
struct Conv<int64_t> {
  template <typename S>
  __device__ static int64_t from(S x) { return (int64_t)x; }
};

// NaN results carry the bits the reference's x86 host would produce, not CDNA's: SSE hands back the FIRST NaN operand (quieted;
// the second one if only that is NaN -- not negated by a subtraction), and an invalid operation (inf - inf, 0 * inf, 0 / 0,
// inf / inf) yields the negative "real indefinite" 0xFFF8000000000000, where v_add/v_mul/v_div_f64 give +qNaN or flip the sign
// of a negated source.  Three selects on values already in registers: free in an HBM-bound kernel.
__device__ __forceinline__ double x86_nan(double r, double x, double y) {
  if (r == r) return r;
  const unsigned long long quiet = 0x0008000000000000ull;
  if (x != x) return __longlong_as_double((long long)((unsigned long long)__double_as_longlong(x) | quiet));
  if (y != y) return __longlong_as_double((long long)((unsigned long long)__double_as_longlong(y) | quiet));
  return __longlong_as_double((long long)0xFFF8000000000000ull);
}

template <typename TO, int OP>
__device__ __forceinline__ TO apply_op(TO x, TO y, bool valid, unsigned long long* err) {
  if constexpr (sizeof(TO) == 8 && OP == PDX_ADD) {
    if constexpr (__is_same(TO, double)) return x86_nan(x + y, x, y);
    else return (int64_t)((uint64_t)x + (uint64_t)y);
  } else if constexpr (OP == PDX_SUB) {
    if constexpr (__is_same(TO, double)) return x86_nan(x - y, x, y);
    else return (int64_t)((uint64_t)x - (uint64_t)y);
  } else if constexpr (OP == PDX_MUL) {
    if constexpr (__is_same(TO, double)) return x86_nan(x * y, x, y);
    else return (int64_t)((uint64_t)x * (uint64_t)y);
  } else if constexpr (OP >= PDX_BIT_OR) {  // integers only (the host never instantiates these for double)
    const uint64_t ux = (uint64_t)(int64_t)x, uy = (uint64_t)(int64_t)y;
    if constexpr (OP == PDX_BIT_OR) return (TO)(int64_t)(ux | uy);
    else if constexpr (OP == PDX_BIT_AND) return (TO)(int64_t)(ux & uy);
    else if constexpr (OP == PDX_BIT_XOR) return (TO)(int64_t)(ux ^ uy);
    else {
      // Arrow's unchecked shifts: an amount outside [0, digits) -- digits = 63 for int64 -- returns the left operand
      const int64_t sh = (int64_t)y;
      if (sh < 0 || sh >= 63) return x;
      if constexpr (OP == PDX_SHIFT_LEFT) return (TO)(int64_t)(ux << sh);
      else return (TO)((int64_t)x >> sh);
    }
  } else {
    if constexpr (__is_same(TO, double)) {
      return x86_nan(x / y, x, y);
    } else {
      // Arrow "divide" (unchecked): truncation toward zero; INT64_MIN / -1 -> 0; zero divisor at a valid slot is an error
      if (!valid) return 0;
      if (y == 0) {
        *err = 1ull;
        return 0;
      }
      if (x == INT64_MIN && y == -1) return 0;
      return x / y;
    }
  }
}

// SCALAR: 0 = two arrays, 1 = b is one broadcast value (Series op Scalar, src/series.cpp:25-28), 2 = a is one broadcast value
// (Scalar op Series, src/scalar.cpp:24-36: CallFunction(name, {scalar, array}) -- the scalar stays the LEFT operand, which
// matters for subtract / divide and for which NaN payload survives)
template <typename TA, typename TB, typename TO, int OP, int SCALAR>
__global__ void __launch_bounds__(256) k_binary(const TA* __restrict__ a, const TB* __restrict__ b, TO* __restrict__ out, int64_t n,
                                                const uint8_t* __restrict__ va, int64_t aoff, const uint8_t* __restrict__ vb,
                                                int64_t boff, unsigned long long* __restrict__ err) {
  constexpr bool kNeedValid = (OP == PDX_DIV) && !__is_same(TO, double);
  constexpr bool SA = SCALAR == 2, SB = SCALAR == 1;
  // Arrow's scalar-array loops for the commutative ops keep the ARRAY element as the first machine operand (measured against
  // Arrow 25.0.0: NaN(scalar) + NaN(array) returns the array's payload in both orders), so add / multiply swap operands
  constexpr bool kSwap = SA && (OP == PDX_ADD || OP == PDX_MUL);
  int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  TO xs = 0, ys = 0;
  bool xs_valid = true, ys_valid = true;
  if constexpr (SB) {
    ys = Conv<TO>::from(b[0]);
    if (kNeedValid && vb) ys_valid = bit_get(vb, boff);
  }
  if constexpr (SA) {
    xs = Conv<TO>::from(a[0]);
    if (kNeedValid && va) xs_valid = bit_get(va, aoff);
  }
  unsigned long long local_err = 0;
  // 4 independent 8-byte streams per thread
  int64_t i = tid;
  for (; i + 3 * stride < n; i += 4 * stride) {
    TO x[4], y[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      x[k] = SA ? xs : Conv<TO>::from(a[i + k * stride]);
      y[k] = SB ? ys : Conv<TO>::from(b[i + k * stride]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bool valid = true;
      if constexpr (kNeedValid) {
        int64_t j = i + k * stride;
        valid = (SA ? xs_valid : (!va || bit_get(va, aoff + j))) && (SB ? ys_valid : (!vb || bit_get(vb, boff + j)));
      }
      out[i + k * stride] = kSwap ? apply_op<TO, OP>(y[k], x[k], valid, &local_err) : apply_op<TO, OP>(x[k], y[k], valid, &local_err);
    }
  }
  for (; i < n; i += stride) {
    TO x = SA ? xs : Conv<TO>::from(a[i]);
    TO y = SB ? ys : Conv<TO>::from(b[i]);
    bool valid = true;
    if constexpr (kNeedValid) valid = (SA ? xs_valid : (!va || bit_get(va, aoff + i))) && (SB ? ys_valid : (!vb || bit_get(vb, boff + i)));
    out[i] = kSwap ? apply_op<TO, OP>(y, x, valid, &local_err) : apply_op<TO, OP>(x, y, valid, &local_err);
  }
  if constexpr (kNeedValid) {
    if (local_err) atomicMax(err, local_err);
  }
}

template <typename TA, typename TB, typename TO, int OP>
static void launch_binary_sb(const pdx_column* a, const pdx_column* b, int scalar, TO* out, unsigned long long* err, hipStream_t st) {
  int64_t n = scalar == 2 ? b->length : a->length;
  const TA* pa = static_cast<const TA*>(a->values) + a->offset;
  const TB* pb = static_cast<const TB*>(b->values) + b->offset;
  dim3 grid(grid_for(n, 256, 4)), block(256);
  if (scalar == 1)
    hipLaunchKernelGGL((k_binary<TA, TB, TO, OP, 1>), grid, block, 0, st, pa, pb, out, n, validity_or_null(a), a->offset,
                       validity_or_null(b), b->offset, err);
  else if (scalar == 2)
    hipLaunchKernelGGL((k_binary<TA, TB, TO, OP, 2>), grid, block, 0, st, pa, pb, out, n, validity_or_null(a), a->offset,
                       validity_or_null(b), b->offset, err);
  else
    hipLaunchKernelGGL((k_binary<TA, TB, TO, OP, 0>), grid, block, 0, st, pa, pb, out, n, validity_or_null(a), a->offset,
                       validity_or_null(b), b->offset, err);
}
template <typename TA, typename TB, typename TO>
static void launch_binary_op(int op, const pdx_column* a, const pdx_column* b, int scalar, TO* out, unsigned long long* err,
                             hipStream_t st) {
  switch (op) {
    case PDX_ADD: launch_binary_sb<TA, TB, TO, PDX_ADD>(a, b, scalar, out, err, st); break;
    case PDX_SUB: launch_binary_sb<TA, TB, TO, PDX_SUB>(a, b, scalar, out, err, st); break;
    case PDX_MUL: launch_binary_sb<TA, TB, TO, PDX_MUL>(a, b, scalar, out, err, st); break;
    case PDX_DIV: launch_binary_sb<TA, TB, TO, PDX_DIV>(a, b, scalar, out, err, st); break;
    default:
      if constexpr (__is_same(TA, int64_t) && __is_same(TB, int64_t) && __is_same(TO, int64_t)) {
        switch (op) {
          case PDX_BIT_OR: launch_binary_sb<TA, TB, TO, PDX_BIT_OR>(a, b, scalar, out, err, st); break;
          case PDX_BIT_AND: launch_binary_sb<TA, TB, TO, PDX_BIT_AND>(a, b, scalar, out, err, st); break;
          case PDX_BIT_XOR: launch_binary_sb<TA, TB, TO, PDX_BIT_XOR>(a, b, scalar, out, err, st); break;
          case PDX_SHIFT_LEFT: launch_binary_sb<TA, TB, TO, PDX_SHIFT_LEFT>(a, b, scalar, out, err, st); break;
          default: launch_binary_sb<TA, TB, TO, PDX_SHIFT_RIGHT>(a, b, scalar, out, err, st); break;
        }
      }
      break;
  }
}

// ---------------------------------------------------------------- comparisons -> bit-packed bools
template <typename T, int OP>
__device__ __forceinline__ bool cmp_op(T x, T y) {
  if constexpr (OP == PDX_EQ) return x == y;
  else if constexpr (OP == PDX_NE) return x != y;
  else if constexpr (OP == PDX_LT) return x < y;
  else if constexpr (OP == PDX_LE) return x <= y;
  else if constexpr (OP == PDX_GT) return x > y;
  else return x >= y;
}

// each wave owns tiles of 4096 rows: 64 ballots -> lane k keeps word k -> one 512-byte store
template <typename TA, typename TB, typename TC, int OP, bool SCALAR_B>
__global__ void __launch_bounds__(256) k_compare(const TA* __restrict__ a, const TB* __restrict__ b, uint8_t* __restrict__ out,
                                                 int64_t n) {
  const int lane = threadIdx.x & 63;
  int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int64_t ntiles = (n + 4095) >> 12;
  TC ys = 0;
  if constexpr (SCALAR_B) ys = Conv<TC>::from(b[0]);
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    int64_t base = t << 12;
    uint64_t myword = 0;
    if (base + 4096 <= n) {
#pragma unroll 8
      for (int k = 0; k < 64; ++k) {
        int64_t i = base + (k << 6) + lane;
        TC x = Conv<TC>::from(a[i]);
        TC y = SCALAR_B ? ys : Conv<TC>::from(b[i]);
        uint64_t bal = __ballot(cmp_op<TC, OP>(x, y));
        if (lane == k) myword = bal;
      }
      reinterpret_cast<uint64_t*>(out)[(base >> 6) + lane] = myword;
    } else {
      for (int k = 0; k < 64; ++k) {
        int64_t i = base + (k << 6) + lane;
        bool p = false;
        if (i < n) {
          TC x = Conv<TC>::from(a[i]);
          TC y = SCALAR_B ? ys : Conv<TC>::from(b[i]);
          p = cmp_op<TC, OP>(x, y);
        }
        uint64_t bal = __ballot(p);
        if (lane == k) myword = bal;
      }
      // ragged tail: byte-granular store of the words that hold rows < n
      int64_t wbase = (base >> 6) + lane;
      int64_t first_row = wbase << 6;
      if (first_row < n) {
        int64_t remain = n - first_row;
        int nbytes = remain >= 64 ? 8 : (int)((remain + 7) >> 3);
        for (int q = 0; q < nbytes; ++q) out[(wbase << 3) + q] = (uint8_t)(myword >> (8 * q));
      }
    }
  }
}

template <typename TA, typename TB, typename TC, int OP>
static void launch_compare_sb(const pdx_column* a, const pdx_column* b, int scalar, uint8_t* out, hipStream_t st) {
  int64_t n = a->length;
  const TA* pa = static_cast<const TA*>(a->values) + a->offset;
  const TB* pb = static_cast<const TB*>(b->values) + b->offset;
  int64_t ntiles = (n + 4095) >> 12;
  dim3 grid(grid_for(ntiles * 64, 256)), block(256);
  if (scalar) hipLaunchKernelGGL((k_compare<TA, TB, TC, OP, true>), grid, block, 0, st, pa, pb, out, n);
  else hipLaunchKernelGGL((k_compare<TA, TB, TC, OP, false>), grid, block, 0, st, pa, pb, out, n);
}
template <typename TA, typename TB, typename TC>
static void launch_compare_op(int op, const pdx_column* a, const pdx_column* b, int scalar, uint8_t* out, hipStream_t st) {
  switch (op) {
    case PDX_EQ: launch_compare_sb<TA, TB, TC, PDX_EQ>(a, b, scalar, out, st); break;
    case PDX_NE: launch_compare_sb<TA, TB, TC, PDX_NE>(a, b, scalar, out, st); break;
    case PDX_LT: launch_compare_sb<TA, TB, TC, PDX_LT>(a, b, scalar, out, st); break;
    case PDX_LE: launch_compare_sb<TA, TB, TC, PDX_LE>(a, b, scalar, out, st); break;
    case PDX_GT: launch_compare_sb<TA, TB, TC, PDX_GT>(a, b, scalar, out, st); break;
    default: launch_compare_sb<TA, TB, TC, PDX_GE>(a, b, scalar, out, st); break;
  }
}

// ---------------------------------------------------------------- logical on bit-packed bools
// mode 0: and, 1: or, 2: invert(a)
__global__ void k_logical(const uint8_t* __restrict__ a, int64_t aoff, int64_t alimit, const uint8_t* __restrict__ b, int64_t boff,
                          int64_t blimit, int mode, int64_t n, uint8_t* __restrict__ out) {
  int64_t nwords = (n + 63) >> 6;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += stride) {
    int64_t base = w << 6;
    uint64_t x = load_bits64(a, aoff + base, alimit);
    uint64_t r;
    if (mode == 2) r = ~x;
    else {
      uint64_t y = load_bits64(b, boff + base, blimit);
      r = mode == 0 ? (x & y) : (x | y);
    }
    int64_t remain = n - base;
    if (remain >= 64) {
      reinterpret_cast<uint64_t*>(out)[w] = r;
    } else {
      r &= (1ull << remain) - 1ull;
      int nbytes = (int)((remain + 7) >> 3);
      for (int k = 0; k < nbytes; ++k) out[(w << 3) + k] = (uint8_t)(r >> (8 * k));
    }
  }
}

static int check_numeric_pair(const pdx_column* a, const pdx_column* b, int scalar_side, const char* what) {
  PDX_TRY(check_column(a, what));
  PDX_TRY(check_column(b, what));
  auto ok = [](int dt) { return dt == PDX_INT64 || dt == PDX_FLOAT64; };
  if (!ok(a->dtype) || !ok(b->dtype)) return fail(PDX_NOT_IMPLEMENTED, std::string(what) + ": only int64/float64 operands are supported");
  if (scalar_side < 0 || scalar_side > PDX_SCALAR_LHS) return fail(PDX_INVALID, std::string(what) + ": scalar side must be 0 (none), 1 (rhs) or 2 (lhs)");
  if (scalar_side) {
    if ((scalar_side == PDX_SCALAR_LHS ? a : b)->length != 1) return fail(PDX_INVALID, std::string(what) + ": scalar operand must have length 1");
  } else if (a->length != b->length) {
    return fail(PDX_INVALID, std::string(what) + ": array lengths differ: " + std::to_string(a->length) + " vs " + std::to_string(b->length));
  }
  return PDX_OK;
}

}  // namespace pdx

using namespace pdx;

// ---------------------------------------------------------------- the implicit int64 -> float64 promotion is a CHECKED cast
// Arrow's DispatchBest inserts a safe cast when an int64 operand meets a float64 one (add ... divide, the comparisons, if_else): a VALID
// value outside +-2^53 fails the whole call with "Integer value ... not in range" (pinned against Arrow C++ 25 by
// tests/cpp/arrow_bridge_test.cpp; null slots are not looked at, +-2^53 themselves pass).  One pass over the int64 operand, mixed-type
// calls only.
__global__ void __launch_bounds__(256) k_int_fits_f64(const long long* __restrict__ v, const uint8_t* __restrict__ valid, int64_t voff, int64_t n,
                                                      unsigned long long* __restrict__ err /* [0] flag, [1] an offending value */) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const long long x = v[i];
    if ((x > 9007199254740992ll || x < -9007199254740992ll) && (!valid || bit_get(valid, voff + i))) {
      err[1] = (unsigned long long)x;
      err[0] = 1ull;
    }
  }
}
static int check_promotion(const pdx_column* a, const pdx_column* b, hipStream_t st) {
  if (a->dtype == b->dtype) return PDX_OK;
  const pdx_column* c = a->dtype == PDX_INT64 ? a : b;  // the operand that is cast
  if (c->dtype != PDX_INT64 || c->length == 0) return PDX_OK;
  Scratch s;
  unsigned long long* err = s.get<unsigned long long>(2);
  PDX_SCRATCH_CHECK(s);
  PDX_HIP(hipMemsetAsync(err, 0, 2 * sizeof(unsigned long long), st));
  hipLaunchKernelGGL(k_int_fits_f64, dim3(grid_for(c->length, 256, 8)), dim3(256), 0, st, static_cast<const long long*>(c->values) + c->offset, validity_or_null(c),
                     c->offset, c->length, err);
  PDX_LAUNCH_CHECK();
  unsigned long long h[2] = {0, 0};
  PDX_HIP(hipMemcpyAsync(h, err, sizeof(h), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
  if (h[0]) return fail(PDX_INVALID, "Integer value " + std::to_string((long long)h[1]) + " not in range: -9007199254740992 to 9007199254740992");
  return PDX_OK;
}

// ---------------------------------------------------------------- functions of one column
constexpr int kPowerOp = 100;  // internal op code of pdx_power
template <int OP, typename TI, typename TO>
__global__ void __launch_bounds__(256) k_unary(const TI* __restrict__ a, TO* __restrict__ out, int64_t n, double expo, const uint8_t* __restrict__ valid,
                                               int64_t voff, unsigned long long* __restrict__ err /* [0] flag, [1] an offending value */) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const TI x = a[i];
    TO r;
    if constexpr (OP == PDX_NEGATE) {
      if constexpr (__is_same(TI, double)) r = -x;
      else r = (TO)(0ull - (unsigned long long)x);
    } else if constexpr (OP == PDX_ABS) {
      if constexpr (__is_same(TI, double)) r = __builtin_fabs(x);
      else if constexpr (__is_same(TI, unsigned long long)) r = x;
      else r = x < 0 ? (TO)(0ull - (unsigned long long)x) : x;
    } else if constexpr (OP == PDX_SIGN) {
      if constexpr (__is_same(TI, double)) r = x != x ? x : (x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : 0.0));
      else if constexpr (__is_same(TI, unsigned long long)) r = x != 0;
      else r = (x > 0) - (x < 0);
    } else if constexpr (OP == PDX_BIT_NOT) {
      r = (TO)~(unsigned long long)x;
    } else {  // SQRT / EXP / power: float64 arithmetic; integers are cast first (Arrow: safe cast, only +-2^53 is exact)
      double d;
      if constexpr (__is_same(TI, double)) {
        d = x;
      } else {
        bool bad;
        if constexpr (__is_same(TI, unsigned long long)) bad = x > (1ull << 53);
        else bad = x > (1ll << 53) || x < -(1ll << 53);
        if (bad && (!valid || bit_get(valid, voff + i))) {
          err[1] = (unsigned long long)x;
          err[0] = 1ull;
        }
        d = (double)x;
      }
      if constexpr (OP == PDX_SQRT) {
        // Arrow: a negative operand gives the positive quiet NaN; a NaN operand comes back quieted with its payload (x86 sqrtsd)
        if (d < 0.0) r = __longlong_as_double(0x7FF8000000000000ll);
        else if (d != d) r = __longlong_as_double(__double_as_longlong(d) | 0x0008000000000000ll);
        else r = __builtin_sqrt(d);
      }
      else if constexpr (OP == PDX_EXP) r = exp(d);
      else r = pow(d, expo);
    }
    out[i] = r;
  }
}
template <int OP>
static int launch_unary(const pdx_column* a, pdx_mut_column* out, double expo, unsigned long long* err, hipStream_t st) {
  const int64_t n = a->length;
  const dim3 grid(grid_for(n, 256, 4)), block(256);
  const uint8_t* valid = validity_or_null(a);
  constexpr bool to_f64 = OP == PDX_SQRT || OP == PDX_EXP || OP == kPowerOp;
  if (a->dtype == PDX_FLOAT64) {
    if constexpr (OP != PDX_BIT_NOT)
      hipLaunchKernelGGL((k_unary<OP, double, double>), grid, block, 0, st, static_cast<const double*>(a->values) + a->offset,
                         static_cast<double*>(out->values), n, expo, valid, a->offset, err);
  } else if (a->dtype == PDX_UINT64) {
    const unsigned long long* in = static_cast<const unsigned long long*>(a->values) + a->offset;
    if constexpr (to_f64) hipLaunchKernelGGL((k_unary<OP, unsigned long long, double>), grid, block, 0, st, in, static_cast<double*>(out->values), n, expo, valid, a->offset, err);
    else if constexpr (OP == PDX_SIGN) hipLaunchKernelGGL((k_unary<OP, unsigned long long, long long>), grid, block, 0, st, in, static_cast<long long*>(out->values), n, expo, valid, a->offset, err);
    else hipLaunchKernelGGL((k_unary<OP, unsigned long long, unsigned long long>), grid, block, 0, st, in, static_cast<unsigned long long*>(out->values), n, expo, valid, a->offset, err);
  } else {
    const long long* in = static_cast<const long long*>(a->values) + a->offset;
    if constexpr (to_f64) hipLaunchKernelGGL((k_unary<OP, long long, double>), grid, block, 0, st, in, static_cast<double*>(out->values), n, expo, valid, a->offset, err);
    else hipLaunchKernelGGL((k_unary<OP, long long, long long>), grid, block, 0, st, in, static_cast<long long*>(out->values), n, expo, valid, a->offset, err);
  }
  PDX_LAUNCH_CHECK();
  return PDX_OK;
}
static int unary_impl(int op, const pdx_column* a, double expo, pdx_mut_column* out, void* stream, const char* who) {
  PDX_TRY(check_column(a, who));
  if (a->dtype != PDX_INT64 && a->dtype != PDX_UINT64 && a->dtype != PDX_FLOAT64)
    return fail(PDX_NOT_IMPLEMENTED, std::string(who) + ": input must be int64, uint64 or float64");
  if (op != kPowerOp && (op < PDX_NEGATE || op > PDX_BIT_NOT)) return fail(PDX_INVALID, std::string(who) + ": unknown op");
  if (op == PDX_BIT_NOT && a->dtype == PDX_FLOAT64) return fail(PDX_NOT_IMPLEMENTED, "Function 'bit_wise_not' has no kernel matching input types (double)");
  const bool to_f64 = op == PDX_SQRT || op == PDX_EXP || op == kPowerOp;
  const int out_dt = to_f64 ? PDX_FLOAT64 : (op == PDX_SIGN && a->dtype != PDX_FLOAT64) ? PDX_INT64 : a->dtype;
  if (!out || out->length < a->length || out->dtype != out_dt) return fail(PDX_INVALID, std::string(who) + ": output dtype / length do not match the result");
  const bool has_nulls = validity_or_null(a) != nullptr;
  if (has_nulls && !out->validity) return fail(PDX_INVALID, std::string(who) + ": input carries nulls but output has no validity buffer");
  hipStream_t st = as_stream(stream);
  const int64_t n = a->length;
  out->length = n;
  out->null_count = has_nulls ? -1 : 0;
  if (n == 0) return PDX_OK;
  if (!out->values) return fail(PDX_INVALID, std::string(who) + ": null output buffer");
  const bool need_err = to_f64 && a->dtype != PDX_FLOAT64;
  Scratch scratch;
  unsigned long long* err = nullptr;
  if (need_err) {
    err = scratch.get<unsigned long long>(2);
    PDX_SCRATCH_CHECK(scratch);
    PDX_HIP(hipMemsetAsync(err, 0, 2 * sizeof(unsigned long long), st));
  }
  switch (op) {
    case PDX_NEGATE: PDX_TRY(launch_unary<PDX_NEGATE>(a, out, expo, err, st)); break;
    case PDX_ABS: PDX_TRY(launch_unary<PDX_ABS>(a, out, expo, err, st)); break;
    case PDX_SIGN: PDX_TRY(launch_unary<PDX_SIGN>(a, out, expo, err, st)); break;
    case PDX_SQRT: PDX_TRY(launch_unary<PDX_SQRT>(a, out, expo, err, st)); break;
    case PDX_EXP: PDX_TRY(launch_unary<PDX_EXP>(a, out, expo, err, st)); break;
    case PDX_BIT_NOT: PDX_TRY(launch_unary<PDX_BIT_NOT>(a, out, expo, err, st)); break;
    default: PDX_TRY(launch_unary<kPowerOp>(a, out, expo, err, st)); break;
  }
  if (out->validity) PDX_TRY(launch_validity_and(a, nullptr, 0, n, static_cast<uint8_t*>(out->validity), st));
  if (need_err) {
    unsigned long long h[2] = {0, 0};
    PDX_HIP(hipMemcpyAsync(h, err, sizeof(h), hipMemcpyDeviceToHost, st));
    PDX_HIP(hipStreamSynchronize(st));
    if (h[0]) {
      const std::string v = a->dtype == PDX_UINT64 ? std::to_string(h[1]) : std::to_string((long long)h[1]);
      return fail(PDX_INVALID, "Integer value " + v + " not in range: " + (a->dtype == PDX_UINT64 ? "0" : "-9007199254740992") + " to 9007199254740992");
    }
  }
  return PDX_OK;
}

// ---------------------------------------------------------------- if_else(cond, a, b)
// SCALAR as in k_binary.  Values: one grid-stride stream (cond bit -> a or b).  Validity: one thread per 64-row output word from the
// words of cond, its validity and the operands' validity: valid = cond_valid & (cond ? a_valid : b_valid).
template <typename TA, typename TB, typename TO, int SCALAR>
__global__ void __launch_bounds__(256) k_if_else(const uint8_t* __restrict__ cond, int64_t coff, int64_t climit, const uint8_t* __restrict__ cvalid,
                                                 const TA* __restrict__ a, const uint8_t* __restrict__ va, int64_t aoff, int64_t alimit,
                                                 const TB* __restrict__ b, const uint8_t* __restrict__ vb, int64_t boff, int64_t blimit,
                                                 TO* __restrict__ out, uint8_t* __restrict__ out_valid, int64_t n) {
  constexpr bool SA = SCALAR == 2, SB = SCALAR == 1;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const TO as = SA ? Conv<TO>::from(a[0]) : TO(0), bs = SB ? Conv<TO>::from(b[0]) : TO(0);
  const bool as_valid = !SA || !va || bit_get(va, aoff), bs_valid = !SB || !vb || bit_get(vb, boff);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const bool c = bit_get(cond, coff + i);
    out[i] = c ? (SA ? as : Conv<TO>::from(a[i])) : (SB ? bs : Conv<TO>::from(b[i]));
  }
  if (!out_valid) return;
  const int64_t nwords = (n + 63) >> 6;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += stride) {
    const int64_t base = w << 6;
    const uint64_t c = load_bits64(cond, coff + base, climit);
    const uint64_t cv = cvalid ? load_bits64(cvalid, coff + base, climit) : ~0ull;
    const uint64_t av = SA ? (as_valid ? ~0ull : 0ull) : (va ? load_bits64(va, aoff + base, alimit) : ~0ull);
    const uint64_t bv = SB ? (bs_valid ? ~0ull : 0ull) : (vb ? load_bits64(vb, boff + base, blimit) : ~0ull);
    uint64_t r = cv & ((c & av) | (~c & bv));
    const int64_t remain = n - base;
    if (remain >= 64) {
      reinterpret_cast<uint64_t*>(out_valid)[w] = r;
    } else {
      r &= (1ull << remain) - 1ull;
      const int nbytes = (int)((remain + 7) >> 3);
      for (int k = 0; k < nbytes; ++k) out_valid[(w << 3) + k] = (uint8_t)(r >> (8 * k));
    }
  }
}
template <typename TA, typename TB, typename TO>
static void launch_if_else(const pdx_column* cond, const pdx_column* a, const pdx_column* b, int scalar, pdx_mut_column* out, int64_t n, hipStream_t st) {
  const dim3 grid(grid_for(n, 256, 4)), block(256);
  const uint8_t* cbits = static_cast<const uint8_t*>(cond->values);
  const TA* pa = static_cast<const TA*>(a->values) + a->offset;
  const TB* pb = static_cast<const TB*>(b->values) + b->offset;
#define IE_ARGS cbits, cond->offset, cond->offset + cond->length, validity_or_null(cond), pa, validity_or_null(a), a->offset, a->offset + a->length, pb, \
                validity_or_null(b), b->offset, b->offset + b->length, static_cast<TO*>(out->values), static_cast<uint8_t*>(out->validity), n
  if (scalar == PDX_SCALAR_RHS) hipLaunchKernelGGL((k_if_else<TA, TB, TO, 1>), grid, block, 0, st, IE_ARGS);
  else if (scalar == PDX_SCALAR_LHS) hipLaunchKernelGGL((k_if_else<TA, TB, TO, 2>), grid, block, 0, st, IE_ARGS);
  else hipLaunchKernelGGL((k_if_else<TA, TB, TO, 0>), grid, block, 0, st, IE_ARGS);
#undef IE_ARGS
}

extern "C" {

int pdx_if_else(const pdx_column* cond, const pdx_column* a, const pdx_column* b, int scalar_side, pdx_mut_column* out, void* stream) {
  PDX_TRY(check_column(cond, "pdx_if_else"));
  if (cond->dtype != PDX_BOOL) return fail(PDX_INVALID, "pdx_if_else: the condition must be PDX_BOOL");
  PDX_TRY(check_column(a, "pdx_if_else"));
  PDX_TRY(check_column(b, "pdx_if_else"));
  auto num = [](int dt) { return dt == PDX_INT64 || dt == PDX_FLOAT64; };
  if (!num(a->dtype) || !num(b->dtype)) return fail(PDX_NOT_IMPLEMENTED, "pdx_if_else: only int64/float64 operands are supported");
  if (scalar_side < 0 || scalar_side > PDX_SCALAR_LHS) return fail(PDX_INVALID, "pdx_if_else: scalar side must be 0 (none), 1 (rhs) or 2 (lhs)");
  const int64_t n = cond->length;
  if ((scalar_side == PDX_SCALAR_LHS ? a->length != 1 : a->length != n) || (scalar_side == PDX_SCALAR_RHS ? b->length != 1 : b->length != n))
    return fail(PDX_INVALID, "pdx_if_else: Array arguments must all be the same length (a scalar operand has length 1)");
  const bool is_f = a->dtype == PDX_FLOAT64 || b->dtype == PDX_FLOAT64;
  if (!out || out->length < n || out->dtype != (is_f ? PDX_FLOAT64 : PDX_INT64)) return fail(PDX_INVALID, "pdx_if_else: output dtype / length do not match the result");
  const bool has_nulls = validity_or_null(cond) || validity_or_null(a) || validity_or_null(b);
  if (has_nulls && !out->validity) return fail(PDX_INVALID, "pdx_if_else: inputs carry nulls but output has no validity buffer");
  hipStream_t st = as_stream(stream);
  out->length = n;
  out->null_count = has_nulls ? -1 : 0;
  if (n == 0) return PDX_OK;
  if (!out->values) return fail(PDX_INVALID, "pdx_if_else: null output buffer");
  PDX_TRY(check_promotion(a, b, st));
  pdx_mut_column o = *out;
  if (!is_f) launch_if_else<int64_t, int64_t, int64_t>(cond, a, b, scalar_side, &o, n, st);
  else if (a->dtype == PDX_FLOAT64 && b->dtype == PDX_FLOAT64) launch_if_else<double, double, double>(cond, a, b, scalar_side, &o, n, st);
  else if (a->dtype == PDX_FLOAT64) launch_if_else<double, int64_t, double>(cond, a, b, scalar_side, &o, n, st);
  else launch_if_else<int64_t, double, double>(cond, a, b, scalar_side, &o, n, st);
  PDX_LAUNCH_CHECK();
  return PDX_OK;
}

// Cast(int64 -> float64): value by value (static_cast<double>), nulls carried over
__global__ void __launch_bounds__(256) k_cast_i64_f64(const long long* __restrict__ a, int64_t n, double* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (double)a[i];
}
int pdx_cast_f64(const pdx_column* a, int checked, pdx_mut_column* out, void* stream) {
  PDX_TRY(check_column(a, "pdx_cast_f64"));
  if (a->dtype != PDX_INT64 && a->dtype != PDX_FLOAT64) return fail(PDX_NOT_IMPLEMENTED, "pdx_cast_f64: input must be int64 or float64");
  if (!out || out->length < a->length || out->dtype != PDX_FLOAT64) return fail(PDX_INVALID, "pdx_cast_f64: output must be PDX_FLOAT64 of the input length");
  const bool has_nulls = validity_or_null(a) != nullptr;
  if (has_nulls && !out->validity) return fail(PDX_INVALID, "pdx_cast_f64: input carries nulls but output has no validity buffer");
  hipStream_t st = as_stream(stream);
  const int64_t n = a->length;
  out->length = n;
  out->null_count = has_nulls ? -1 : 0;
  if (n == 0) return PDX_OK;
  if (!out->values) return fail(PDX_INVALID, "pdx_cast_f64: null output buffer");
  if (a->dtype == PDX_FLOAT64) {
    PDX_HIP(hipMemcpyAsync(out->values, static_cast<const double*>(a->values) + a->offset, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
  } else {
    if (checked) {
      pdx_column as_f = *a;  // (check_promotion looks at the int64 operand of a mixed pair)
      as_f.dtype = PDX_FLOAT64;
      PDX_TRY(check_promotion(a, &as_f, st));
    }
    hipLaunchKernelGGL(k_cast_i64_f64, dim3(grid_for(n, 256, 8)), dim3(256), 0, st, static_cast<const long long*>(a->values) + a->offset, n,
                       static_cast<double*>(out->values));
    PDX_LAUNCH_CHECK();
  }
  if (out->validity) PDX_TRY(launch_validity_and(a, nullptr, 0, n, static_cast<uint8_t*>(out->validity), st));
  return PDX_OK;
}

int pdx_unary(int op, const pdx_column* a, pdx_mut_column* out, void* stream) { return unary_impl(op, a, 0.0, out, stream, "pdx_unary"); }
int pdx_power(const pdx_column* a, double exponent, pdx_mut_column* out, void* stream) { return unary_impl(kPowerOp, a, exponent, out, stream, "pdx_power"); }

int pdx_binary(int op, const pdx_column* a, const pdx_column* b, int b_is_scalar, pdx_mut_column* out, void* stream) {
  PDX_TRY(check_numeric_pair(a, b, b_is_scalar, "pdx_binary"));
  if (op < PDX_ADD || op > PDX_SHIFT_RIGHT) return fail(PDX_INVALID, "pdx_binary: unknown op");
  if (op >= PDX_BIT_OR && (a->dtype != PDX_INT64 || b->dtype != PDX_INT64))
    return fail(PDX_NOT_IMPLEMENTED, "pdx_binary: bit-wise operators and shifts have no kernel matching floating-point input types");
  const pdx_column* arr = b_is_scalar == PDX_SCALAR_LHS ? b : a;  // the operand that gives the result its length
  if (!out || out->length < arr->length) return fail(PDX_INVALID, "pdx_binary: output too small");
  const bool is_f = a->dtype == PDX_FLOAT64 || b->dtype == PDX_FLOAT64;
  const int out_dt = is_f ? PDX_FLOAT64 : PDX_INT64;
  if (out->dtype != out_dt) return fail(PDX_INVALID, "pdx_binary: output dtype must be the promoted input dtype");
  const bool has_nulls = validity_or_null(a) || validity_or_null(b);
  if (has_nulls && !out->validity) return fail(PDX_INVALID, "pdx_binary: inputs carry nulls but output has no validity buffer");
  hipStream_t st = as_stream(stream);
  int64_t n = arr->length;
  out->length = n;
  out->null_count = has_nulls ? -1 : 0;
  if (n == 0) return PDX_OK;
  if (!out->values) return fail(PDX_INVALID, "pdx_binary: null output buffer");
  PDX_TRY(check_promotion(a, b, st));
  const bool need_err = (op == PDX_DIV) && !is_f;
  Scratch scratch;
  unsigned long long* err = nullptr;
  if (need_err) {
    err = scratch.get<unsigned long long>(1);
    PDX_SCRATCH_CHECK(scratch);
    PDX_HIP(hipMemsetAsync(err, 0, sizeof(unsigned long long), st));
  }
  if (is_f) {
    double* o = static_cast<double*>(out->values);
    if (a->dtype == PDX_FLOAT64 && b->dtype == PDX_FLOAT64) launch_binary_op<double, double, double>(op, a, b, b_is_scalar, o, err, st);
    else if (a->dtype == PDX_FLOAT64) launch_binary_op<double, int64_t, double>(op, a, b, b_is_scalar, o, err, st);
    else launch_binary_op<int64_t, double, double>(op, a, b, b_is_scalar, o, err, st);
  } else {
    launch_binary_op<int64_t, int64_t, int64_t>(op, a, b, b_is_scalar, static_cast<int64_t*>(out->values), err, st);
  }
  PDX_LAUNCH_CHECK();
  if (out->validity) {  // AND is symmetric: the array operand goes first, the scalar's one bit is broadcast
    if (b_is_scalar == PDX_SCALAR_LHS) PDX_TRY(launch_validity_and(b, a, 1, n, static_cast<uint8_t*>(out->validity), st));
    else PDX_TRY(launch_validity_and(a, b, b_is_scalar, n, static_cast<uint8_t*>(out->validity), st));
  }
  if (need_err) {
    unsigned long long h = 0;
    PDX_HIP(hipMemcpyAsync(&h, err, sizeof(h), hipMemcpyDeviceToHost, st));
    PDX_HIP(hipStreamSynchronize(st));
    if (h) return fail(PDX_INVALID, "divide by zero");
  }
  return PDX_OK;
}

int pdx_compare(int op, const pdx_column* a, const pdx_column* b, int b_is_scalar, pdx_mut_column* out, void* stream) {
  PDX_TRY(check_numeric_pair(a, b, b_is_scalar, "pdx_compare"));
  if (op < PDX_EQ || op > PDX_GE) return fail(PDX_INVALID, "pdx_compare: unknown op");
  if (b_is_scalar == PDX_SCALAR_LHS) {
    // scalar OP array == array OP' scalar with the mirrored relation (booleans carry no NaN payload, so this is exact)
    static const int mirrored[6] = {PDX_EQ, PDX_NE, PDX_GT, PDX_GE, PDX_LT, PDX_LE};
    const pdx_column* t = a;
    a = b;
    b = t;
    op = mirrored[op];
    b_is_scalar = PDX_SCALAR_RHS;
  }
  if (!out || out->length < a->length || out->dtype != PDX_BOOL) return fail(PDX_INVALID, "pdx_compare: output must be PDX_BOOL of the input length");
  const bool has_nulls = validity_or_null(a) || validity_or_null(b);
  if (has_nulls && !out->validity) return fail(PDX_INVALID, "pdx_compare: inputs carry nulls but output has no validity buffer");
  hipStream_t st = as_stream(stream);
  int64_t n = a->length;
  out->length = n;
  out->null_count = has_nulls ? -1 : 0;
  if (n == 0) return PDX_OK;
  if (!out->values) return fail(PDX_INVALID, "pdx_compare: null output buffer");
  PDX_TRY(check_promotion(a, b, st));
  uint8_t* o = static_cast<uint8_t*>(out->values);
  if (a->dtype == PDX_FLOAT64 && b->dtype == PDX_FLOAT64) launch_compare_op<double, double, double>(op, a, b, b_is_scalar, o, st);
  else if (a->dtype == PDX_FLOAT64) launch_compare_op<double, int64_t, double>(op, a, b, b_is_scalar, o, st);
  else if (b->dtype == PDX_FLOAT64) launch_compare_op<int64_t, double, double>(op, a, b, b_is_scalar, o, st);
  else launch_compare_op<int64_t, int64_t, int64_t>(op, a, b, b_is_scalar, o, st);
  PDX_LAUNCH_CHECK();
  if (out->validity) PDX_TRY(launch_validity_and(a, b, b_is_scalar, n, static_cast<uint8_t*>(out->validity), st));
  return PDX_OK;
}

static int logical_impl(int mode, const pdx_column* a, const pdx_column* b, pdx_mut_column* out, void* stream, const char* what) {
  PDX_TRY(check_column(a, what));
  if (a->dtype != PDX_BOOL) return fail(PDX_INVALID, std::string(what) + ": operands must be PDX_BOOL");
  if (b) {
    PDX_TRY(check_column(b, what));
    if (b->dtype != PDX_BOOL) return fail(PDX_INVALID, std::string(what) + ": operands must be PDX_BOOL");
    if (a->length != b->length) return fail(PDX_INVALID, std::string(what) + ": array lengths differ");
  }
  if (!out || out->length < a->length || out->dtype != PDX_BOOL) return fail(PDX_INVALID, std::string(what) + ": output must be PDX_BOOL of the input length");
  const bool has_nulls = validity_or_null(a) || (b && validity_or_null(b));
  if (has_nulls && !out->validity) return fail(PDX_INVALID, std::string(what) + ": inputs carry nulls but output has no validity buffer");
  hipStream_t st = as_stream(stream);
  int64_t n = a->length;
  out->length = n;
  out->null_count = has_nulls ? -1 : 0;
  if (n == 0) return PDX_OK;
  int64_t nwords = (n + 63) >> 6;
  hipLaunchKernelGGL(k_logical, dim3(grid_for(nwords, 256)), dim3(256), 0, st, static_cast<const uint8_t*>(a->values), a->offset,
                     a->offset + a->length, b ? static_cast<const uint8_t*>(b->values) : nullptr, b ? b->offset : 0,
                     b ? b->offset + b->length : 0, mode, n, static_cast<uint8_t*>(out->values));
  PDX_LAUNCH_CHECK();
  if (out->validity) PDX_TRY(launch_validity_and(a, b, 0, n, static_cast<uint8_t*>(out->validity), st));
  return PDX_OK;
}

int pdx_logical(int op, const pdx_column* a, const pdx_column* b, pdx_mut_column* out, void* stream) {
  if (op != PDX_AND && op != PDX_OR) return fail(PDX_INVALID, "pdx_logical: unknown op");
  if (!b) return fail(PDX_INVALID, "pdx_logical: null column");
  return logical_impl(op == PDX_AND ? 0 : 1, a, b, out, stream, "pdx_logical");
}
int pdx_invert(const pdx_column* a, pdx_mut_column* out, void* stream) { return logical_impl(2, a, nullptr, out, stream, "pdx_invert"); }

}  // extern "C"
