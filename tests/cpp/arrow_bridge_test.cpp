// arrow_bridge_test.cpp -- TEST INFRASTRUCTURE: the reference-side binding of INTEGRATION.md section 1 compiled against REAL Arrow types
// (Arrow C++ 25 from the pyarrow wheel: the third-party library the reference forwards to; nothing of the reference is compiled) and
// run against Arrow's own kernels on seeded inputs.
//
// The reference's boundary on this path is arrow::compute::CallFunction(name, {Datum...}, options) and arrow::compute::Grouper
// (src/series.cpp:19-33, src/ndframe.cpp:26-31, src/dataframe.cpp:461-492, 1571-1600).  pd::pdx below is what a maintainer adds:
//   DeviceArray(const arrow::ArrayData&)   buffers uploaded whole, the slice `offset` kept (values AND validity are addressed with it)
//   CallFunction(name, args, options)      same names, Datum in / Datum out, dispatching to the pdx_* entry points
//   Grouper                                Make / Consume / GetUniques / num_groups over pdx_groupby_*
// main() compares every result with arrow::compute::CallFunction / arrow::compute::Grouper on the same arrays (group ids up to Arrow's
// permutation inside a mini-batch's block of new ids, everything else exactly): sliced arrays with a
// non-zero offset (also one that is not a multiple of 8: bitmaps are bit-shifted), nulls, NaN, 0 / 1 / 17 / 1e5 rows; doubles bit for bit.
//
// Build (tests/test_gpu_arrow_bridge.py): g++ -std=c++20 -I<pyarrow>/include -Iinclude arrow_bridge_test.cpp -l:libarrow_compute.so.2500
//   -l:libarrow.so.2500 -lpdx_hip.  Exit status 0 = every comparison held.
#include <arrow/api.h>
#include <arrow/compute/api.h>
#include <arrow/compute/row/grouper.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "pdx/abi.h"

namespace cp = arrow::compute;

namespace pd::pdx {

inline void ThrowOnPdx(int status) {  // same convention as pd::ThrowOnFailure (reference src/core.h:181-194)
  if (status != PDX_OK) throw std::runtime_error(pdx_last_error());
}
inline arrow::Status ToStatus(int status) {
  if (status == PDX_OK) return arrow::Status::OK();
  if (status == PDX_INDEX_ERROR) return arrow::Status::IndexError(pdx_last_error());
  if (status == PDX_NOT_IMPLEMENTED) return arrow::Status::NotImplemented(pdx_last_error());
  return arrow::Status::Invalid(pdx_last_error());
}
inline int DtypeOf(const arrow::DataType& t) {
  switch (t.id()) {
    case arrow::Type::DOUBLE: return PDX_FLOAT64;
    case arrow::Type::BOOL: return PDX_BOOL;
    case arrow::Type::TIMESTAMP: return PDX_TIMESTAMP_NS;
    case arrow::Type::UINT64: return PDX_UINT64;
    case arrow::Type::INT64: return PDX_INT64;
    default: throw std::runtime_error("pd::pdx: unsupported Arrow type " + t.ToString());
  }
}

// Upload the buffers of an arrow::ArrayData once (GPUSeries ctor precedent, reference src/cudf/series.cpp:22-33)
struct DeviceArray {
  pdx_column col{};
  void* values = nullptr;
  void* validity = nullptr;
  explicit DeviceArray(const arrow::ArrayData& a) {
    const int64_t n = a.length, off = a.offset;
    col.dtype = DtypeOf(*a.type);
    col.length = n;
    col.offset = off;
    col.null_count = a.GetNullCount();
    const size_t bitmap_bytes = (size_t)(n + off + 7) / 8;
    const size_t vbytes = col.dtype == PDX_BOOL ? bitmap_bytes : (size_t)(n + off) * 8;
    ThrowOnPdx(pdx_malloc(&values, vbytes + 16));
    if (vbytes) ThrowOnPdx(pdx_to_device(values, a.buffers[1]->data(), vbytes, nullptr));
    if (col.null_count != 0 && a.buffers[0]) {
      ThrowOnPdx(pdx_malloc(&validity, bitmap_bytes + 16));
      ThrowOnPdx(pdx_to_device(validity, a.buffers[0]->data(), bitmap_bytes, nullptr));
    }
    col.values = values;
    col.validity = validity;
  }
  DeviceArray(const DeviceArray&) = delete;
  ~DeviceArray() {
    pdx_free(values);
    pdx_free(validity);
  }
};

// A device output column of `n` rows and its way back into an arrow::Array
struct DeviceOut {
  pdx_mut_column col{};
  int64_t cap;
  DeviceOut(int dtype, int64_t n) : cap(n) {
    col.dtype = dtype;
    col.length = n;
    const size_t vb = dtype == PDX_BOOL ? (size_t)(n + 7) / 8 : (size_t)n * 8;
    ThrowOnPdx(pdx_malloc(&col.values, vb + 64));
    ThrowOnPdx(pdx_malloc(&col.validity, (size_t)(n + 7) / 8 + 64));
  }
  DeviceOut(const DeviceOut&) = delete;
  ~DeviceOut() {
    pdx_free(col.values);
    pdx_free(col.validity);
  }
  std::shared_ptr<arrow::Array> ToArrow(const std::shared_ptr<arrow::DataType>& type) const {
    const int64_t n = col.length;
    const size_t vb = col.dtype == PDX_BOOL ? (size_t)(n + 7) / 8 : (size_t)n * 8, bb = (size_t)(n + 7) / 8;
    auto vbuf = arrow::AllocateBuffer((int64_t)vb + 8).ValueOrDie();
    std::memset(vbuf->mutable_data(), 0, vb + 8);
    if (vb) ThrowOnPdx(pdx_to_host(vbuf->mutable_data(), col.values, vb, nullptr));
    std::shared_ptr<arrow::Buffer> nbuf;
    int64_t nulls = 0;
    if (col.null_count != 0) {
      auto b = arrow::AllocateBuffer((int64_t)bb + 8).ValueOrDie();
      std::memset(b->mutable_data(), 0, bb + 8);
      if (bb) ThrowOnPdx(pdx_to_host(b->mutable_data(), col.validity, bb, nullptr));
      for (int64_t i = 0; i < n; ++i) nulls += !((b->data()[i >> 3] >> (i & 7)) & 1);
      if (nulls) nbuf = std::move(b);
    }
    return arrow::MakeArray(arrow::ArrayData::Make(type, n, {nbuf, std::shared_ptr<arrow::Buffer>(std::move(vbuf))}, nulls));
  }
};

inline std::shared_ptr<arrow::DataType> ArrowType(int dtype) {
  return dtype == PDX_FLOAT64 ? arrow::float64() : dtype == PDX_BOOL ? arrow::boolean() : dtype == PDX_UINT64 ? arrow::uint64() : arrow::int64();
}
inline std::shared_ptr<arrow::Array> ScalarAsArray(const arrow::Scalar& s) { return arrow::MakeArrayFromScalar(s, 1).ValueOrDie(); }

// arrow::compute::CallFunction for the kernel names the reference uses on this path
inline arrow::Result<arrow::Datum> CallFunction(const std::string& name, const std::vector<arrow::Datum>& args, const cp::FunctionOptions* opts = nullptr) {
  static const std::pair<const char*, int> kBinary[] = {{"add", PDX_ADD}, {"subtract", PDX_SUB}, {"multiply", PDX_MUL}, {"divide", PDX_DIV}};
  static const std::pair<const char*, int> kCompare[] = {{"equal", PDX_EQ},   {"not_equal", PDX_NE},  {"less", PDX_LT},
                                                         {"less_equal", PDX_LE}, {"greater", PDX_GT}, {"greater_equal", PDX_GE}};
  static const std::pair<const char*, int> kAgg[] = {{"sum", PDX_AGG_SUM}, {"mean", PDX_AGG_MEAN}, {"min", PDX_AGG_MIN}, {"max", PDX_AGG_MAX}, {"count", PDX_AGG_COUNT}};
  try {
    auto binary_like = [&](int op, bool compare) -> arrow::Result<arrow::Datum> {
      // Series op Series, Series op Scalar (src/series.cpp:25-28), Scalar op Series (src/scalar.cpp:24-41): a scalar is a length-1 column
      const bool a_scalar = args[0].is_scalar(), b_scalar = args[1].is_scalar();
      auto a = a_scalar ? ScalarAsArray(*args[0].scalar()) : args[0].make_array();
      auto b = b_scalar ? ScalarAsArray(*args[1].scalar()) : args[1].make_array();
      DeviceArray da(*a->data()), db(*b->data());
      const int side = a_scalar ? PDX_SCALAR_LHS : b_scalar ? PDX_SCALAR_RHS : PDX_SCALAR_NONE;
      const int64_t n = a_scalar ? b->length() : a->length();
      const int out_dt = compare ? PDX_BOOL : ((da.col.dtype == PDX_FLOAT64 || db.col.dtype == PDX_FLOAT64) ? PDX_FLOAT64 : PDX_INT64);
      DeviceOut out(out_dt, n);
      ARROW_RETURN_NOT_OK(ToStatus(compare ? pdx_compare(op, &da.col, &db.col, side, &out.col, nullptr) : pdx_binary(op, &da.col, &db.col, side, &out.col, nullptr)));
      return arrow::Datum(out.ToArrow(ArrowType(out_dt)));
    };
    for (auto& [nm, op] : kBinary)
      if (name == nm) return binary_like(op, false);
    for (auto& [nm, op] : kCompare)
      if (name == nm) return binary_like(op, true);
    if (name == "and" || name == "or") {
      auto a = args[0].make_array(), b = args[1].make_array();
      DeviceArray da(*a->data()), db(*b->data());
      DeviceOut out(PDX_BOOL, a->length());
      ARROW_RETURN_NOT_OK(ToStatus(pdx_logical(name == "and" ? PDX_AND : PDX_OR, &da.col, &db.col, &out.col, nullptr)));
      return arrow::Datum(out.ToArrow(arrow::boolean()));
    }
    if (name == "invert") {
      auto a = args[0].make_array();
      DeviceArray da(*a->data());
      DeviceOut out(PDX_BOOL, a->length());
      ARROW_RETURN_NOT_OK(ToStatus(pdx_invert(&da.col, &out.col, nullptr)));
      return arrow::Datum(out.ToArrow(arrow::boolean()));
    }
    for (auto& [nm, kind] : kAgg)
      if (name == nm) {  // NDFrame::sum / mean / min / max / count: ScalarAggregateOptions{skip_nulls, min_count = 1} (src/ndframe.cpp:26-31)
        auto a = args[0].make_array();
        DeviceArray da(*a->data());
        pdx_scalar s{};
        ARROW_RETURN_NOT_OK(ToStatus(pdx_aggregate(kind, &da.col, &s, nullptr)));
        if (kind == PDX_AGG_COUNT) return arrow::Datum(std::make_shared<arrow::Int64Scalar>(s.v.i64));
        if (s.dtype == PDX_FLOAT64) return arrow::Datum(s.is_valid ? std::make_shared<arrow::DoubleScalar>(s.v.f64) : std::make_shared<arrow::DoubleScalar>());
        return arrow::Datum(s.is_valid ? std::make_shared<arrow::Int64Scalar>(s.v.i64) : std::make_shared<arrow::Int64Scalar>());
      }
    if (name == "min_max") {  // MinMax at src/resample.cpp:223, GroupBy::min_max: struct{min, max}
      auto a = args[0].make_array();
      DeviceArray da(*a->data());
      pdx_scalar mn{}, mx{};
      ARROW_RETURN_NOT_OK(ToStatus(pdx_aggregate(PDX_AGG_MIN, &da.col, &mn, nullptr)));
      ARROW_RETURN_NOT_OK(ToStatus(pdx_aggregate(PDX_AGG_MAX, &da.col, &mx, nullptr)));
      auto mk = [&](const pdx_scalar& s) -> std::shared_ptr<arrow::Scalar> {
        if (a->type_id() == arrow::Type::DOUBLE) return s.is_valid ? std::make_shared<arrow::DoubleScalar>(s.v.f64) : std::make_shared<arrow::DoubleScalar>();
        return s.is_valid ? std::make_shared<arrow::Int64Scalar>(s.v.i64) : std::make_shared<arrow::Int64Scalar>();
      };
      ARROW_ASSIGN_OR_RAISE(auto st, arrow::StructScalar::Make({mk(mn), mk(mx)}, {"min", "max"}));
      return arrow::Datum(st);
    }
    if (name == "filter" || name == "array_filter") {  // FilterOptions{EMIT_NULL | DROP} (src/dataframe.cpp:461-475, src/series.cpp:130-144)
      auto a = args[0].make_array(), m = args[1].make_array();
      const auto* fo = static_cast<const cp::FilterOptions*>(opts);
      const int emit_null = fo && fo->null_selection_behavior == cp::FilterOptions::EMIT_NULL ? 1 : 0;
      if (m->length() != a->length()) return arrow::Status::Invalid("filter: mask and array differ in length");
      DeviceArray da(*a->data()), dm(*m->data());
      int64_t cnt = 0;
      ARROW_RETURN_NOT_OK(ToStatus(pdx_filter_count(&dm.col, emit_null, &cnt, nullptr)));
      DeviceOut out(da.col.dtype, cnt);
      ARROW_RETURN_NOT_OK(ToStatus(pdx_filter(&da.col, 1, &dm.col, emit_null, &out.col, nullptr)));
      return arrow::Datum(out.ToArrow(a->type()));
    }
    if (name == "take" || name == "array_take") {  // src/dataframe.cpp:477-492
      auto a = args[0].make_array(), idx = args[1].make_array();
      DeviceArray da(*a->data()), di(*idx->data());
      DeviceOut out(da.col.dtype, idx->length());
      ARROW_RETURN_NOT_OK(ToStatus(pdx_take(&da.col, 1, &di.col, &out.col, nullptr)));
      return arrow::Datum(out.ToArrow(a->type()));
    }
  } catch (const std::exception& e) {
    return arrow::Status::Invalid(e.what());
  }
  return arrow::Status::NotImplemented("pd::pdx::CallFunction: ", name);
}

// arrow::compute::Grouper over pdx_groupby_* (GroupBy::makeGroups, src/dataframe.cpp:1571-1600)
class Grouper {
 public:
  ~Grouper() { pdx_groupby_destroy(gb_); }
  // Consume: group ids of every row (uint32, first-occurrence order)
  arrow::Result<arrow::Datum> Consume(const std::shared_ptr<arrow::Array>& keys) {
    key_type_ = keys->type();
    key_.reset(new DeviceArray(*keys->data()));
    ARROW_RETURN_NOT_OK(ToStatus(pdx_groupby_create(&key_->col, nullptr, &gb_)));
    const int64_t n = keys->length();
    void* dids = nullptr;
    ARROW_RETURN_NOT_OK(ToStatus(pdx_malloc(&dids, (size_t)n * 4 + 16)));
    arrow::Status st = ToStatus(pdx_groupby_group_ids(gb_, static_cast<uint32_t*>(dids), nullptr));
    auto buf = arrow::AllocateBuffer(n * 4 + 8).ValueOrDie();
    if (st.ok() && n) st = ToStatus(pdx_to_host(buf->mutable_data(), dids, (size_t)n * 4, nullptr));
    pdx_free(dids);
    ARROW_RETURN_NOT_OK(st);
    return arrow::Datum(arrow::MakeArray(arrow::ArrayData::Make(arrow::uint32(), n, {nullptr, std::shared_ptr<arrow::Buffer>(std::move(buf))}, 0)));
  }
  uint32_t num_groups() const { return (uint32_t)pdx_groupby_num_groups(gb_); }
  arrow::Result<std::shared_ptr<arrow::Array>> GetUniques() {
    DeviceOut out(key_->col.dtype, num_groups());
    ARROW_RETURN_NOT_OK(ToStatus(pdx_groupby_unique_keys(gb_, &out.col, nullptr)));
    return out.ToArrow(key_type_);
  }
  // GROUPBY_AGG / GROUPBY_NUMERIC_AGG (src/pd_core_macros.h:5-147): one column of per-group results in group-id order
  arrow::Result<std::shared_ptr<arrow::Array>> Aggregate(const std::shared_ptr<arrow::Array>& values, int kind) {
    DeviceArray dv(*values->data());
    const int out_dt = kind == PDX_AGG_COUNT ? PDX_INT64 : kind == PDX_AGG_MEAN ? PDX_FLOAT64 : dv.col.dtype;
    DeviceOut out(out_dt, num_groups());
    ARROW_RETURN_NOT_OK(ToStatus(pdx_groupby_agg(gb_, &dv.col, &kind, 1, &out.col, nullptr)));
    return out.ToArrow(ArrowType(out_dt));
  }

 private:
  pdx_groupby* gb_ = nullptr;
  std::unique_ptr<DeviceArray> key_;
  std::shared_ptr<arrow::DataType> key_type_;
};

}  // namespace pd::pdx

// ---------------------------------------------------------------- the comparison harness
static int g_failed = 0, g_checked = 0;
static void Fail(const std::string& what) {
  ++g_failed;
  std::fprintf(stderr, "MISMATCH: %s\n", what.c_str());
}
// bit-exact equality of two arrays (doubles by bit pattern on valid slots, NaN == NaN only with the same bits)
static bool SameArray(const arrow::Array& x, const arrow::Array& y) {
  if (x.length() != y.length() || !x.type()->Equals(*y.type()) || x.null_count() != y.null_count()) return false;
  for (int64_t i = 0; i < x.length(); ++i) {
    if (x.IsNull(i) != y.IsNull(i)) return false;
    if (x.IsNull(i)) continue;
    switch (x.type_id()) {
      case arrow::Type::DOUBLE: {
        const double a = static_cast<const arrow::DoubleArray&>(x).Value(i), b = static_cast<const arrow::DoubleArray&>(y).Value(i);
        if (std::memcmp(&a, &b, 8) != 0) return false;
        break;
      }
      case arrow::Type::BOOL:
        if (static_cast<const arrow::BooleanArray&>(x).Value(i) != static_cast<const arrow::BooleanArray&>(y).Value(i)) return false;
        break;
      case arrow::Type::UINT32:
        if (static_cast<const arrow::UInt32Array&>(x).Value(i) != static_cast<const arrow::UInt32Array&>(y).Value(i)) return false;
        break;
      default:
        if (static_cast<const arrow::Int64Array&>(x).Value(i) != static_cast<const arrow::Int64Array&>(y).Value(i)) return false;
    }
  }
  return true;
}
static bool SameScalar(const arrow::Scalar& a, const arrow::Scalar& b) {
  if (a.is_valid != b.is_valid) return false;
  if (!a.is_valid) return true;
  if (a.type->id() == arrow::Type::DOUBLE && b.type->id() == arrow::Type::DOUBLE) {
    const double x = static_cast<const arrow::DoubleScalar&>(a).value, y = static_cast<const arrow::DoubleScalar&>(b).value;
    return std::memcmp(&x, &y, 8) == 0 || (std::isnan(x) && std::isnan(y));  // (reductions guarantee NaN-ness, not the NaN's bits: DESIGN section 4)
  }
  if (a.type->id() == arrow::Type::STRUCT) {
    const auto& sa = static_cast<const arrow::StructScalar&>(a);
    const auto& sb = static_cast<const arrow::StructScalar&>(b);
    for (size_t i = 0; i < sa.value.size(); ++i)
      if (!SameScalar(*sa.value[i], *sb.value[i])) return false;
    return true;
  }
  return a.Equals(b);
}
static void Compare(const std::string& what, const arrow::Result<arrow::Datum>& got, const arrow::Result<arrow::Datum>& want) {
  ++g_checked;
  if (got.ok() != want.ok()) return Fail(what + ": status differs: " + got.status().ToString() + " vs " + want.status().ToString());
  if (!want.ok()) {  // both failed: the reference's tests only require that the call throws; the leading text is Arrow's
    if (got.status().code() != want.status().code()) Fail(what + ": error class differs: " + got.status().ToString() + " vs " + want.status().ToString());
    return;
  }
  if (want->is_scalar()) {
    if (!got->is_scalar() || !SameScalar(*got->scalar(), *want->scalar())) Fail(what + ": scalar " + (got->is_scalar() ? got->scalar()->ToString() : "?") + " vs " + want->scalar()->ToString());
    return;
  }
  if (!SameArray(*got->make_array(), *want->make_array())) Fail(what + ": arrays differ (" + std::to_string(want->length()) + " rows)");
}

struct Inputs {
  std::shared_ptr<arrow::Array> f64a, f64b, i64a, i64b, boola, boolb, keys, idx;
};
// n rows sliced out of longer arrays at `off` (the parent buffers stay: Arrow slices are zero-copy offsets); nulls when asked
static Inputs MakeInputs(int64_t n, int64_t off, bool nulls, uint64_t seed) {
  std::mt19937_64 rng(seed);
  const int64_t N = n + off + 5;
  arrow::DoubleBuilder fa, fb;
  arrow::Int64Builder ia, ib, kb, xb;
  arrow::BooleanBuilder ba, bb;
  static const double special[] = {0.0, -0.0, NAN, INFINITY, -INFINITY, 1e308, -1e308, 5e-324};
  for (int64_t i = 0; i < N; ++i) {
    auto null = [&] { return nulls && rng() % 7 == 0; };
    auto dbl = [&] { return rng() % 11 == 0 ? special[rng() % 8] : std::ldexp((double)(int64_t)(rng() >> 11) - 4.5e15, (int)(rng() % 40) - 60); };
    auto ign = [&] { return (int64_t)(rng() % 2001) - 1000; };
    if (null()) (void)fa.AppendNull(); else (void)fa.Append(dbl());
    if (null()) (void)fb.AppendNull(); else (void)fb.Append(dbl());
    if (null()) (void)ia.AppendNull(); else (void)ia.Append(rng() % 13 == 0 ? (int64_t)(rng()) : ign());
    if (null()) (void)ib.AppendNull(); else { int64_t v = ign(); (void)ib.Append(v == 0 ? 7 : v); }  // (a zero divisor is its own test)
    if (null()) (void)ba.AppendNull(); else (void)ba.Append((bool)(rng() & 1));
    if (null()) (void)bb.AppendNull(); else (void)bb.Append(rng() % 3 == 0);
    if (nulls && rng() % 50 == 0) (void)kb.AppendNull(); else (void)kb.Append((int64_t)(rng() % 97) * 1000003 - 17);
    if (null()) (void)xb.AppendNull(); else (void)xb.Append(n ? (int64_t)(rng() % (uint64_t)n) : 0);
  }
  auto sl = [&](auto& b) { return b.Finish().ValueOrDie()->Slice(off, n); };
  Inputs in;
  in.f64a = sl(fa); in.f64b = sl(fb); in.i64a = sl(ia); in.i64b = sl(ib); in.boola = sl(ba); in.boolb = sl(bb); in.keys = sl(kb); in.idx = sl(xb);
  return in;
}

int main() {
  if (!cp::Initialize().ok()) {
    std::fprintf(stderr, "arrow::compute::Initialize failed\n");
    return 2;
  }
  if (pdx_init(0) != PDX_OK) {
    std::fprintf(stderr, "pdx_init: %s\n", pdx_last_error());
    return 2;
  }
  const cp::ScalarAggregateOptions agg_opts(/*skip_nulls=*/true, /*min_count=*/1);
  for (int64_t n : {(int64_t)0, (int64_t)1, (int64_t)17, (int64_t)100000})
    for (int64_t off : {(int64_t)0, (int64_t)3, (int64_t)64})
      for (bool nulls : {false, true}) {
        const Inputs in = MakeInputs(n, off, nulls, 1000 + (uint64_t)n * 7 + (uint64_t)off * 3 + nulls);
        const std::string tag = " n=" + std::to_string(n) + " off=" + std::to_string(off) + " nulls=" + std::to_string(nulls);
        // ---- Series op Series / Scalar (src/series.cpp:19-33, 229-235; src/scalar.cpp:24-41)
        for (const char* f : {"add", "subtract", "multiply", "divide"}) {
          Compare(std::string(f) + " f64,f64" + tag, pd::pdx::CallFunction(f, {in.f64a, in.f64b}), cp::CallFunction(f, {in.f64a, in.f64b}));
          Compare(std::string(f) + " i64,i64" + tag, pd::pdx::CallFunction(f, {in.i64a, in.i64b}), cp::CallFunction(f, {in.i64a, in.i64b}));
          Compare(std::string(f) + " i64,f64" + tag, pd::pdx::CallFunction(f, {in.i64a, in.f64b}), cp::CallFunction(f, {in.i64a, in.f64b}));
          Compare(std::string(f) + " f64,scalar" + tag, pd::pdx::CallFunction(f, {in.f64a, arrow::Datum(2.5)}), cp::CallFunction(f, {in.f64a, arrow::Datum(2.5)}));
          Compare(std::string(f) + " scalar,i64" + tag, pd::pdx::CallFunction(f, {arrow::Datum((int64_t)2), in.i64b}), cp::CallFunction(f, {arrow::Datum((int64_t)2), in.i64b}));
        }
        if (n) Compare("divide by zero" + tag, pd::pdx::CallFunction("divide", {in.i64a, arrow::Datum((int64_t)0)}), cp::CallFunction("divide", {in.i64a, arrow::Datum((int64_t)0)}));
        // ---- comparisons, and / or / invert (src/series.cpp:247-261, 319)
        for (const char* f : {"equal", "not_equal", "less", "less_equal", "greater", "greater_equal"}) {
          Compare(std::string(f) + " f64" + tag, pd::pdx::CallFunction(f, {in.f64a, in.f64b}), cp::CallFunction(f, {in.f64a, in.f64b}));
          Compare(std::string(f) + " i64,scalar" + tag, pd::pdx::CallFunction(f, {in.i64a, arrow::Datum((int64_t)5)}), cp::CallFunction(f, {in.i64a, arrow::Datum((int64_t)5)}));
        }
        Compare("and" + tag, pd::pdx::CallFunction("and", {in.boola, in.boolb}), cp::CallFunction("and", {in.boola, in.boolb}));
        Compare("or" + tag, pd::pdx::CallFunction("or", {in.boola, in.boolb}), cp::CallFunction("or", {in.boola, in.boolb}));
        Compare("invert" + tag, pd::pdx::CallFunction("invert", {in.boola}), cp::CallFunction("invert", {in.boola}));
        // ---- whole-array aggregates (src/ndframe.cpp:26-31)
        for (const char* f : {"sum", "mean", "min", "max", "count", "min_max"}) {
          const cp::FunctionOptions* o = std::string(f) == "count" ? nullptr : &agg_opts;
          Compare(std::string(f) + " f64" + tag, pd::pdx::CallFunction(f, {in.f64a}, o), cp::CallFunction(f, {in.f64a}, o));
          Compare(std::string(f) + " i64" + tag, pd::pdx::CallFunction(f, {in.i64b}, o), cp::CallFunction(f, {in.i64b}, o));
        }
        // ---- filter (EMIT_NULL as DataFrame::where, DROP as the Series index) and take (src/dataframe.cpp:461-492)
        for (auto sel : {cp::FilterOptions::EMIT_NULL, cp::FilterOptions::DROP}) {
          const cp::FilterOptions fo(sel);
          Compare("filter f64" + tag, pd::pdx::CallFunction("filter", {in.f64a, in.boola}, &fo), cp::CallFunction("filter", {in.f64a, in.boola}, &fo));
          Compare("filter i64" + tag, pd::pdx::CallFunction("filter", {in.i64a, in.boolb}, &fo), cp::CallFunction("filter", {in.i64a, in.boolb}, &fo));
        }
        Compare("take f64" + tag, pd::pdx::CallFunction("take", {in.f64a, in.idx}), cp::CallFunction("take", {in.f64a, in.idx}));
        if (n) Compare("take out of bounds" + tag, pd::pdx::CallFunction("take", {in.i64a, arrow::Datum(arrow::MakeArrayFromScalar(arrow::Int64Scalar(n), 1).ValueOrDie())}),
                       cp::CallFunction("take", {in.i64a, arrow::Datum(arrow::MakeArrayFromScalar(arrow::Int64Scalar(n), 1).ValueOrDie())}));
        // ---- Grouper (src/dataframe.cpp:1571-1600) + per-group sum / mean / min / max / count (src/pd_core_macros.h:5-147).
        // Group ORDER: this backend numbers groups by first occurrence; Arrow's Grouper walks the batch in mini-batches of 128, 256,
        // 512, then 1024 rows and gives the keys first seen in a mini-batch one contiguous block of ids -- the block first-occurrence
        // numbering gives them -- permuted inside the block (include/pdx/abi.h at pdx_groupby_create).  Checked here: the two id
        // columns describe the same partition of the rows, every Arrow id lies in its mini-batch's block, and uniques / aggregates
        // agree group by group under that mapping.
        if (n) {
          auto key_batch = cp::ExecBatch::Make(std::vector<arrow::Datum>{in.keys}).ValueOrDie();
          auto ag = cp::Grouper::Make(key_batch.GetTypes()).ValueOrDie();
          arrow::Datum want_ids = ag->Consume(cp::ExecSpan(key_batch)).ValueOrDie();
          pd::pdx::Grouper pg;
          auto got_ids = pg.Consume(in.keys).ValueOrDie().make_array();
          const auto& ours = static_cast<const arrow::UInt32Array&>(*got_ids);
          const auto theirs_ptr = want_ids.array_as<arrow::UInt32Array>();
          const auto& theirs = *theirs_ptr;
          const int64_t G = (int64_t)ag->num_groups();
          ++g_checked;
          if ((int64_t)pg.num_groups() != G) Fail("Grouper::num_groups" + tag);
          std::vector<int64_t> to_arrow((size_t)G, -1), first_row((size_t)G, -1);
          bool same_partition = (int64_t)pg.num_groups() == G;
          for (int64_t i = 0; i < n && same_partition; ++i) {
            const uint32_t o = ours.Value(i), a = theirs.Value(i);
            if (o >= (uint32_t)G) { same_partition = false; break; }
            if (to_arrow[o] < 0) { to_arrow[o] = a; first_row[o] = i; }
            same_partition = to_arrow[o] == (int64_t)a;
          }
          std::vector<char> hit((size_t)G, 0);
          for (int64_t g = 0; g < G && same_partition; ++g) {
            same_partition = to_arrow[(size_t)g] >= 0 && !hit[(size_t)to_arrow[(size_t)g]] && (g == 0 || first_row[(size_t)g] > first_row[(size_t)g - 1]);
            if (same_partition) hit[(size_t)to_arrow[(size_t)g]] = 1;
          }
          ++g_checked;
          if (!same_partition) Fail("Grouper::Consume" + tag + ": not the same partition of the rows / ids not in first-occurrence order");
          else {
            auto minibatch = [](int64_t row) { return row < 128 ? 0 : row < 384 ? 1 : row < 896 ? 2 : 3 + (row - 896) / 1024; };
            int64_t g0 = 0;
            bool in_block = true;
            while (g0 < G) {  // ids [g0, g1): the groups first seen in one mini-batch
              int64_t g1 = g0;
              while (g1 < G && minibatch(first_row[(size_t)g1]) == minibatch(first_row[(size_t)g0])) ++g1;
              for (int64_t g = g0; g < g1; ++g) in_block = in_block && to_arrow[(size_t)g] >= g0 && to_arrow[(size_t)g] < g1;
              g0 = g1;
            }
            ++g_checked;
            if (!in_block) Fail("Grouper::Consume" + tag + ": an Arrow id lies outside its mini-batch's block of first-occurrence ids");
            auto want_u = ag->GetUniques().ValueOrDie().values[0].make_array();
            auto got_u = pg.GetUniques().ValueOrDie();
            for (int64_t g = 0; g < G; ++g) {
              ++g_checked;
              if (!SameScalar(*got_u->GetScalar(g).ValueOrDie(), *want_u->GetScalar(to_arrow[(size_t)g]).ValueOrDie())) Fail("Grouper::GetUniques" + tag + " group " + std::to_string(g));
            }
            auto groupings = cp::Grouper::MakeGroupings(theirs, ag->num_groups()).ValueOrDie();
            auto grouped = cp::Grouper::ApplyGroupings(*groupings, *in.f64a).ValueOrDie();
            const std::pair<const char*, int> kinds[] = {{"sum", PDX_AGG_SUM}, {"mean", PDX_AGG_MEAN}, {"min", PDX_AGG_MIN}, {"max", PDX_AGG_MAX}, {"count", PDX_AGG_COUNT}};
            for (auto& [f, kind] : kinds) {
              auto got = pg.Aggregate(in.f64a, kind).ValueOrDie();
              for (int64_t g = 0; g < G; ++g) {
                const cp::FunctionOptions* o = kind == PDX_AGG_COUNT ? nullptr : &agg_opts;
                auto want = cp::CallFunction(f, {grouped->value_slice(to_arrow[(size_t)g])}, o).ValueOrDie().scalar();
                auto have = got->GetScalar(g).ValueOrDie();
                ++g_checked;
                if (!SameScalar(*have, *want)) Fail(std::string("group ") + f + tag + " group " + std::to_string(g) + ": " + have->ToString() + " vs " + want->ToString());
              }
            }
          }
        }
      }
  std::printf("arrow_bridge_test: %d comparisons against Arrow C++ %s, %d mismatches\n", g_checked, ARROW_VERSION_STRING, g_failed);
  return g_failed ? 1 : 0;
}
