// pdx.hpp -- C++ host facade over the C ABI (include/pdx/abi.h): the same names, argument meaning and error behaviour as the
// reference's pd::Series / pd::DataFrame / pd::GroupBy / pd::Resampler / pd::concat for the hot path, with the data resident
// in HBM.  Header-only; link with -lpdx_hip.  Errors are std::runtime_error(pdx_last_error()), the reference's
// ReturnOrThrowOnFailure convention (src/core.h:181-194).
//
// Reference surface mirrored (file:line relative to the reference repository):
//   pd::Series      src/series.h:20-516   operators src/series.cpp:19-33,229-261; where/take 130-159; aggregations src/ndframe.cpp:119-220
//   pd::DataFrame   src/dataframe.h:75-709   BinaryFunction src/dataframe.cpp:233-275; where/take 461-492; group_by 1227-1235
//   pd::GroupBy     src/group_by.h:22-299    aggregations src/pd_core_macros.h:5-147
//   pd::Resampler   src/group_by.h:255-299   pd::resample src/resample.h:51-122
//   pd::concat      src/concat.h:56-64
// Differences kept deliberately small: values are constructed from host std::vector (like the reference) and uploaded once
// (pd::GPUSeries precedent, src/cudf/series.h:10-101); the default index is an implicit 0..n-1 range instead of a materialised
// uint64 array (src/ndframe.cpp:100-107); only int64 / double / bool / timestamp[ns] columns exist on this path.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <cstdio>
#include <functional>
#include <map>
#include <memory>
#include <array>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "pdx/abi.h"

namespace pd {

inline void ThrowOnFailure(int status) {
  if (status != PDX_OK) throw std::runtime_error(pdx_last_error());
}

// ---------------------------------------------------------------- device buffer / column (shared ownership, like arrow::Buffer)
struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  std::shared_ptr<void> keep;  // set: `ptr` aliases memory owned by `keep` (an IPC frame's device copy) and is not freed here
  explicit DeviceBuffer(size_t n) : bytes(n) { ThrowOnFailure(pdx_malloc(&ptr, n ? n : 1)); }
  DeviceBuffer(const void* borrowed, size_t n, std::shared_ptr<void> owner) : ptr(const_cast<void*>(borrowed)), bytes(n), keep(std::move(owner)) {}
  ~DeviceBuffer() {
    if (!keep) pdx_free(ptr);
  }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
};
using BufferPtr = std::shared_ptr<DeviceBuffer>;

inline size_t bitmap_bytes(int64_t n) { return (size_t)((n + 7) / 8 + 16); }

struct Array {  // arrow::ArrayData analogue in HBM
  int dtype = PDX_INT64;
  int64_t length = 0, offset = 0, null_count = 0;
  BufferPtr values, validity;

  pdx_column c() const {
    pdx_column col{};
    col.dtype = dtype;
    col.length = length;
    col.offset = offset;
    col.null_count = validity ? null_count : 0;
    col.validity = validity ? validity->ptr : nullptr;
    col.values = values ? values->ptr : nullptr;
    return col;
  }
  pdx_mut_column mut() {
    pdx_mut_column m{};
    m.dtype = dtype;
    m.length = length;
    m.null_count = -1;
    m.validity = validity ? validity->ptr : nullptr;
    m.values = values ? values->ptr : nullptr;
    return m;
  }
  bool has_nulls() const { return validity && null_count != 0; }

  static Array Empty(int dtype, int64_t n, bool with_validity) {
    Array a;
    a.dtype = dtype;
    a.length = n;
    a.values = std::make_shared<DeviceBuffer>(dtype == PDX_BOOL ? bitmap_bytes(n) : (size_t)(n > 0 ? n : 1) * 8);
    if (with_validity) {
      a.validity = std::make_shared<DeviceBuffer>(bitmap_bytes(n));
      a.null_count = -1;
    }
    return a;
  }
  static std::vector<uint8_t> PackBits(const std::vector<bool>& b) {
    std::vector<uint8_t> out(bitmap_bytes((int64_t)b.size()), 0);
    for (size_t i = 0; i < b.size(); ++i)
      if (b[i]) out[i >> 3] |= (uint8_t)(1u << (i & 7));
    return out;
  }
  template <typename T>
  static Array Make(const std::vector<T>& v, const std::vector<bool>* valid = nullptr) {
    Array a;
    const int64_t n = (int64_t)v.size();
    a.length = n;
    std::vector<bool> nan_valid;
    if constexpr (std::is_same_v<T, bool>) {
      a.dtype = PDX_BOOL;
      auto bits = PackBits(v);
      a.values = std::make_shared<DeviceBuffer>(bits.size());
      ThrowOnFailure(pdx_to_device(a.values->ptr, bits.data(), bits.size(), nullptr));
    } else {
      a.dtype = std::is_floating_point_v<T> ? PDX_FLOAT64 : PDX_INT64;
      using W = std::conditional_t<std::is_floating_point_v<T>, double, int64_t>;
      std::vector<W> wide(v.begin(), v.end());
      if constexpr (std::is_floating_point_v<T>) {
        // ArrayT<T>::Make: NaN becomes an Arrow null on construction (src/core.h:404-436)
        if (!valid) {
          bool any = false;
          nan_valid.resize(v.size(), true);
          for (size_t i = 0; i < v.size(); ++i)
            if (std::isnan(v[i])) { nan_valid[i] = false; any = true; }
          if (any) valid = &nan_valid;
        }
      }
      a.values = std::make_shared<DeviceBuffer>((size_t)(n > 0 ? n : 1) * 8);
      if (n) ThrowOnFailure(pdx_to_device(a.values->ptr, wide.data(), (size_t)n * 8, nullptr));
    }
    if (valid) {
      if ((int64_t)valid->size() != n) throw std::runtime_error("validity length differs from values length");
      auto bits = PackBits(*valid);
      a.validity = std::make_shared<DeviceBuffer>(bits.size());
      ThrowOnFailure(pdx_to_device(a.validity->ptr, bits.data(), bits.size(), nullptr));
      a.null_count = 0;
      for (bool b : *valid) a.null_count += !b;
    }
    return a;
  }
  template <typename T>
  std::vector<T> values_as() const {
    std::vector<T> out((size_t)length);
    if (dtype == PDX_BOOL) {
      std::vector<uint8_t> bits(bitmap_bytes(length + offset));
      ThrowOnFailure(pdx_to_host(bits.data(), values->ptr, (size_t)((length + offset + 7) / 8), nullptr));
      for (int64_t i = 0; i < length; ++i) out[(size_t)i] = (T)((bits[(size_t)((i + offset) >> 3)] >> ((i + offset) & 7)) & 1);
      return out;
    }
    if (dtype == PDX_FLOAT64) {
      std::vector<double> raw((size_t)length);
      if (length) ThrowOnFailure(pdx_to_host(raw.data(), static_cast<const double*>(values->ptr) + offset, (size_t)length * 8, nullptr));
      for (int64_t i = 0; i < length; ++i) out[(size_t)i] = (T)raw[(size_t)i];
    } else {
      std::vector<int64_t> raw((size_t)length);
      if (length) ThrowOnFailure(pdx_to_host(raw.data(), static_cast<const int64_t*>(values->ptr) + offset, (size_t)length * 8, nullptr));
      for (int64_t i = 0; i < length; ++i) out[(size_t)i] = (T)raw[(size_t)i];
    }
    return out;
  }
  std::vector<bool> valid_flags() const {
    std::vector<bool> out((size_t)length, true);
    if (!validity) return out;
    std::vector<uint8_t> bits(bitmap_bytes(length + offset));
    ThrowOnFailure(pdx_to_host(bits.data(), validity->ptr, (size_t)((length + offset + 7) / 8), nullptr));
    for (int64_t i = 0; i < length; ++i) out[(size_t)i] = (bits[(size_t)((i + offset) >> 3)] >> ((i + offset) & 7)) & 1;
    return out;
  }
};

// ---------------------------------------------------------------- pd::Scalar (src/scalar.h:62-241)
struct Scalar {
  pdx_scalar s{};
  Scalar() = default;
  explicit Scalar(pdx_scalar v) : s(v) {}
  Scalar(int64_t v) { s.dtype = PDX_INT64; s.is_valid = 1; s.v.i64 = v; }
  Scalar(int v) : Scalar((int64_t)v) {}
  Scalar(double v) { s.dtype = PDX_FLOAT64; s.is_valid = 1; s.v.f64 = v; }
  Scalar(bool v) { s.dtype = PDX_BOOL; s.is_valid = 1; s.v.i64 = v ? 1 : 0; }
  bool isValid() const { return s.is_valid != 0; }
  template <typename T>
  T as() const {
    if (!s.is_valid) {
      if constexpr (std::is_floating_point_v<T>) return std::nan("");  // null -> NaN for floats (src/scalar.h:105-150)
      throw std::runtime_error("scalar is null");
    }
    return s.dtype == PDX_FLOAT64 ? (T)s.v.f64 : (T)s.v.i64;
  }
  bool operator==(double x) const { return s.is_valid && (s.dtype == PDX_FLOAT64 ? s.v.f64 == x : (double)s.v.i64 == x); }
  Array to_array() const {
    std::vector<bool> valid{isValid()};
    if (s.dtype == PDX_BOOL) return Array::Make(std::vector<bool>{s.v.i64 != 0}, &valid);
    return s.dtype == PDX_FLOAT64 ? Array::Make(std::vector<double>{s.v.f64}, &valid) : Array::Make(std::vector<int64_t>{s.v.i64}, &valid);
  }
};

class DataFrame;
struct GroupBy;
struct Resampler;

// ---------------------------------------------------------------- pd::Series
class Series {
 public:
  Array m_array;
  std::optional<Array> m_index;  // nullopt = implicit 0..n-1 range
  std::string m_name;
  bool m_is_index = false;

  Series() = default;
  Series(Array a, std::optional<Array> index = std::nullopt, std::string name = "", bool is_index = false)
      : m_array(std::move(a)), m_index(std::move(index)), m_name(std::move(name)), m_is_index(is_index) {}
  template <typename T>
  explicit Series(const std::vector<T>& v, std::string name = "") : m_array(Array::Make(v)), m_name(std::move(name)) {}
  template <typename T>
  Series(const std::vector<T>& v, const std::vector<bool>& valid, std::string name = "") : m_array(Array::Make(v, &valid)), m_name(std::move(name)) {}

  int64_t size() const { return m_array.length; }
  int dtype() const { return m_array.dtype; }
  const std::string& name() const { return m_name; }
  template <typename T>
  std::vector<T> values() const { return m_array.values_as<T>(); }
  Scalar at(int64_t i) const {
    if (i < 0 || i >= size()) throw std::runtime_error("index out of range");
    Array one = m_array;
    one.offset += i;
    one.length = 1;
    pdx_scalar s{};
    auto c = one.c();
    // a length-1 min is the element itself (null-aware)
    ThrowOnFailure(pdx_aggregate(m_array.dtype == PDX_BOOL ? PDX_AGG_COUNT : PDX_AGG_MIN, &c, &s, nullptr));
    return Scalar(s);
  }

  // ---- Series::broadcast / reindex (src/series.cpp:212-227, 1255-1309): operands with different EXPLICIT indexes are aligned
  // on the sorted union of their labels; labels missing on one side give null rows
  bool same_index(const Series& o) const {
    if (!m_index || !o.m_index) return o.size() == size();  // implicit ranges: positional
    if (m_index->length != o.m_index->length || m_index->dtype != o.m_index->dtype) return false;
    if (m_index->length == 0) return true;
    Array a = *m_index, b = *o.m_index;
    a.dtype = b.dtype = PDX_INT64;  // labels compare as 64-bit patterns
    Array eq = run_compare(PDX_EQ, a, b, false);
    auto cm = eq.c();
    int64_t m = 0;
    ThrowOnFailure(pdx_filter_count(&cm, /*emit_null=*/0, &m, nullptr));
    return m == m_index->length;
  }
  // fillValue (src/series.cpp:1295-1302: `fillValue ? AppendScalar(*fillValue->scalar) : AppendNull()`): labels the old index lacks
  // get the scalar instead of null -- one pdx_if_else on the take indices' validity bitmap (cond = label present), so a present
  // label whose value is null stays null.  The scalar's type must be the column's (Arrow's AppendScalar check).
  static Array fill_absent(const Array& taken, const Array& idx, const std::optional<Scalar>& fillValue) {
    if (!fillValue || !fillValue->isValid() || !idx.validity) return taken;
    static const char* names[] = {"int64", "double", "bool", "uint64", "timestamp[ns]"};
    if (taken.dtype == PDX_BOOL || fillValue->s.dtype == PDX_BOOL || (taken.dtype == PDX_FLOAT64) != (fillValue->s.dtype == PDX_FLOAT64))
      throw std::runtime_error(std::string("Cannot append scalar of type ") + names[fillValue->s.dtype] + " to builder for type " + names[taken.dtype]);
    Array present;
    present.dtype = PDX_BOOL;
    present.length = idx.length;
    present.offset = idx.offset;
    present.values = idx.validity;
    Array view = taken;
    if (view.dtype != PDX_FLOAT64) view.dtype = PDX_INT64;  // uint64 / timestamp labels move as 64-bit patterns
    Array out = run_if_else(present, view, fillValue->to_array(), PDX_SCALAR_RHS);
    out.dtype = taken.dtype;
    return out;
  }
  static Array reindex_plan(const Array& oldIndex, const Array& newIndex) {
    if (newIndex.dtype != oldIndex.dtype) throw std::runtime_error("type(NewIndex) != type(CurrentIndex).");
    Array idx = Array::Empty(PDX_INT64, newIndex.length, true);
    auto co = oldIndex.c(), cn = newIndex.c();
    auto mi = idx.mut();
    ThrowOnFailure(pdx_reindex_indices(&co, &cn, &mi, nullptr));
    idx.null_count = mi.null_count;
    return idx;
  }
  Series reindex(const Array& newIndex, const std::optional<Scalar>& fillValue = std::nullopt) const {
    if (!m_index) throw std::runtime_error("reindex needs an explicit index");
    Array idx = reindex_plan(*m_index, newIndex);
    return Series(fill_absent(run_take({m_array}, idx)[0], idx, fillValue), newIndex, m_name);
  }
  std::array<Series, 2> broadcast(const Series& o) const {
    if (same_index(o)) return {*this, o};
    if (m_index->dtype != o.m_index->dtype) throw std::runtime_error("type(NewIndex) != type(CurrentIndex).");
    Array u = Array::Empty(m_index->dtype, m_index->length + o.m_index->length, false);
    auto ca = m_index->c(), cb = o.m_index->c();
    auto mu = u.mut();
    ThrowOnFailure(pdx_index_union(&ca, &cb, /*sort=*/1, &mu, nullptr));
    u.length = mu.length;
    return {reindex(u), o.reindex(u)};
  }

  // ---- BINARY_OPERATOR (src/series.cpp:19-33): `auto [x, y] = broadcast(a)` then one kernel over equal-length operands
  Series binary(int op, const Series& o) const {
    if (m_index && o.m_index && !same_index(o)) {
      auto xy = broadcast(o);
      return xy[0].wrap(run_binary(op, xy[0].m_array, xy[1].m_array, false));
    }
    if (o.size() != size()) throw std::runtime_error("Array arguments must all be the same length");
    return wrap(run_binary(op, m_array, o.m_array, false));
  }
  Series binary(int op, const Scalar& o) const { return wrap(run_binary(op, m_array, o.to_array(), true)); }
  Series operator+(const Series& o) const { return binary(PDX_ADD, o); }
  Series operator-(const Series& o) const { return binary(PDX_SUB, o); }
  Series operator*(const Series& o) const { return binary(PDX_MUL, o); }
  Series operator/(const Series& o) const { return binary(PDX_DIV, o); }
  Series operator+(const Scalar& o) const { return binary(PDX_ADD, o); }
  Series operator-(const Scalar& o) const { return binary(PDX_SUB, o); }
  Series operator*(const Scalar& o) const { return binary(PDX_MUL, o); }
  Series operator/(const Scalar& o) const { return binary(PDX_DIV, o); }
  // Series::if_else / where(cond, other) (src/series.cpp:1203-1209, 1247-1253): cond ? *this : other
  Series if_else(const Series& cond, const Series& other) const {
    if (other.size() != size() || cond.size() != size()) throw std::runtime_error("Array arguments must all be the same length");
    return wrap(run_if_else(cond.m_array, m_array, other.m_array, PDX_SCALAR_NONE));
  }
  Series if_else(const Series& cond, const Scalar& other) const {
    if (cond.size() != size()) throw std::runtime_error("Array arguments must all be the same length");
    return wrap(run_if_else(cond.m_array, m_array, other.to_array(), PDX_SCALAR_RHS));
  }
  Series where(const Series& cond, const Series& other) const { return if_else(cond, other); }
  Series where(const Series& cond, const Scalar& other) const { return if_else(cond, other); }
  // BINARY_OPERATOR(| & ^ << >>) (src/series.cpp:237-245): bit_wise_or / and / xor, shift_left / shift_right on integers
  Series operator|(const Series& o) const { return binary(PDX_BIT_OR, o); }
  Series operator&(const Series& o) const { return binary(PDX_BIT_AND, o); }
  Series operator^(const Series& o) const { return binary(PDX_BIT_XOR, o); }
  Series operator<<(const Series& o) const { return binary(PDX_SHIFT_LEFT, o); }
  Series operator>>(const Series& o) const { return binary(PDX_SHIFT_RIGHT, o); }
  Series operator|(const Scalar& o) const { return binary(PDX_BIT_OR, o); }
  Series operator&(const Scalar& o) const { return binary(PDX_BIT_AND, o); }
  Series operator^(const Scalar& o) const { return binary(PDX_BIT_XOR, o); }
  Series operator<<(const Scalar& o) const { return binary(PDX_SHIFT_LEFT, o); }
  Series operator>>(const Scalar& o) const { return binary(PDX_SHIFT_RIGHT, o); }
  // ---- functions of one column: Series::abs / exp / pow / sign / sqrt (src/series.h:89-109); operator- = CallFunction("negate")
  Series operator-() const { return wrap(run_unary(PDX_NEGATE, m_array)); }
  Series abs() const { return wrap(run_unary(PDX_ABS, m_array)); }
  Series sign() const { return wrap(run_unary(PDX_SIGN, m_array)); }
  Series sqrt() const { return wrap(run_unary(PDX_SQRT, m_array)); }
  Series exp() const { return wrap(run_unary(PDX_EXP, m_array)); }
  Series pow(double x) const { return wrap(run_power(m_array, x)); }

  // ---- comparisons (src/series.cpp:247-257) and logical (259-261, 319)
  Series compare(int op, const Series& o) const {
    if (o.size() != size()) throw std::runtime_error("Array arguments must all be the same length");
    return wrap(run_compare(op, m_array, o.m_array, false));
  }
  Series compare(int op, const Scalar& o) const { return wrap(run_compare(op, m_array, o.to_array(), true)); }
  Series operator<(const Series& o) const { return compare(PDX_LT, o); }
  Series operator<=(const Series& o) const { return compare(PDX_LE, o); }
  Series operator>(const Series& o) const { return compare(PDX_GT, o); }
  Series operator>=(const Series& o) const { return compare(PDX_GE, o); }
  Series operator==(const Series& o) const { return compare(PDX_EQ, o); }
  Series operator!=(const Series& o) const { return compare(PDX_NE, o); }
  Series operator<(const Scalar& o) const { return compare(PDX_LT, o); }
  Series operator<=(const Scalar& o) const { return compare(PDX_LE, o); }
  Series operator>(const Scalar& o) const { return compare(PDX_GT, o); }
  Series operator>=(const Scalar& o) const { return compare(PDX_GE, o); }
  Series operator==(const Scalar& o) const { return compare(PDX_EQ, o); }
  Series operator!=(const Scalar& o) const { return compare(PDX_NE, o); }
  Series operator&&(const Series& o) const { return wrap(run_logical(PDX_AND, m_array, o.m_array)); }
  Series operator||(const Series& o) const { return wrap(run_logical(PDX_OR, m_array, o.m_array)); }
  Series operator!() const {
    Array out = Array::Empty(PDX_BOOL, size(), m_array.has_nulls());
    auto a = m_array.c();
    auto m = out.mut();
    ThrowOnFailure(pdx_invert(&a, &m, nullptr));
    out.null_count = m.null_count;
    return wrap(std::move(out));
  }

  // ---- NDFrame aggregations (src/ndframe.cpp:119-220)
  Scalar agg(int kind) const {
    pdx_scalar s{};
    auto c = m_array.c();
    ThrowOnFailure(pdx_aggregate(kind, &c, &s, nullptr));
    return Scalar(s);
  }
  Scalar sum() const { return agg(PDX_AGG_SUM); }
  Scalar mean() const { return agg(PDX_AGG_MEAN); }
  Scalar min() const { return agg(PDX_AGG_MIN); }
  Scalar max() const { return agg(PDX_AGG_MAX); }
  Scalar count() const { return agg(PDX_AGG_COUNT); }
  // NDFrame::count_na / all / any / nunique (src/ndframe.cpp:110-127), Series::unique (src/series.h:380)
  int64_t count_na() const { return size() - count().as<int64_t>(); }
  bool all() const { return bool_counts("all").second == 0; }
  bool any() const { return bool_counts("any").first > 0; }
  Series unique() const {
    if (m_array.dtype == PDX_FLOAT64 || m_array.dtype == PDX_BOOL) throw std::runtime_error("unique: integer-like columns only through this facade");
    auto ck = m_array.c();
    pdx_groupby* h = nullptr;
    ThrowOnFailure(pdx_groupby_create(&ck, nullptr, &h));
    const int64_t G = pdx_groupby_num_groups(h);
    Array u = Array::Empty(m_array.dtype, G, true);
    auto mu = u.mut();
    const int rc = pdx_groupby_unique_keys(h, &mu, nullptr);
    pdx_groupby_destroy(h);
    ThrowOnFailure(rc);
    u.null_count = mu.null_count;
    return Series(u, std::nullopt, m_name);
  }
  int64_t nunique() const { return unique().count().as<int64_t>(); }

 private:
  // (valid AND true, valid AND false) of a boolean Series; Arrow has no all / any kernel for other types, and min_count = 1
  std::pair<int64_t, int64_t> bool_counts(const char* what) const {
    if (m_array.dtype != PDX_BOOL) throw std::runtime_error(std::string("Function '") + what + "' has no kernel matching input types");
    int64_t t = 0, f = 0;
    auto cm = m_array.c();
    ThrowOnFailure(pdx_filter_count(&cm, /*emit_null=*/0, &t, nullptr));
    Series inv = !(*this);
    auto ci = inv.m_array.c();
    ThrowOnFailure(pdx_filter_count(&ci, /*emit_null=*/0, &f, nullptr));
    if (t + f == 0) throw std::runtime_error(std::string(what) + "() of a Series without a valid value is null");
    return {t, f};
  }

 public:
  // ---- where / take / operator[] (src/series.cpp:130-159, src/ndframe.cpp:347-350)
  Series where(const Series& mask) const {
    if (m_is_index) throw std::runtime_error("where() is not supported on an index Series");
    if (mask.dtype() != PDX_BOOL) throw std::runtime_error("filter mask must be boolean");
    auto outs = run_filter(columns_with_index(), mask.m_array);
    return Series(outs[0], m_index ? std::optional<Array>(outs[1]) : std::nullopt, m_name);
  }
  Series take(const Series& idx) const {
    if (idx.dtype() == PDX_BOOL) throw std::runtime_error("take indices must be integers, not boolean");
    auto outs = run_take(columns_with_index(), idx.m_array);
    return Series(outs[0], m_index ? std::optional<Array>(outs[1]) : std::nullopt, m_name);
  }
  Series operator[](const Series& s) const { return s.dtype() == PDX_BOOL ? where(s) : take(s); }

  // ---- sort (src/series.cpp:864-868, 978-992, 1211-1229): CallFunction("array_sort_indices") + Take of values and index + Slice
  Series argsort(bool ascending = true) const {
    Array idx = Array::Empty(PDX_UINT64, size(), false);
    auto ca = m_array.c();
    auto mi = idx.mut();
    ThrowOnFailure(pdx_argsort(&ca, ascending ? 1 : 0, &mi, nullptr));
    return wrap(idx);
  }
  Series sort(bool ascending = true) const {
    Array idx = argsort(ascending).m_array;
    auto outs = run_take(columns_with_index(), idx);
    // (the implicit 0..n-1 index taken by the sort indices is the sort indices)
    return Series(outs[0], m_index ? outs[1] : idx, m_name);
  }
  Series n_largest(int n) const { return sort(false).head(n); }
  Series n_smallest(int n) const { return sort(true).head(n); }
  Series head(int64_t n) const {  // array()->Slice(0, n): zero copy
    if (size() <= n) return *this;
    Series s = *this;
    s.m_array.length = n;
    if (s.m_array.validity) s.m_array.null_count = -1;
    if (s.m_index) s.m_index->length = n;
    return s;
  }

  inline Resampler resample(const std::string& rule, bool closed_right = false, bool label_right = false) const;

  // ---- shared kernels-through-ABI helpers (also used by DataFrame)
  static Array run_binary(int op, const Array& a, const Array& b, bool scalar) {
    const bool is_f = a.dtype == PDX_FLOAT64 || b.dtype == PDX_FLOAT64;
    Array out = Array::Empty(is_f ? PDX_FLOAT64 : PDX_INT64, a.length, a.has_nulls() || b.has_nulls());
    auto ca = a.c(), cb = b.c();
    auto m = out.mut();
    ThrowOnFailure(pdx_binary(op, &ca, &cb, scalar, &m, nullptr));
    out.null_count = m.null_count;
    return out;
  }
  // arrow::compute::Cast(column, float64()): checked = the default safe cast (pd::concat, src/concat.cpp:127), unchecked = static_cast per
  // value (Arrow's mean over int64 input)
  static Array run_cast_f64(const Array& a, bool checked) {
    Array out = Array::Empty(PDX_FLOAT64, a.length, a.has_nulls());
    auto ca = a.c();
    auto m = out.mut();
    ThrowOnFailure(pdx_cast_f64(&ca, checked ? 1 : 0, &m, nullptr));
    out.null_count = m.null_count;
    return out;
  }
  static Array run_if_else(const Array& cond, const Array& a, const Array& b, int side) {
    const bool is_f = a.dtype == PDX_FLOAT64 || b.dtype == PDX_FLOAT64;
    Array out = Array::Empty(is_f ? PDX_FLOAT64 : PDX_INT64, cond.length, cond.has_nulls() || a.has_nulls() || b.has_nulls());
    auto cc = cond.c(), ca = a.c(), cb = b.c();
    auto m = out.mut();
    ThrowOnFailure(pdx_if_else(&cc, &ca, &cb, side, &m, nullptr));
    out.null_count = m.null_count;
    return out;
  }
  static Array run_unary(int op, const Array& a) {
    const int out_dt = (op == PDX_SQRT || op == PDX_EXP) ? PDX_FLOAT64 : (op == PDX_SIGN && a.dtype != PDX_FLOAT64) ? PDX_INT64 : a.dtype;
    Array out = Array::Empty(out_dt, a.length, a.has_nulls());
    auto ca = a.c();
    auto m = out.mut();
    ThrowOnFailure(pdx_unary(op, &ca, &m, nullptr));
    out.null_count = m.null_count;
    return out;
  }
  static Array run_power(const Array& a, double x) {
    Array out = Array::Empty(PDX_FLOAT64, a.length, a.has_nulls());
    auto ca = a.c();
    auto m = out.mut();
    ThrowOnFailure(pdx_power(&ca, x, &m, nullptr));
    out.null_count = m.null_count;
    return out;
  }
  static Array run_compare(int op, const Array& a, const Array& b, bool scalar) {
    Array out = Array::Empty(PDX_BOOL, a.length, a.has_nulls() || b.has_nulls());
    auto ca = a.c(), cb = b.c();
    auto m = out.mut();
    ThrowOnFailure(pdx_compare(op, &ca, &cb, scalar, &m, nullptr));
    out.null_count = m.null_count;
    return out;
  }
  static Array run_logical(int op, const Array& a, const Array& b) {
    Array out = Array::Empty(PDX_BOOL, a.length, a.has_nulls() || b.has_nulls());
    auto ca = a.c(), cb = b.c();
    auto m = out.mut();
    ThrowOnFailure(pdx_logical(op, &ca, &cb, &m, nullptr));
    out.null_count = m.null_count;
    return out;
  }
  static std::vector<Array> run_filter(const std::vector<Array>& cols, const Array& mask) {
    std::vector<pdx_column> in;
    for (auto& c : cols) in.push_back(c.c());
    auto cm = mask.c();
    int64_t m = 0;
    if (!cols.empty() && cols[0].length != mask.length)
      throw std::runtime_error("Filter inputs must all be the same length");
    ThrowOnFailure(pdx_filter_count(&cm, /*emit_null=*/1, &m, nullptr));
    std::vector<Array> outs;
    std::vector<pdx_mut_column> mo;
    for (auto& c : cols) {
      outs.push_back(Array::Empty(c.dtype, m, c.has_nulls() || mask.has_nulls()));
      mo.push_back(outs.back().mut());
    }
    ThrowOnFailure(pdx_filter(in.data(), (int)in.size(), &cm, 1, mo.data(), nullptr));
    for (size_t i = 0; i < outs.size(); ++i) outs[i].null_count = mo[i].null_count;
    return outs;
  }
  static std::vector<Array> run_take(const std::vector<Array>& cols, const Array& idx) {
    std::vector<pdx_column> in;
    for (auto& c : cols) in.push_back(c.c());
    auto ci = idx.c();
    std::vector<Array> outs;
    std::vector<pdx_mut_column> mo;
    for (auto& c : cols) {
      outs.push_back(Array::Empty(c.dtype, idx.length, c.has_nulls() || idx.has_nulls()));
      mo.push_back(outs.back().mut());
    }
    ThrowOnFailure(pdx_take(in.data(), (int)in.size(), &ci, mo.data(), nullptr));
    for (size_t i = 0; i < outs.size(); ++i) outs[i].null_count = mo[i].null_count;
    return outs;
  }

 private:
  // ReturnSeriesOrThrowOnError (src/series.cpp:1364-1384): equal length -> same index; the result name is reset to ""
  Series wrap(Array a) const { return Series(std::move(a), m_index, ""); }
  std::vector<Array> columns_with_index() const {
    std::vector<Array> cols{m_array};
    if (m_index) cols.push_back(*m_index);
    return cols;
  }
};
// BINARY_OPERATOR_2 (src/scalar.cpp:12-56): Scalar op Series = CallFunction(name, {scalar, s.array()}) with the Series' index --
// the scalar stays the LEFT operand (2 - s, 2 / s), pdx_binary / pdx_compare with PDX_SCALAR_LHS
inline Series scalar_lhs(int op, const Scalar& a, const Series& b, bool cmp) {
  Array sa = a.to_array();
  Array out = cmp ? Array::Empty(PDX_BOOL, b.size(), b.m_array.has_nulls() || sa.has_nulls())
                  : Array::Empty((sa.dtype == PDX_FLOAT64 || b.m_array.dtype == PDX_FLOAT64) ? PDX_FLOAT64 : PDX_INT64, b.size(),
                                 b.m_array.has_nulls() || sa.has_nulls());
  auto ca = sa.c(), cb = b.m_array.c();
  auto m = out.mut();
  ThrowOnFailure(cmp ? pdx_compare(op, &ca, &cb, PDX_SCALAR_LHS, &m, nullptr) : pdx_binary(op, &ca, &cb, PDX_SCALAR_LHS, &m, nullptr));
  out.null_count = m.null_count;
  return Series(std::move(out), b.m_index, "");
}
inline Series operator+(const Scalar& a, const Series& b) { return scalar_lhs(PDX_ADD, a, b, false); }
inline Series operator-(const Scalar& a, const Series& b) { return scalar_lhs(PDX_SUB, a, b, false); }
inline Series operator*(const Scalar& a, const Series& b) { return scalar_lhs(PDX_MUL, a, b, false); }
inline Series operator/(const Scalar& a, const Series& b) { return scalar_lhs(PDX_DIV, a, b, false); }
inline Series operator<(const Scalar& a, const Series& b) { return scalar_lhs(PDX_LT, a, b, true); }
inline Series operator<=(const Scalar& a, const Series& b) { return scalar_lhs(PDX_LE, a, b, true); }
inline Series operator>(const Scalar& a, const Series& b) { return scalar_lhs(PDX_GT, a, b, true); }
inline Series operator>=(const Scalar& a, const Series& b) { return scalar_lhs(PDX_GE, a, b, true); }
inline Series operator==(const Scalar& a, const Series& b) { return scalar_lhs(PDX_EQ, a, b, true); }
inline Series operator!=(const Scalar& a, const Series& b) { return scalar_lhs(PDX_NE, a, b, true); }

// ---------------------------------------------------------------- handle shared by GroupBy / Resampler
struct GroupHandle {
  pdx_groupby* h = nullptr;
  explicit GroupHandle(pdx_groupby* p) : h(p) {}
  ~GroupHandle() { pdx_groupby_destroy(h); }
  GroupHandle(const GroupHandle&) = delete;
  GroupHandle& operator=(const GroupHandle&) = delete;
};

// ---------------------------------------------------------------- pd::DataFrame
class DataFrame {
 public:
  std::vector<std::string> m_names;
  std::vector<Array> m_columns;
  std::optional<Array> m_index;

  DataFrame() = default;
  template <typename T>
  explicit DataFrame(const std::map<std::string, std::vector<T>>& cols) {
    for (auto& kv : cols) {
      m_names.push_back(kv.first);
      m_columns.push_back(Array::Make(kv.second));
    }
    check();
  }
  DataFrame(std::vector<std::string> names, std::vector<Array> cols, std::optional<Array> index = std::nullopt)
      : m_names(std::move(names)), m_columns(std::move(cols)), m_index(std::move(index)) { check(); }

  int64_t num_rows() const { return m_columns.empty() ? 0 : m_columns[0].length; }
  int64_t num_columns() const { return (int64_t)m_columns.size(); }
  int column_index(const std::string& name) const {
    for (size_t i = 0; i < m_names.size(); ++i)
      if (m_names[i] == name) return (int)i;
    throw std::runtime_error("no column named " + name);
  }
  Series operator[](const std::string& name) const { return Series(m_columns[(size_t)column_index(name)], m_index, name); }
  DataFrame operator[](const Series& s) const { return s.dtype() == PDX_BOOL ? where(s) : take(s); }

  // BinaryFunction (src/dataframe.cpp:233-275): the same kernel over every column
  DataFrame binary(int op, const DataFrame& o) const {
    if (o.num_rows() != num_rows() || o.num_columns() != num_columns()) throw std::runtime_error("DataFrame shapes differ");
    std::vector<Array> out;
    for (size_t i = 0; i < m_columns.size(); ++i) out.push_back(Series::run_binary(op, m_columns[i], o.m_columns[i], false));
    return DataFrame(m_names, out, m_index);
  }
  DataFrame binary(int op, const Series& o) const {
    if (o.size() != num_rows()) throw std::runtime_error("Array arguments must all be the same length");
    std::vector<Array> out;
    for (auto& c : m_columns) out.push_back(Series::run_binary(op, c, o.m_array, false));
    return DataFrame(m_names, out, m_index);
  }
  DataFrame binary(int op, const Scalar& o) const {
    std::vector<Array> out;
    Array s = o.to_array();
    for (auto& c : m_columns) out.push_back(Series::run_binary(op, c, s, true));
    return DataFrame(m_names, out, m_index);
  }
  // DataFrame::unary("negate" | "bit_wise_not"), UNARY_FUNCTION(abs | exp | sign | sqrt), pow (src/dataframe.cpp:251-275, 919-935)
  DataFrame unary(int op) const {
    std::vector<Array> out;
    for (auto& c : m_columns) out.push_back(Series::run_unary(op, c));
    return DataFrame(m_names, out, m_index);
  }
  DataFrame operator-() const { return unary(PDX_NEGATE); }
  DataFrame operator~() const { return unary(PDX_BIT_NOT); }
  DataFrame abs() const { return unary(PDX_ABS); }
  DataFrame sign() const { return unary(PDX_SIGN); }
  DataFrame sqrt() const { return unary(PDX_SQRT); }
  DataFrame exp() const { return unary(PDX_EXP); }
  DataFrame pow(double x) const {
    std::vector<Array> out;
    for (auto& c : m_columns) out.push_back(Series::run_power(c, x));
    return DataFrame(m_names, out, m_index);
  }
  // BINARY_OPERATOR_DF(> >= < <= == !=) (src/dataframe.cpp:563-573; DataFrame / Series / Scalar right-hand sides declared at
  // src/dataframe.h:476-520): the compare kernel over every column -> a frame of bit-packed boolean columns
  DataFrame compare(int op, const DataFrame& o) const {
    if (o.num_rows() != num_rows() || o.num_columns() != num_columns()) throw std::runtime_error("DataFrame shapes differ");
    std::vector<Array> out;
    for (size_t i = 0; i < m_columns.size(); ++i) out.push_back(Series::run_compare(op, m_columns[i], o.m_columns[i], false));
    return DataFrame(m_names, out, m_index);
  }
  DataFrame compare(int op, const Series& o) const {
    if (o.size() != num_rows()) throw std::runtime_error("Array arguments must all be the same length");
    std::vector<Array> out;
    for (auto& c : m_columns) out.push_back(Series::run_compare(op, c, o.m_array, false));
    return DataFrame(m_names, out, m_index);
  }
  DataFrame compare(int op, const Scalar& o) const {
    std::vector<Array> out;
    Array s = o.to_array();
    for (auto& c : m_columns) out.push_back(Series::run_compare(op, c, s, true));
    return DataFrame(m_names, out, m_index);
  }
  template <typename R> DataFrame operator>(const R& o) const { return compare(PDX_GT, o); }
  template <typename R> DataFrame operator>=(const R& o) const { return compare(PDX_GE, o); }
  template <typename R> DataFrame operator<(const R& o) const { return compare(PDX_LT, o); }
  template <typename R> DataFrame operator<=(const R& o) const { return compare(PDX_LE, o); }
  template <typename R> DataFrame operator==(const R& o) const { return compare(PDX_EQ, o); }
  template <typename R> DataFrame operator!=(const R& o) const { return compare(PDX_NE, o); }
  // BINARY_OPERATOR_DF(&&, and) / (||, or) (src/dataframe.cpp:575-577): Arrow's non-Kleene "and" / "or" over boolean frames; a
  // Scalar is broadcast (a null scalar makes every row null)
  DataFrame logical(int op, const DataFrame& o) const {
    if (o.num_rows() != num_rows() || o.num_columns() != num_columns()) throw std::runtime_error("DataFrame shapes differ");
    std::vector<Array> out;
    for (size_t i = 0; i < m_columns.size(); ++i) out.push_back(Series::run_logical(op, m_columns[i], o.m_columns[i]));
    return DataFrame(m_names, out, m_index);
  }
  DataFrame logical(int op, const Series& o) const {
    if (o.size() != num_rows()) throw std::runtime_error("Array arguments must all be the same length");
    std::vector<Array> out;
    for (auto& c : m_columns) out.push_back(Series::run_logical(op, c, o.m_array));
    return DataFrame(m_names, out, m_index);
  }
  DataFrame logical(int op, const Scalar& o) const {
    if (o.s.dtype != PDX_BOOL) throw std::runtime_error("Function 'and' / 'or' has no kernel matching input types (boolean expected)");
    const std::vector<bool> ok((size_t)num_rows(), o.isValid());
    Array b = Array::Make(std::vector<bool>((size_t)num_rows(), o.s.v.i64 != 0), o.isValid() ? nullptr : &ok);
    std::vector<Array> out;
    for (auto& c : m_columns) out.push_back(Series::run_logical(op, c, b));
    return DataFrame(m_names, out, m_index);
  }
  template <typename R> DataFrame operator&&(const R& o) const { return logical(PDX_AND, o); }
  template <typename R> DataFrame operator||(const R& o) const { return logical(PDX_OR, o); }
  // DataFrame::reindex / reindexAsync (src/dataframe.cpp:1139-1186, src/dataframe.h:403-406): ONE take plan for every column
  DataFrame reindex(const Array& newIndex, const std::optional<Scalar>& fillValue = std::nullopt) const {
    if (!m_index) throw std::runtime_error("reindex needs an explicit index");
    Array idx = Series::reindex_plan(*m_index, newIndex);
    auto taken = Series::run_take(m_columns, idx);
    std::vector<Array> out;
    for (auto& t : taken) out.push_back(Series::fill_absent(t, idx, fillValue));
    return DataFrame(m_names, out, newIndex);
  }
  DataFrame reindexAsync(const Array& newIndex, const std::optional<Scalar>& fillValue = std::nullopt) const { return reindex(newIndex, fillValue); }
  template <typename R> DataFrame operator|(const R& o) const { return binary(PDX_BIT_OR, o); }   // BINARY_OPERATOR_DF, src/dataframe.cpp:553-561
  template <typename R> DataFrame operator&(const R& o) const { return binary(PDX_BIT_AND, o); }
  template <typename R> DataFrame operator^(const R& o) const { return binary(PDX_BIT_XOR, o); }
  template <typename R> DataFrame operator<<(const R& o) const { return binary(PDX_SHIFT_LEFT, o); }
  template <typename R> DataFrame operator>>(const R& o) const { return binary(PDX_SHIFT_RIGHT, o); }
  template <typename R> DataFrame operator+(const R& o) const { return binary(PDX_ADD, o); }
  template <typename R> DataFrame operator-(const R& o) const { return binary(PDX_SUB, o); }
  template <typename R> DataFrame operator*(const R& o) const { return binary(PDX_MUL, o); }
  template <typename R> DataFrame operator/(const R& o) const { return binary(PDX_DIV, o); }

  // NDFrame::sum on a frame (src/ndframe.h:329-335): every column (chunk) summed, totals added in column order
  Scalar sum() const {
    bool first = true, is_f = false;
    double f = 0;
    int64_t i = 0;
    for (auto& c : m_columns) {
      Scalar s = Series(c).sum();
      if (!s.isValid()) continue;
      if (s.s.dtype == PDX_FLOAT64) { f = first ? s.s.v.f64 : f + s.s.v.f64; is_f = true; }
      else i = first ? s.s.v.i64 : (int64_t)((uint64_t)i + (uint64_t)s.s.v.i64);
      first = false;
    }
    if (first) return Scalar();
    return is_f ? Scalar(f) : Scalar(i);
  }
  // NDFrame::count/mean/min/max on a frame (src/ndframe.cpp:119-220 over GetInternalArray() = one ChunkedArray of all columns,
  // src/ndframe.h:329-335): chunk results folded in column order; mean = total of the per-chunk double sums / total valid count
  Scalar count() const {
    int64_t n = 0;
    for (auto& c : m_columns) n += Series(c).count().as<int64_t>();
    return Scalar(n);
  }
  Scalar mean() const {
    double tot = 0.0;
    int64_t cnt = 0;
    for (auto& c : m_columns) {
      Series col(c.dtype == PDX_FLOAT64 ? c : Series::run_cast_f64(c, false));  // int64 chunks sum as doubles (static_cast per value)
      Scalar s = col.sum();
      if (!s.isValid()) continue;
      tot += s.s.v.f64;
      cnt += s.s.count;
    }
    return cnt ? Scalar(tot / (double)cnt) : Scalar();
  }
  Scalar extreme(int kind) const {
    Scalar best;
    for (auto& c : m_columns) {  // the first of ties across chunks wins; a NaN chunk result never replaces a number
      Scalar x = Series(c).agg(kind);
      if (!x.isValid()) continue;
      if (!best.isValid()) { best = x; continue; }
      if (x.s.dtype == PDX_FLOAT64) {
        const double b = best.s.v.f64, v = x.s.v.f64;
        if ((b != b && v == v) || (kind == PDX_AGG_MIN ? v < b : v > b)) best = x;
      } else if (kind == PDX_AGG_MIN ? x.s.v.i64 < best.s.v.i64 : x.s.v.i64 > best.s.v.i64) {
        best = x;
      }
    }
    return best;
  }
  Scalar min() const { return extreme(PDX_AGG_MIN); }
  Scalar max() const { return extreme(PDX_AGG_MAX); }

  DataFrame where(const Series& mask) const {
    if (mask.dtype() != PDX_BOOL) throw std::runtime_error("filter mask must be boolean");
    auto outs = Series::run_filter(columns_with_index(), mask.m_array);
    return rebuild(outs);
  }
  DataFrame take(const Series& idx) const {
    if (idx.dtype() == PDX_BOOL) throw std::runtime_error("take indices must be integers, not boolean");
    auto outs = Series::run_take(columns_with_index(), idx.m_array);
    return rebuild(outs);
  }
  // DataFrame::sort_index (src/dataframe.cpp:1062-1071): the index sorted (array_sort_indices), the frame taken by the same indices.
  // Also the order-independent view of a group-by result: group ORDER is first occurrence here and a bounded permutation of it in
  // Arrow's Grouper (pdx/abi.h at pdx_groupby_create); sorted by key both frames are identical.
  DataFrame sort_index(bool ascending = true, bool ignore_index = false) const {
    DataFrame explicit_ix = *this;
    if (!explicit_ix.m_index) {  // the implicit range index, materialised (uint_range, src/ndframe.cpp:100-107)
      std::vector<int64_t> r((size_t)num_rows());
      for (size_t i = 0; i < r.size(); ++i) r[i] = (int64_t)i;
      explicit_ix.m_index = Array::Make(r);
    }
    const Series order = Series(*explicit_ix.m_index).argsort(ascending);
    auto outs = Series::run_take(explicit_ix.columns_with_index(), order.m_array);
    DataFrame r = explicit_ix.rebuild(outs);
    if (ignore_index) r.m_index.reset();
    return r;
  }

  // ---- Arrow IPC (src/dataframe.cpp:726-791).  toBinary: schema + ONE record batch + custom metadata; every column is written
  // (the reference computes `columns` and then serialises m_array whole); `index`: the index as a last int64 column of that name.
  std::vector<uint8_t> toBinary(const std::optional<std::string>& index = std::nullopt,
                                const std::map<std::string, std::string>& metadata = {}) const {
    std::vector<Array> cols = m_columns;
    std::vector<std::string> names = m_names;
    if (index) {
      Array ix;
      if (m_index) ix = *m_index;
      else {
        std::vector<int64_t> r((size_t)num_rows());
        for (size_t i = 0; i < r.size(); ++i) r[i] = (int64_t)i;
        ix = Array::Make(r);
      }
      ix.dtype = PDX_INT64;
      cols.push_back(ix);
      names.push_back(*index);
    }
    std::vector<pdx_column> in;
    std::vector<const char*> cn, kv;
    for (auto& c : cols) in.push_back(c.c());
    for (auto& n : names) cn.push_back(n.c_str());
    for (auto& m : metadata) { kv.push_back(m.first.c_str()); kv.push_back(m.second.c_str()); }
    void* blob = nullptr;
    size_t size = 0;
    ThrowOnFailure(pdx_ipc_write(in.data(), cn.data(), (int)in.size(), kv.data(), (int)kv.size() / 2, 0, nullptr, &blob, &size));
    std::vector<uint8_t> out(static_cast<uint8_t*>(blob), static_cast<uint8_t*>(blob) + size);
    pdx_ipc_free_blob(blob);
    return out;
  }
  // readBinary: exactly one record batch; ONE host->device copy of its body, the columns alias it (kept alive by their buffers)
  static DataFrame readBinary(const uint8_t* blob, size_t size, const std::optional<std::string>& index = std::nullopt) {
    pdx_ipc_frame* raw = nullptr;
    ThrowOnFailure(pdx_ipc_open(blob, size, &raw));
    std::shared_ptr<void> frame(raw, [](void* p) { pdx_ipc_destroy(static_cast<pdx_ipc_frame*>(p)); });
    ThrowOnFailure(pdx_ipc_load(raw, nullptr));
    std::vector<std::string> names;
    std::vector<Array> cols;
    std::optional<Array> idx;
    for (int i = 0; i < pdx_ipc_num_columns(raw); ++i) {
      pdx_column c{};
      ThrowOnFailure(pdx_ipc_column(raw, i, &c));
      Array a;
      a.dtype = c.dtype;
      a.length = c.length;
      a.null_count = c.null_count;
      a.values = std::make_shared<DeviceBuffer>(c.values, (size_t)c.length * 8, frame);
      if (c.validity) a.validity = std::make_shared<DeviceBuffer>(c.validity, bitmap_bytes(c.length), frame);
      const std::string nm = pdx_ipc_column_name(raw, i);
      if (index && nm == *index && !idx) {
        if (a.dtype == PDX_INT64) a.dtype = PDX_TIMESTAMP_NS;  // Cast(int64 -> timestamp[ns]) of the index column
        idx = a;
      } else {
        names.push_back(nm);
        cols.push_back(a);
      }
    }
    return DataFrame(names, cols, idx);
  }

  // readParquet (src/dataframe.cpp:646-683): ONE row group -> a frame on the device with the default range index.  The file is read
  // into host memory, the library walks the footer / page headers there, the column chunks travel in one copy and are decoded by kernels
  static DataFrame readParquet(const uint8_t* blob, size_t size) {
    pdx_parquet_file* raw = nullptr;
    ThrowOnFailure(pdx_parquet_open(blob, size, &raw));
    std::shared_ptr<void> file(raw, [](void* p) { pdx_parquet_destroy(static_cast<pdx_parquet_file*>(p)); });
    ThrowOnFailure(pdx_parquet_load(raw, nullptr));
    std::vector<std::string> names;
    std::vector<Array> cols;
    for (int i = 0; i < pdx_parquet_num_columns(raw); ++i) {
      pdx_column c{};
      ThrowOnFailure(pdx_parquet_column(raw, i, &c));
      Array a;
      a.dtype = c.dtype;
      a.length = c.length;
      a.null_count = c.null_count;
      a.values = std::make_shared<DeviceBuffer>(c.values, c.dtype == PDX_BOOL ? bitmap_bytes(c.length) : (size_t)c.length * 8, file);
      if (c.validity) a.validity = std::make_shared<DeviceBuffer>(c.validity, bitmap_bytes(c.length), file);
      names.push_back(pdx_parquet_column_name(raw, i));
      cols.push_back(a);
    }
    return DataFrame(names, cols, std::nullopt);
  }
  // toParquet(filepath, indexField) (src/dataframe.cpp:685-724): the columns (+ the index as a last column named indexField) as one row group
  void toParquet(const std::string& path, const std::string& indexField = "") const {
    std::vector<Array> cols = m_columns;
    std::vector<std::string> names = m_names;
    if (!indexField.empty()) {
      if (m_index) cols.push_back(*m_index);
      else {
        std::vector<int64_t> r((size_t)num_rows());
        for (size_t i = 0; i < r.size(); ++i) r[i] = (int64_t)i;
        cols.push_back(Array::Make(r));
      }
      names.push_back(indexField);
    }
    std::vector<pdx_column> cc;
    std::vector<const char*> cn;
    for (size_t i = 0; i < cols.size(); ++i) {
      cc.push_back(cols[i].c());
      cn.push_back(names[i].c_str());
    }
    void* blob = nullptr;
    size_t size = 0;
    ThrowOnFailure(pdx_parquet_write(cc.data(), cn.data(), (int)cc.size(), /*columns_on_host=*/0, nullptr, &blob, &size));
    std::FILE* fh = std::fopen(path.c_str(), "wb");
    const bool ok = fh && std::fwrite(blob, 1, size, fh) == size;
    if (fh) std::fclose(fh);
    pdx_parquet_free_blob(blob);
    if (!ok) throw std::runtime_error("IOError: Failed to write '" + path + "'");
  }
  static DataFrame readParquet(const std::string& path) {
    std::FILE* fh = std::fopen(path.c_str(), "rb");
    if (!fh) throw std::runtime_error("IOError: Failed to open local file '" + path + "'");
    std::vector<uint8_t> bytes;
    uint8_t buf[1 << 16];
    for (size_t got; (got = std::fread(buf, 1, sizeof buf, fh)) > 0;) bytes.insert(bytes.end(), buf, buf + got);
    std::fclose(fh);
    return readParquet(bytes.data(), bytes.size());
  }

  inline GroupBy group_by(const std::string& key) const;
  inline Resampler resample(const std::string& rule, bool closed_right = false, bool label_right = false) const;
  // DataFrame::downsample (src/dataframe.h:575-578, src/dataframe.cpp:1265-1290)
  inline Resampler downsample(const std::string& rule, bool closed_label_right = true, bool weekStartsMonday = true,
                              bool startEpoch = true) const;

 private:
  void check() const {
    for (auto& c : m_columns)
      if (c.length != num_rows()) throw std::runtime_error("all columns must have the same length");
  }
  std::vector<Array> columns_with_index() const {
    std::vector<Array> cols = m_columns;
    if (m_index) cols.push_back(*m_index);
    return cols;
  }
  DataFrame rebuild(std::vector<Array>& outs) const {
    std::optional<Array> idx;
    if (m_index) {
      idx = outs.back();
      outs.pop_back();
    }
    return DataFrame(m_names, outs, idx);
  }
};

// Scalar op DataFrame (BinaryImpl(DataFrame), src/scalar.cpp:24-28): the same scalar-lhs kernel over every column
inline DataFrame scalar_lhs(int op, const Scalar& a, const DataFrame& df) {
  std::vector<Array> out;
  for (auto& c : df.m_columns) out.push_back(scalar_lhs(op, a, Series(c), false).m_array);
  return DataFrame(df.m_names, out, df.m_index);
}
inline DataFrame operator+(const Scalar& a, const DataFrame& b) { return scalar_lhs(PDX_ADD, a, b); }
inline DataFrame operator-(const Scalar& a, const DataFrame& b) { return scalar_lhs(PDX_SUB, a, b); }
inline DataFrame operator*(const Scalar& a, const DataFrame& b) { return scalar_lhs(PDX_MUL, a, b); }
inline DataFrame operator/(const Scalar& a, const DataFrame& b) { return scalar_lhs(PDX_DIV, a, b); }

// ---------------------------------------------------------------- pd::GroupBy (src/group_by.h:22-299)
struct GroupBy {
  DataFrame df;
  std::shared_ptr<GroupHandle> handle;
  int key_dtype = PDX_INT64;

  GroupBy(const std::string& key, DataFrame frame) : df(std::move(frame)) {  // the ctor runs makeGroups(key) (group_by.h:24-31)
    const Array& k = df.m_columns[(size_t)df.column_index(key)];
    key_dtype = k.dtype;
    auto c = k.c();
    pdx_groupby* h = nullptr;
    ThrowOnFailure(pdx_groupby_create(&c, nullptr, &h));
    handle = std::make_shared<GroupHandle>(h);
  }
  GroupBy(DataFrame frame, std::shared_ptr<GroupHandle> h, int kd) : df(std::move(frame)), handle(std::move(h)), key_dtype(kd) {}

  size_t groupSize() const { return (size_t)pdx_groupby_num_groups(handle->h); }
  Array unique() const {  // uniqueKeys (group_by.h:52-55)
    Array out = Array::Empty(key_dtype, (int64_t)groupSize(), true);
    auto m = out.mut();
    ThrowOnFailure(pdx_groupby_unique_keys(handle->h, &m, nullptr));
    return out;
  }
  Array agg_array(const std::string& arg, int kind) const {
    const Array& v = df.m_columns[(size_t)df.column_index(arg)];
    const bool boolean = kind == PDX_AGG_ALL || kind == PDX_AGG_ANY, counting = kind == PDX_AGG_COUNT || kind == PDX_AGG_COUNT_DISTINCT;
    int out_dt = (kind == PDX_AGG_MEAN || kind == PDX_AGG_VARIANCE || kind == PDX_AGG_STDDEV) ? PDX_FLOAT64 : counting ? PDX_INT64 : boolean ? PDX_BOOL : v.dtype;
    Array out = Array::Empty(out_dt, (int64_t)groupSize(), boolean || (v.has_nulls() && !counting));
    auto c = v.c();
    auto m = out.mut();
    // the reference's constructor groups every column once (processEach, src/dataframe.cpp:1539-1554) and sum() / mean() / count()
    // reuse it: the first aggregation of a column binds it (one sort by group, kept in the handle together with the per-group
    // results); `df` keeps the buffers alive and unchanged for as long as the handle can look them up
    if (v.dtype == PDX_INT64 || v.dtype == PDX_FLOAT64) ThrowOnFailure(pdx_groupby_bind(handle->h, &c, nullptr));
    ThrowOnFailure(pdx_groupby_agg(handle->h, &c, &kind, 1, &m, nullptr));
    out.null_count = m.null_count;
    return out;
  }
  // GROUPBY_AGG / GROUPBY_NUMERIC_AGG overloads (src/pd_core_macros.h:5-147): one column -> Series, several -> DataFrame,
  // both indexed by uniqueKeys
  Series agg(const std::string& arg, int kind) const { return Series(agg_array(arg, kind), unique(), arg); }
  DataFrame agg(const std::vector<std::string>& args, int kind) const {
    std::vector<Array> cols;
    for (auto& a : args) cols.push_back(agg_array(a, kind));
    return DataFrame(args, cols, unique());
  }
  Series sum(const std::string& a) const { return agg(a, PDX_AGG_SUM); }
  Series mean(const std::string& a) const { return agg(a, PDX_AGG_MEAN); }
  Series min(const std::string& a) const { return agg(a, PDX_AGG_MIN); }
  Series max(const std::string& a) const { return agg(a, PDX_AGG_MAX); }
  Series count(const std::string& a) const { return agg(a, PDX_AGG_COUNT); }
  // src/dataframe.cpp:1516-1536 (variance, stddev, product), 1698-1810 (first, last)
  Series variance(const std::string& a) const { return agg(a, PDX_AGG_VARIANCE); }
  Series stddev(const std::string& a) const { return agg(a, PDX_AGG_STDDEV); }
  Series product(const std::string& a) const { return agg(a, PDX_AGG_PRODUCT); }
  Series first(const std::string& a) const { return agg(a, PDX_AGG_FIRST); }
  Series last(const std::string& a) const { return agg(a, PDX_AGG_LAST); }
  // GROUPBY_NUMERIC_AGG(all | any | count_distinct) (src/dataframe.cpp:1520-1526) and GroupBy::min_max (1602-1696)
  Series all(const std::string& a) const { return agg(a, PDX_AGG_ALL); }
  Series any(const std::string& a) const { return agg(a, PDX_AGG_ANY); }
  Series count_distinct(const std::string& a) const { return agg(a, PDX_AGG_COUNT_DISTINCT); }
  DataFrame min_max(const std::string& a) const { return DataFrame({"min", "max"}, {agg_array(a, PDX_AGG_MIN), agg_array(a, PDX_AGG_MAX)}); }
  DataFrame sum(const std::vector<std::string>& a) const { return agg(a, PDX_AGG_SUM); }
  DataFrame mean(const std::vector<std::string>& a) const { return agg(a, PDX_AGG_MEAN); }
  DataFrame min(const std::vector<std::string>& a) const { return agg(a, PDX_AGG_MIN); }
  DataFrame max(const std::vector<std::string>& a) const { return agg(a, PDX_AGG_MAX); }
  DataFrame count(const std::vector<std::string>& a) const { return agg(a, PDX_AGG_COUNT); }
  DataFrame variance(const std::vector<std::string>& a) const { return agg(a, PDX_AGG_VARIANCE); }
  DataFrame stddev(const std::vector<std::string>& a) const { return agg(a, PDX_AGG_STDDEV); }
  DataFrame product(const std::vector<std::string>& a) const { return agg(a, PDX_AGG_PRODUCT); }
  DataFrame first(const std::vector<std::string>& a) const { return agg(a, PDX_AGG_FIRST); }
  DataFrame last(const std::vector<std::string>& a) const { return agg(a, PDX_AGG_LAST); }

  // ---- walking the groups (src/group_by.h:39-77; src/dataframe.cpp:1354-1510).  The reference materialises every group's arrays in
  // the constructor (Grouper::MakeGroupings + ApplyGroupings); here the groupings are built on first use (pdx_groupby_groupings) and a
  // group's frame is one take of its rows.
  struct Groupings {
    Array rows;                    // int64, device: the rows of group 0, then of group 1, ..., each ascending
    std::vector<int64_t> offsets;  // G + 1
  };
  const Groupings& groupings() const {
    if (!m_groupings) {
      auto g = std::make_shared<Groupings>();
      const int64_t n = pdx_groupby_num_rows(handle->h), G = (int64_t)groupSize();
      g->rows = Array::Empty(PDX_INT64, n, false);
      DeviceBuffer off((size_t)(G + 1) * 8);
      ThrowOnFailure(pdx_groupby_groupings(handle->h, static_cast<int64_t*>(g->rows.values->ptr), static_cast<int64_t*>(off.ptr), nullptr));
      g->offsets.resize((size_t)G + 1);
      ThrowOnFailure(pdx_to_host(g->offsets.data(), off.ptr, (size_t)(G + 1) * 8, nullptr));
      m_groupings = g;
    }
    return *m_groupings;
  }
  Scalar GetKeyByIndex(int64_t i) const {  // uniqueKeys->GetScalar(i) (group_by.h:57-60)
    if (!m_keys) {
      Array u = unique();
      m_keys = std::make_shared<std::pair<std::vector<int64_t>, std::vector<bool>>>(u.values_as<int64_t>(), u.valid_flags());
      if (u.dtype == PDX_FLOAT64) m_keys_f = std::make_shared<std::vector<double>>(u.values_as<double>());
    }
    if (i < 0 || i >= (int64_t)m_keys->first.size()) throw std::runtime_error("Index " + std::to_string(i) + " out of bounds");
    if (!m_keys->second[(size_t)i]) return Scalar();
    return m_keys_f ? Scalar((*m_keys_f)[(size_t)i]) : Scalar(m_keys->first[(size_t)i]);
  }
  int64_t index_of_key(const Scalar& key) const {
    for (int64_t i = 0; i < (int64_t)groupSize(); ++i) {
      const Scalar k = GetKeyByIndex(i);
      if (k.isValid() != key.isValid()) continue;
      if (!k.isValid() || k.as<double>() == key.as<double>()) return i;
    }
    throw std::out_of_range("invalid key");  // groups.at(key) (group_by.h:41-49)
  }
  DataFrame MakeSubDataFrame(int64_t groupIndex) const {  // (group_by.h:62-73; the schema argument is the frame's own here)
    const Groupings& g = groupings();
    if (groupIndex < 0 || groupIndex >= (int64_t)groupSize()) throw std::runtime_error("Index " + std::to_string(groupIndex) + " out of bounds");
    Array idx = g.rows;
    idx.offset = g.offsets[(size_t)groupIndex];
    idx.length = g.offsets[(size_t)groupIndex + 1] - idx.offset;
    std::vector<Array> cols = df.m_columns;
    cols.push_back(df.m_index ? *df.m_index : row_numbers());
    auto outs = Series::run_take(cols, idx);
    Array index = outs.back();
    outs.pop_back();
    return DataFrame(df.m_names, outs, index);
  }
  DataFrame MakeSubDataFrame(const Scalar& key) const { return MakeSubDataFrame(index_of_key(key)); }
  template <typename T>
  std::vector<Array> group(T value) const { return MakeSubDataFrame(Scalar(value)).m_columns; }  // GroupBy::group (group_by.h:38-50)
  const DataFrame& getDF() const { return df; }

  // apply (src/dataframe.cpp:1430-1510): one call per group over its sub-frame; scalars -> Series indexed by the unique keys, arrays of
  // the groups' lengths -> concatenated in group order under the frame's own index; per column: Series -> scalar for every column
  Series apply(const std::function<Scalar(DataFrame const&)>& fn) const {
    std::vector<Scalar> r;
    for (int64_t i = 0; i < (int64_t)groupSize(); ++i) r.push_back(fn(MakeSubDataFrame(i)));
    return Series(build_array(r), unique());
  }
  inline Series apply(const std::function<Array(DataFrame const&)>& fn) const;
  DataFrame apply(const std::function<Scalar(Series const&)>& fn, bool index_keys = false) const {
    const int64_t G = (int64_t)groupSize();
    std::vector<DataFrame> subs;
    for (int64_t i = 0; i < G; ++i) subs.push_back(MakeSubDataFrame(i));
    std::vector<Array> cols;
    for (size_t c = 0; c < df.m_names.size(); ++c) {
      std::vector<Scalar> r;
      for (auto& sub : subs) r.push_back(fn(Series(sub.m_columns[c], sub.m_index, df.m_names[c])));
      cols.push_back(build_array(r));
    }
    return index_keys ? DataFrame(df.m_names, cols, unique()) : DataFrame(df.m_names, cols);
  }
  // apply_async (src/dataframe.cpp:1354-1408): the same results; the per-column form is indexed by the unique keys
  Series apply_async(const std::function<Scalar(DataFrame const&)>& fn) const { return apply(fn); }
  DataFrame apply_async(const std::function<Scalar(Series const&)>& fn) const { return apply(fn, true); }
  inline DataFrame apply_chunk(const std::function<DataFrame(DataFrame const&)>& fn) const;  // (src/dataframe.cpp:1411-1428)

  static Array build_array(const std::vector<Scalar>& r) {  // buildArray(ScalarVector): int64 unless a double is among the scalars
    bool any_f = false;
    for (auto& x : r) any_f = any_f || (x.isValid() && x.s.dtype == PDX_FLOAT64);
    std::vector<bool> valid;
    for (auto& x : r) valid.push_back(x.isValid());
    if (any_f) {
      std::vector<double> v;
      for (auto& x : r) v.push_back(x.isValid() ? x.as<double>() : 0.0);
      return Array::Make(v, &valid);
    }
    std::vector<int64_t> v;
    for (auto& x : r) v.push_back(x.isValid() ? x.as<int64_t>() : 0);
    return Array::Make(v, &valid);
  }

 private:
  Array row_numbers() const {  // the implicit index: uint64 0 .. n-1 (uint_range, src/ndframe.cpp:100-107)
    std::vector<int64_t> v((size_t)df.num_rows());
    for (size_t i = 0; i < v.size(); ++i) v[i] = (int64_t)i;
    Array a = Array::Make(v);
    a.dtype = PDX_UINT64;
    return a;
  }
  mutable std::shared_ptr<Groupings> m_groupings;
  mutable std::shared_ptr<std::pair<std::vector<int64_t>, std::vector<bool>>> m_keys;
  mutable std::shared_ptr<std::vector<double>> m_keys_f;
};

// ---------------------------------------------------------------- pd::Resampler (src/group_by.h:255-299)
struct Resampler : GroupBy {
  using GroupBy::GroupBy;
  Array index() const { return unique(); }
  // RESAMPLE_GROUP_BY_FUNCTION (group_by.h:249-253): aggregate ALL columns, index = labels of the non-empty bins
  DataFrame sum() const { return agg(df.m_names, PDX_AGG_SUM); }
  DataFrame mean() const { return agg(df.m_names, PDX_AGG_MEAN); }
  DataFrame min() const { return agg(df.m_names, PDX_AGG_MIN); }
  DataFrame max() const { return agg(df.m_names, PDX_AGG_MAX); }
  DataFrame count() const { return agg(df.m_names, PDX_AGG_COUNT); }
};

// rule string -> nanoseconds: splitTimeSpan + the unit table of pd::resample (src/resample.h:51-89); fixed durations only
inline int64_t rule_to_ns(const std::string& rule) {
  size_t p = 0;
  while (p < rule.size() && std::isdigit((unsigned char)rule[p])) ++p;
  int64_t mult = p ? std::stoll(rule.substr(0, p)) : 1;
  std::string u = rule.substr(p);
  if (u == "T" || u == "min") return mult * 60000000000LL;
  if (u == "S") return mult * 1000000000LL;
  if (u == "L" || u == "ms") return mult * 1000000LL;
  if (u == "U" || u == "us") return mult * 1000LL;
  if (u == "N" || u == "ns") return mult;
  throw std::runtime_error("resample rule '" + rule + "': only [T/min S L/ms U/us N/ns] are supported on this path");
}

inline Resampler resample(const DataFrame& df, int64_t freq_ns, bool closed_right = false, bool label_right = false,
                          int origin = PDX_ORIGIN_START_DAY, int64_t origin_custom_ns = 0, int64_t offset_ns = 0) {
  if (!df.m_index) throw std::runtime_error("axis must be a TimestampArray but got the implicit range index");
  auto c = df.m_index->c();
  pdx_groupby* h = nullptr;
  ThrowOnFailure(pdx_resample_create(&c, freq_ns, closed_right, label_right, origin, origin_custom_ns, offset_ns, nullptr, &h));
  return Resampler(df, std::make_shared<GroupHandle>(h), PDX_TIMESTAMP_NS);
}
inline Resampler resample(const DataFrame& df, const std::string& rule, bool closed_right = false, bool label_right = false) {
  return resample(df, rule_to_ns(rule), closed_right, label_right);
}
inline Resampler resample(const Series& s, int64_t freq_ns, bool closed_right = false, bool label_right = false) {
  return resample(DataFrame({s.name().empty() ? "0" : s.name()}, {s.m_array}, s.m_index), freq_ns, closed_right, label_right);
}
// getCalendarUnit (src/core.cpp:135-172): first letter of the rule's unit
inline int calendar_unit(char c) {
  switch (c) {
    case 'n': return PDX_UNIT_NANOSECOND;
    case 'u': return PDX_UNIT_MICROSECOND;
    case 'm': return PDX_UNIT_MILLISECOND;
    case 'S': return PDX_UNIT_SECOND;
    case 'T': return PDX_UNIT_MINUTE;
    case 'H': return PDX_UNIT_HOUR;
    case 'D': return PDX_UNIT_DAY;
    case 'Q': return PDX_UNIT_QUARTER;
    case 'W': return PDX_UNIT_WEEK;
    case 'M': return PDX_UNIT_MONTH;
    default: throw std::runtime_error(std::string("invalid unit got ") + c);
  }
}
// Ceil/FloorTemporal of the index -> (M / W / Y / Q and *E rules: one day less) -> Resampler keyed on the binned index
inline Resampler DataFrame::downsample(const std::string& rule, bool closed_label_right, bool weekStartsMonday, bool startEpoch) const {
  if (!m_index || m_index->dtype != PDX_TIMESTAMP_NS) throw std::runtime_error("downsample needs a timestamp[ns] index");
  size_t p = 0;
  while (p < rule.size() && !std::isalpha((unsigned char)rule[p])) ++p;  // splitTimeSpan (src/core.cpp:110-133)
  const int64_t mult = p ? std::stoll(rule.substr(0, p)) : 1;
  const std::string unit = rule.substr(p);
  if (unit.empty()) throw std::runtime_error("invalid unit got ");
  Array binned = Array::Empty(PDX_TIMESTAMP_NS, m_index->length, m_index->has_nulls());
  auto ci = m_index->c();
  auto mb = binned.mut();
  ThrowOnFailure(pdx_round_temporal(closed_label_right ? 1 : 0, &ci, mult, calendar_unit(unit[0]), weekStartsMonday, startEpoch, &mb, nullptr));
  binned.null_count = mb.null_count;
  if (unit.back() == 'E' || unit == "M" || unit == "W" || unit == "Y" || unit == "Q") {
    Array as_i64 = binned;
    as_i64.dtype = PDX_INT64;
    binned = Series::run_binary(PDX_SUB, as_i64, Scalar((int64_t)86400000000000LL).to_array(), true);
    binned.dtype = PDX_TIMESTAMP_NS;
  }
  // the grouping itself comes from the un-rounded index in one call (runs of equal labels on a sorted axis: no dictionary of `binned`)
  pdx_groupby* h = nullptr;
  const bool day_less = unit.back() == 'E' || unit == "M" || unit == "W" || unit == "Y" || unit == "Q";
  ThrowOnFailure(pdx_downsample_create(&ci, mult, calendar_unit(unit[0]), closed_label_right ? 1 : 0, weekStartsMonday, startEpoch,
                                       day_less ? -86400000000000LL : 0, nullptr, &h));
  return Resampler(DataFrame(m_names, m_columns, binned), std::make_shared<GroupHandle>(h), PDX_TIMESTAMP_NS);
}
inline GroupBy DataFrame::group_by(const std::string& key) const { return GroupBy(key, *this); }
inline Resampler DataFrame::resample(const std::string& rule, bool cr, bool lr) const { return pd::resample(*this, rule, cr, lr); }
inline Resampler Series::resample(const std::string& rule, bool cr, bool lr) const { return pd::resample(*this, rule_to_ns(rule), cr, lr); }

// date_range(start, periods, freq) for the tests (src/core.cpp:333-360 shape): timestamp[ns] index
inline Array date_range(int64_t start_ns, int periods, int64_t freq_ns = 60000000000LL) {
  std::vector<int64_t> t((size_t)periods);
  for (int i = 0; i < periods; ++i) t[(size_t)i] = start_ns + (int64_t)i * freq_ns;
  Array a = Array::Make(t);
  a.dtype = PDX_TIMESTAMP_NS;
  return a;
}

// ---------------------------------------------------------------- pd::concat rows (src/concat.h:56-64, src/concat.cpp:116-190)
inline Array concat_arrays(const std::vector<Array>& parts) {
  std::vector<pdx_column> in;
  int64_t total = 0;
  bool nulls = false;
  for (auto& p : parts) {
    in.push_back(p.c());
    total += p.length;
    nulls = nulls || p.has_nulls();
  }
  Array out = Array::Empty(parts[0].dtype, total, nulls);
  auto m = out.mut();
  ThrowOnFailure(pdx_concat(in.data(), (int)in.size(), &m, nullptr));
  out.null_count = m.null_count;
  return out;
}
// Concatenator::concatenateRows (src/concat.cpp:116-190): same-named columns are promoted to a common numeric type
// (resolveDuplicateFieldName + promoteTypes, src/concat.cpp:90-114, src/core.cpp:452-485), schemas unified in order of first
// appearance with nulls where a frame lacks a column; inner_join drops every column that holds a null
inline DataFrame concat(const std::vector<DataFrame>& dfs, bool ignore_index = false, bool inner_join = false) {
  if (dfs.empty()) throw std::runtime_error("concat of zero frames");
  std::vector<std::string> names;
  for (auto& d : dfs)
    for (auto& nm : d.m_names)
      if (std::find(names.begin(), names.end(), nm) == names.end()) names.push_back(nm);
  std::vector<Array> cols;
  std::vector<std::string> kept;
  for (auto& nm : names) {
    bool any_f = false, any_other = false;
    int other_dt = -1;
    for (auto& d : dfs) {
      auto it = std::find(d.m_names.begin(), d.m_names.end(), nm);
      if (it == d.m_names.end()) continue;
      const int dt = d.m_columns[(size_t)(it - d.m_names.begin())].dtype;
      if (dt == PDX_FLOAT64) any_f = true;
      else if (dt != PDX_INT64) {
        if (other_dt >= 0 && other_dt != dt) throw std::runtime_error("concat: no common numeric type for column " + nm);
        other_dt = dt;
        any_other = true;
      }
    }
    if (any_other && any_f) throw std::runtime_error("concat: no common numeric type for column " + nm);
    const int dt = any_other ? other_dt : any_f ? PDX_FLOAT64 : PDX_INT64;
    std::vector<Array> parts;
    for (auto& d : dfs) {
      auto it = std::find(d.m_names.begin(), d.m_names.end(), nm);
      if (it == d.m_names.end()) {
        std::vector<int64_t> zeros((size_t)d.num_rows(), 0);
        const std::vector<bool> none((size_t)d.num_rows(), false);
        Array nul = Array::Make(zeros, &none);
        nul.dtype = dt;
        parts.push_back(nul);
      } else {
        Array c = d.m_columns[(size_t)(it - d.m_names.begin())];
        if (c.dtype != dt) c = Series::run_cast_f64(c, true);  // arrow::compute::Cast(column, double), default (safe) options
        parts.push_back(c);
      }
    }
    Array col = concat_arrays(parts);
    if (inner_join && col.has_nulls() && col.null_count != 0) continue;
    cols.push_back(col);
    kept.push_back(nm);
  }
  std::optional<Array> index;
  if (!ignore_index) {  // each frame's index is carried along: [0,1,0,1] (tests/concat_test.cpp:40-50)
    std::vector<Array> parts;
    for (auto& d : dfs) {
      if (d.m_index) parts.push_back(*d.m_index);
      else {
        std::vector<int64_t> r((size_t)d.num_rows());
        for (size_t i = 0; i < r.size(); ++i) r[i] = (int64_t)i;
        parts.push_back(Array::Make(r));
      }
    }
    index = concat_arrays(parts);
  }
  return DataFrame(kept, cols, index);
}

// ---------------------------------------------------------------- row-range shards over the GPUs of one node (pdx_dist_*)
// The reference has no distributed code: this is what its host code would call to shard df.group_by(key).sum / mean / count over one
// process per GPU.  Every process holds a row range of the frame (ranks in row order) and ends with the FULL result, bit-identical
// to the single-process one.  The unique id travels by whatever channel the host has (MPI, a file, a socket): 128 bytes.
namespace dist {
struct Communicator {
  pdx_dist* h = nullptr;
  static std::array<char, 128> unique_id() {  // rank 0
    std::array<char, 128> id{};
    ThrowOnFailure(pdx_dist_unique_id(id.data()));
    return id;
  }
  Communicator(const std::array<char, 128>& id, int world, int rank) { ThrowOnFailure(pdx_dist_init(id.data(), world, rank, &h)); }
  ~Communicator() { pdx_dist_destroy(h); }
  Communicator(const Communicator&) = delete;
  Communicator& operator=(const Communicator&) = delete;
  int world() const { return pdx_dist_world(h); }
  int rank() const { return pdx_dist_rank(h); }
};
// shard.group_by(key).{sum, mean, count}(col) over all shards: a frame indexed by the global unique keys (first-occurrence order over
// the WHOLE column) with columns "sum", "mean", "count"; row_offset = index of the shard's first row in the whole frame
inline DataFrame group_by_sum_mean_count(Communicator& comm, const DataFrame& shard, const std::string& key, const std::string& col, int64_t row_offset) {
  const Array& k = shard.m_columns[(size_t)shard.column_index(key)];
  const Array& v = shard.m_columns[(size_t)shard.column_index(col)];
  auto ck = k.c(), cv = v.c();
  pdx_dist_groupby* raw = nullptr;
  ThrowOnFailure(pdx_dist_groupby_sum_mean_count(comm.h, &ck, &cv, row_offset, nullptr, &raw));
  std::shared_ptr<pdx_dist_groupby> g(raw, [](pdx_dist_groupby* p) { pdx_dist_groupby_destroy(p); });
  const int64_t G = pdx_dist_groupby_num_groups(raw);
  Array keys = Array::Empty(k.dtype, G, true), sums = Array::Empty(PDX_FLOAT64, G, false), means = Array::Empty(PDX_FLOAT64, G, false),
        counts = Array::Empty(PDX_INT64, G, false);
  auto mk = keys.mut();
  ThrowOnFailure(pdx_dist_groupby_fetch(raw, &mk, nullptr, static_cast<double*>(sums.values->ptr), static_cast<double*>(means.values->ptr),
                                        static_cast<int64_t*>(counts.values->ptr), nullptr));
  return DataFrame({"sum", "mean", "count"}, {sums, means, counts}, keys);
}
// shard.group_by(key).{min, max, count}(col) -- and the sum of an int64 column -- over all shards (pdx_dist_groupby_order_free: every rank reduces
// its shard without a value sort, dense per-group partials, one all-gather, fold in rank order): a frame indexed by the global unique keys with
// one column per kind, named "sum" / "min" / "max" / "count".  Values may carry nulls.
inline DataFrame group_by_order_free(Communicator& comm, const DataFrame& shard, const std::string& key, const std::string& col, const std::vector<int>& kinds,
                                     int64_t row_offset) {
  const Array& k = shard.m_columns[(size_t)shard.column_index(key)];
  const Array& v = shard.m_columns[(size_t)shard.column_index(col)];
  auto ck = k.c(), cv = v.c();
  pdx_dist_agg* raw = nullptr;
  ThrowOnFailure(pdx_dist_groupby_order_free(comm.h, &ck, &cv, kinds.data(), (int)kinds.size(), row_offset, nullptr, &raw));
  std::shared_ptr<pdx_dist_agg> g(raw, [](pdx_dist_agg* p) { pdx_dist_agg_destroy(p); });
  const int64_t G = pdx_dist_agg_num_groups(raw);
  Array keys = Array::Empty(k.dtype, G, true);
  std::vector<Array> outs;
  std::vector<std::string> names;
  std::vector<pdx_mut_column> mo;
  for (int kind : kinds) {
    outs.push_back(Array::Empty(kind == PDX_AGG_COUNT || kind == PDX_AGG_SUM ? PDX_INT64 : v.dtype, G, true));
    names.push_back(kind == PDX_AGG_SUM ? "sum" : kind == PDX_AGG_MIN ? "min" : kind == PDX_AGG_MAX ? "max" : "count");
  }
  for (auto& o : outs) mo.push_back(o.mut());
  auto mk = keys.mut();
  ThrowOnFailure(pdx_dist_agg_fetch(raw, &mk, nullptr, mo.data(), nullptr));
  for (size_t i = 0; i < outs.size(); ++i) {
    outs[i].null_count = mo[i].null_count;
    if (mo[i].null_count == 0) outs[i].validity.reset();
  }
  return DataFrame(names, outs, keys);
}
// pd::resample(shard, rule).agg(kind)(col) over a sorted axis sharded by row ranges: a Series of the aggregate indexed by the labels of the
// non-empty bins of the WHOLE axis (every rank gets all of it)
inline Series resample_agg(Communicator& comm, const Array& ts_shard, const Array& values_shard, int kind, int64_t freq_ns, bool closed_right = false,
                           bool label_right = false, int origin = PDX_ORIGIN_START_DAY, int64_t offset_ns = 0) {
  auto ct = ts_shard.c(), cv = values_shard.c();
  pdx_dist_resampled* raw = nullptr;
  ThrowOnFailure(pdx_dist_resample(comm.h, &ct, &cv, &kind, 1, freq_ns, closed_right, label_right, origin, 0, offset_ns, nullptr, &raw));
  std::shared_ptr<pdx_dist_resampled> g(raw, [](pdx_dist_resampled* p) { pdx_dist_resampled_destroy(p); });
  const int64_t G = pdx_dist_resampled_num_bins(raw);
  const int dt = (kind == PDX_AGG_MEAN || kind == PDX_AGG_VARIANCE || kind == PDX_AGG_STDDEV) ? PDX_FLOAT64 : kind == PDX_AGG_COUNT ? PDX_INT64 : values_shard.dtype;
  Array labels = Array::Empty(PDX_TIMESTAMP_NS, G, false), out = Array::Empty(dt, G, true);
  auto ml = labels.mut(), mo = out.mut();
  ThrowOnFailure(pdx_dist_resampled_fetch(raw, &ml, &mo, nullptr));
  out.null_count = mo.null_count;
  if (mo.null_count == 0) out.validity.reset();
  return Series(out, labels, "");
}
// pd::concat of the shards' columns in rank order (the all-gather(v) merge)
inline Array concat(Communicator& comm, const Array& part, int64_t total_rows) {
  Array out = Array::Empty(part.dtype, total_rows, true);
  auto c = part.c();
  auto m = out.mut();
  ThrowOnFailure(pdx_dist_concat(comm.h, &c, &m, nullptr));
  out.length = m.length;
  out.null_count = m.null_count;
  if (m.null_count == 0) out.validity.reset();
  return out;
}
}  // namespace dist

// GroupBy::apply(fn -> Array) and apply_chunk need concat: defined here
inline Series GroupBy::apply(const std::function<Array(DataFrame const&)>& fn) const {
  std::vector<Array> parts;
  for (int64_t i = 0; i < (int64_t)groupSize(); ++i) {
    DataFrame sub = MakeSubDataFrame(i);
    Array r = fn(sub);
    if (r.length != sub.num_rows())
      throw std::runtime_error("Failed to Merge Apply::Functor due to inconsistent Row Length\n" + std::to_string(r.length) + " != " + std::to_string(sub.num_rows()));
    parts.push_back(r);
  }
  return Series(concat_arrays(parts), df.m_index);
}
inline DataFrame GroupBy::apply_chunk(const std::function<DataFrame(DataFrame const&)>& fn) const {
  std::vector<DataFrame> parts;
  for (int64_t i = 0; i < (int64_t)groupSize(); ++i) parts.push_back(fn(MakeSubDataFrame(i)));
  return concat(parts);
}

}  // namespace pd
