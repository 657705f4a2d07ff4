// test_facade.cpp -- the reference's own Catch2 cases for the hot path, replayed against the C++ facade (pandasarrow_amd/cpp/
// pdx.hpp -> C ABI -> HIP kernels).  Each block cites the reference test it mirrors (file:line under the reference repository).
// Built with g++ (host code only) and run on the GPU box by tests/test_gpu_cpp_facade.py.
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "pdx.hpp"

static int g_checks = 0, g_failed = 0;
#define REQUIRE(cond)                                                              \
  do {                                                                             \
    ++g_checks;                                                                    \
    if (!(cond)) {                                                                 \
      ++g_failed;                                                                  \
      std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond);                \
    }                                                                              \
  } while (0)
#define REQUIRE_THROWS(expr)                                                       \
  do {                                                                             \
    ++g_checks;                                                                    \
    bool threw = false;                                                            \
    try { (void)(expr); } catch (const std::runtime_error&) { threw = true; }      \
    if (!threw) {                                                                  \
      ++g_failed;                                                                  \
      std::printf("FAILED %s:%d  expected std::runtime_error: %s\n", __FILE__, __LINE__, #expr); \
    }                                                                              \
  } while (0)
static bool approx(double a, double b) { return std::fabs(a - b) <= 1e-9 * (1 + std::fabs(b)); }
using namespace pd;

// tests/series_arithmetric_test.cpp:12-118
static void test_series_math() {
  std::vector<long> int_vec = {1, 2, 3, 4, 5};
  Series int_series(int_vec);
  REQUIRE((int_series + 2).at(0).as<long>() == 3);
  REQUIRE((int_series - 2).at(0).as<long>() == -1);
  REQUIRE((int_series * 2).at(0).as<long>() == 2);
  REQUIRE((int_series / 2).at(0).as<long>() == 0);
  REQUIRE((-int_series).at(0).as<long>() == -1);
  std::vector<double> double_vec = {1.1, 2.2, 3.3, 4.4, 5.5};
  Series double_series(double_vec);
  REQUIRE(approx((double_series + 2).at(0).as<double>(), 3.1));
  REQUIRE(approx((double_series - 2).at(0).as<double>(), -0.9));
  REQUIRE(approx((double_series * 2).at(0).as<double>(), 2.2));
  REQUIRE(approx((double_series / 2).at(0).as<double>(), 0.55));
  REQUIRE(approx((-double_series).at(0).as<double>(), -1.1));
  std::vector<long> int_vec2 = {2, 4, 6, 8, 10};
  Series int_series2(int_vec2);
  REQUIRE((int_series + int_series2).at(0).as<long>() == 3);
  REQUIRE((int_series / int_series2).at(0).as<long>() == 0);
  auto diff = double_series - int_series2;  // mixed double (-) int64 -> float64
  REQUIRE(diff.dtype() == PDX_FLOAT64);
  REQUIRE(diff.name() == "");
  auto dv = diff.values<double>();
  const double exp[] = {-0.9, -1.8, -2.7, -3.6, -4.5};
  for (int i = 0; i < 5; ++i) REQUIRE(approx(dv[i], exp[i]));
  // Scalar on the LEFT (src/scalar.cpp:24-41): the scalar stays the first operand of subtract / divide
  REQUIRE(((Scalar(10) - int_series).values<long>() == std::vector<long>{9, 8, 7, 6, 5}));
  REQUIRE(((Scalar(10) / int_series).values<long>() == std::vector<long>{10, 5, 3, 2, 2}));
  REQUIRE(((Scalar(3) + int_series).values<long>() == std::vector<long>{4, 5, 6, 7, 8}));
  REQUIRE(approx((Scalar(1.0) / double_series).at(1).as<double>(), 1.0 / 2.2));
  REQUIRE(((Scalar(3) < int_series).values<bool>() == std::vector<bool>{false, false, false, true, true}));
  REQUIRE_THROWS(Scalar(1) / Series(std::vector<long>{1, 0}));
  Series named(int_vec, "int_series2");
  auto add = int_series + named;
  REQUIRE(add.size() == 5 && add.dtype() == PDX_INT64 && add.name() == "");
  REQUIRE((add.values<long>() == std::vector<long>{2, 4, 6, 8, 10}));
  REQUIRE(((int_series * named).values<long>() == std::vector<long>{1, 4, 9, 16, 25}));
  REQUIRE(((int_series / named).values<long>() == std::vector<long>{1, 1, 1, 1, 1}));
  REQUIRE_THROWS(int_series / Series(std::vector<long>{1, 0, 1, 1, 1}));  // ArrowInvalid: divide by zero
  REQUIRE_THROWS(int_series + Series(std::vector<long>{1, 2, 3}));        // length mismatch
}

// tests/series_indexing_test.cpp:12-74
static void test_series_where_take() {
  Series s1(std::vector<int>{1, 2, 3, 4, 5});
  REQUIRE_THROWS(Series(Array::Make(std::vector<int>{1, 2, 3, 4, 5}), std::nullopt, "", /*is_index=*/true)
                     .where(Series(std::vector<bool>{true, false, true, false, true})));
  REQUIRE_THROWS(s1.where(Series(std::vector<bool>{false, true, true})));  // mask of a different size
  REQUIRE_THROWS(s1.take(Series(std::vector<bool>{false, true, true, true, false})));
  Series result = s1.take(Series(std::vector<int32_t>{1, 3, 4}));
  REQUIRE(result.size() == 3);
  REQUIRE((result.values<int>() == std::vector<int>{2, 4, 5}));
  REQUIRE_THROWS(s1.take(Series(std::vector<int>{0, 7})));  // ArrowIndexError
  // where(mask, Scalar): the scalar where the mask is false (tests/series_indexing_test.cpp:36-52)
  Series filled = s1.where(Series(std::vector<bool>{false, true, true, true, false}), Scalar(3));
  REQUIRE((filled.values<int>() == std::vector<int>{3, 2, 3, 4, 3}));
  REQUIRE((s1.if_else(Series(std::vector<bool>{true, false, true, false, true}), Series(std::vector<int>{9, 8, 7, 6, 5})).values<int>() ==
           std::vector<int>{1, 8, 3, 6, 5}));
  auto picked = s1[s1 > Scalar(2)];                         // operator[](bool Series) == where
  REQUIRE((picked.values<int>() == std::vector<int>{3, 4, 5}));
}

// tests/series_aggregation_test.cpp:122-196 ; NaN -> null on construction tests/series_test.cpp:127-186
static void test_series_aggregations() {
  // all / any / count_na / unique / nunique (tests/series_aggregation_test.cpp:12-120)
  REQUIRE(Series(std::vector<bool>{false, true, true, false, true}).all() == false);
  REQUIRE(Series(std::vector<bool>{true, true, true, true, true}).all() == true);
  REQUIRE_THROWS(Series(std::vector<int>{1, 2, 3, 4, 5}).any());
  REQUIRE(Series(std::vector<bool>{true, false, true, true, false}).any() == true);
  REQUIRE(Series(std::vector<bool>{false, false, false, false, false}).any() == false);
  REQUIRE(Series(std::vector<int>{1, 2, 3, 4, 5}).count_na() == 0);
  {
    const std::vector<bool> v7{true, true, true, true, true, true, false};
    REQUIRE(Series(Array::Make(std::vector<int>{1, 2, 3, 4, 5, 6, 7}, &v7)).count_na() == 1);
    const std::vector<bool> v13{true, true, true, true, true, true, true, true, true, true, true, true, false};
    REQUIRE(Series(Array::Make(std::vector<int>{1, 2, 3, 4, 5, 0, 7, 1, 2, 3, 4, 5, -1}, &v13)).nunique() == 7);
  }
  REQUIRE(Series(std::vector<int>{1, 2, 3, 4, 5, 0, 7, 1, 2, 3, 4, 5}).nunique() == 7);
  REQUIRE((Series(std::vector<int>{1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 6}).unique().values<int>() == std::vector<int>{1, 2, 3, 4, 5, 6}));
  Series s(std::vector<int>{1, 2, 3, 4, 5});
  REQUIRE(s.min().as<int>() == 1 && s.max().as<int>() == 5);
  Series sn(std::vector<int>{1, 2, 3, 4, 5}, std::vector<bool>{true, true, true, true, false});
  REQUIRE(sn.min().as<int>() == 1 && sn.max().as<int>() == 4);
  REQUIRE(s.mean() == 3.0);
  REQUIRE(sn.mean() == 2.5);
  REQUIRE(Series(std::vector<int>{1, 2, 3, 4, 5}, std::vector<bool>{false, true, true, true, true}).mean() == 3.5);
  Series nan_series(std::vector<double>{1.0, std::nan(""), 3.0});
  REQUIRE(nan_series.count().as<long>() == 2);
  REQUIRE(nan_series.sum() == 4.0);
  REQUIRE(!Series(std::vector<double>{}).sum().isValid());  // min_count = 1
}

// Series::broadcast / reindex (src/series.cpp:212-227, 1255-1309): labels {1,2,3} + {2,3,4} -> {1,2,3,4}, ends null
static void test_broadcast_unequal_indexes() {
  Series a(Array::Make(std::vector<double>{10.0, 20.0, 30.0}), Array::Make(std::vector<long>{3, 1, 2}), "a");
  Series b(Array::Make(std::vector<double>{1.0, 2.0, 3.0}), Array::Make(std::vector<long>{2, 4, 3}), "b");
  Series c = a + b;
  REQUIRE(c.size() == 4);
  REQUIRE((c.m_index->values_as<long>() == std::vector<long>{1, 2, 3, 4}));
  REQUIRE(!c.at(0).isValid() && !c.at(3).isValid());
  REQUIRE(c.at(1) == 31.0);  // label 2: 30 + 1
  REQUIRE(c.at(2) == 13.0);  // label 3: 10 + 3
  Series r = a.reindex(Array::Make(std::vector<long>{2, 9}));
  REQUIRE(r.at(0) == 30.0 && !r.at(1).isValid());
}

// tests/dataframe_iterator_test.cpp:11-77
static void test_groupby() {
  DataFrame df(std::map<std::string, std::vector<int32_t>>{{"a", {1, 1, 3, 1, 1, 1, 3, 8, 2, 2}}, {"b", {10, 9, 8, 7, 6, 5, 4, 3, 2, 1}}});
  auto groupby = df.group_by("a");
  REQUIRE(groupby.groupSize() == 4);
  REQUIRE((groupby.unique().values_as<long>() == std::vector<long>{1, 3, 8, 2}));  // first-occurrence order
  auto result = groupby.sum(std::vector<std::string>{"a", "b"});
  REQUIRE(result.num_rows() == 4 && result.num_columns() == 2);
  REQUIRE((result["a"].values<int64_t>() == std::vector<int64_t>{5, 6, 8, 4}));
  REQUIRE((result["b"].values<int64_t>() == std::vector<int64_t>{37, 12, 3, 3}));
  // DataFrame::sort_index (src/dataframe.cpp:1062-1071): the key-sorted view of a result compares equal whatever order the groups came in
  auto by_key = result.sort_index();
  REQUIRE((by_key.m_index->values_as<long>() == std::vector<long>{1, 2, 3, 8}));
  REQUIRE((by_key["b"].values<int64_t>() == std::vector<int64_t>{37, 3, 12, 3}));
  REQUIRE((result.sort_index(false)["a"].values<int64_t>() == std::vector<int64_t>{8, 6, 4, 5}));
  REQUIRE(!result.sort_index(true, true).m_index.has_value());
  // tests/cudf_examples/dataframe_resample_test.cpp:71-248 (gender male=0 female=1)
  DataFrame people({"gender", "age", "height"},
                   {Array::Make(std::vector<long>{0, 1, 0, 0, 1, 0, 0, 1, 0, 0}), Array::Make(std::vector<int>{16, 10, 10, 20, 30, 40, 15, 25, 35, 45}),
                    Array::Make(std::vector<int>{9, 9, 9, 9, 9, 8, 8, 8, 8, 8})});
  GroupBy g("gender", people);
  REQUIRE(g.groupSize() == 2);
  auto mean = g.mean(std::vector<std::string>{"age", "height"});
  REQUIRE(mean["age"].values<double>()[0] == 25.857142857142858);
  REQUIRE(mean["age"].values<double>()[1] == 21.666666666666668);
  REQUIRE(mean["height"].values<double>()[0] == 8.428571428571429);
  REQUIRE(mean["height"].values<double>()[1] == 8.666666666666666);
  REQUIRE((g.max("age").values<int>() == std::vector<int>{45, 30}));
  REQUIRE((g.min("age").values<int>() == std::vector<int>{10, 10}));
  REQUIRE((g.sum("age").values<int64_t>() == std::vector<int64_t>{181, 65}));
  REQUIRE((g.count("age").values<int64_t>() == std::vector<int64_t>{7, 3}));
  // the "next" group-by aggregations (src/dataframe.cpp:1516-1536, 1698-1810); expected values from Arrow 25.0.0's kernels
  REQUIRE(g.variance("age").values<double>()[0] == 164.40816326530611);
  REQUIRE(g.variance("age").values<double>()[1] == 72.22222222222223);
  REQUIRE(g.stddev("age").values<double>()[0] == 12.822174669895357);
  REQUIRE((g.product("age").values<int64_t>() == std::vector<int64_t>{3024000000LL, 7500}));
  REQUIRE((g.first("age").values<int>() == std::vector<int>{16, 10}));
  REQUIRE((g.last("age").values<int>() == std::vector<int>{45, 25}));
  // the remaining aggregations of src/dataframe.cpp:1520-1526, 1602-1696 on a small frame of our own
  DataFrame f2({"k", "b", "v"}, {Array::Make(std::vector<long>{1, 1, 3, 1, 3, 8}), Array::Make(std::vector<bool>{true, true, false, true, true, true}),
                                 Array::Make(std::vector<double>{1.0, 1.0, 2.0, 3.0, 2.0, 7.0})});
  auto g2 = f2.group_by("k");
  REQUIRE((g2.all("b").values<bool>() == std::vector<bool>{true, false, true}));
  REQUIRE((g2.any("b").values<bool>() == std::vector<bool>{true, true, true}));
  REQUIRE((g2.count_distinct("v").values<long>() == std::vector<long>{2, 1, 1}));
  auto mm = g2.min_max("v");
  REQUIRE((mm["min"].values<double>() == std::vector<double>{1.0, 2.0, 7.0}));
  REQUIRE((mm["max"].values<double>() == std::vector<double>{3.0, 2.0, 7.0}));
  // frame-level aggregates over all columns as chunks (src/ndframe.h:329-335)
  DataFrame f3({"x", "y"}, {Array::Make(std::vector<double>{1.0, 2.0}), Array::Make(std::vector<double>{3.0, 6.0})});
  REQUIRE(f3.sum().as<double>() == 12.0);
  REQUIRE(f3.mean().as<double>() == 3.0);
  REQUIRE(f3.min().as<double>() == 1.0);
  REQUIRE(f3.max().as<double>() == 6.0);
  REQUIRE(f3.count().as<long>() == 4);
  // Arrow IPC round trip (src/dataframe.cpp:726-791): toBinary with the index as a last int64 column, readBinary pulls it back out
  DataFrame f4({"x", "y"}, {Array::Make(std::vector<double>{1.5, 2.5, 3.5}), Array::Make(std::vector<long>{7, 8, 9})}, date_range(946684800000000000LL, 3));
  auto blob = f4.toBinary(std::string("__index__"), {{"origin", "facade"}});
  auto f5 = DataFrame::readBinary(blob.data(), blob.size(), std::string("__index__"));
  REQUIRE(f5.m_names == (std::vector<std::string>{"x", "y"}));
  REQUIRE((f5["x"].values<double>() == std::vector<double>{1.5, 2.5, 3.5}));
  REQUIRE((f5["y"].values<long>() == std::vector<long>{7, 8, 9}));
  REQUIRE(f5.m_index && f5.m_index->dtype == PDX_TIMESTAMP_NS);
  REQUIRE((f5.m_index->values_as<int64_t>() == f4.m_index->values_as<int64_t>()));
  REQUIRE((f5["x"] + f5["y"]).sum().as<double>() == 31.5);
}

// tests/series_resample_test.cpp:12-85
static void test_resample() {
  const int64_t t0 = 946684800000000000LL;  // 2000-01-01
  auto index = date_range(t0, 9);
  Series series(Array::Make(std::vector<long>{0, 1, 2, 3, 4, 5, 6, 7, 8}), index, "v");
  const int64_t min3 = 3 * 60000000000LL;
  {
    auto r = resample(series, min3);
    REQUIRE((r.index().values_as<int64_t>() == std::vector<int64_t>{t0, t0 + min3, t0 + 2 * min3}));
    REQUIRE((r.sum()["v"].values<long>() == std::vector<long>{3, 12, 21}));
  }
  {
    auto r = resample(series, min3, false, true);
    REQUIRE((r.index().values_as<int64_t>() == std::vector<int64_t>{t0 + min3, t0 + 2 * min3, t0 + 3 * min3}));
    REQUIRE((r.sum()["v"].values<long>() == std::vector<long>{3, 12, 21}));
  }
  {
    auto r = resample(series, min3, true, true);
    REQUIRE(r.index().length == 4);
    REQUIRE((r.index().values_as<int64_t>() == std::vector<int64_t>{t0, t0 + min3, t0 + 2 * min3, t0 + 3 * min3}));
    REQUIRE((r.sum()["v"].values<long>() == std::vector<long>{0, 6, 15, 15}));
  }
  REQUIRE((series.resample("3T").sum()["v"].values<long>() == std::vector<long>{3, 12, 21}));
  // DataFrame::downsample (tests/series_resample_test.cpp:87-129): floor / ceil_temporal bins of the same nine points
  DataFrame frame({"i"}, {series.m_array}, index);
  {
    auto r = frame.downsample("3T", false);
    REQUIRE((r.index().values_as<int64_t>() == std::vector<int64_t>{t0, t0 + min3, t0 + 2 * min3}));
    REQUIRE((r.sum()["i"].values<long>() == std::vector<long>{3, 12, 21}));
  }
  {
    auto r = frame.downsample("3T", true);
    REQUIRE((r.index().values_as<int64_t>() == std::vector<int64_t>{t0, t0 + min3, t0 + 2 * min3, t0 + 3 * min3}));
    REQUIRE((r.sum()["i"].values<long>() == std::vector<long>{0, 6, 15, 15}));
  }
  REQUIRE_THROWS(frame.downsample("3Y"));
}

// tests/concat_test.cpp:10-50 ; DataFrame element-wise tests/dataframe_arithmetric_test.cpp:49-195
static void test_concat_and_frame_ops() {
  DataFrame df1({"number"}, {Array::Make(std::vector<long>{1, 2})});
  DataFrame df3({"number"}, {Array::Make(std::vector<long>{3, 4})});
  auto r = concat({df1, df3});
  REQUIRE((r["number"].values<long>() == std::vector<long>{1, 2, 3, 4}));
  REQUIRE((r.m_index->values_as<long>() == std::vector<long>{0, 1, 0, 1}));
  // ConcatenateRows (tests/concat_test.cpp:591-749): schema union with null fill, int64 + double -> double, inner join
  {
    DataFrame d1({"a", "b"}, {Array::Make(std::vector<long>{1, 2, 3}), Array::Make(std::vector<long>{4, 5, 6})});
    DataFrame d2({"a", "e"}, {Array::Make(std::vector<double>{7.0, 8.0, 9.0}), Array::Make(std::vector<long>{13, 14, 15})});
    auto u = concat({d1, d2});
    REQUIRE((u.m_names == std::vector<std::string>{"a", "b", "e"}));
    REQUIRE(u["a"].dtype() == PDX_FLOAT64);
    REQUIRE((u["a"].values<double>() == std::vector<double>{1, 2, 3, 7, 8, 9}));
    REQUIRE(u["b"].m_array.null_count == 3);
    REQUIRE(u["e"].m_array.null_count == 3);
    auto in = concat({d1, d2}, /*ignore_index=*/false, /*inner_join=*/true);
    REQUIRE((in.m_names == std::vector<std::string>{"a"}));
    REQUIRE((in.m_index->values_as<long>() == std::vector<long>{0, 1, 2, 0, 1, 2}));
  }
  DataFrame a(std::map<std::string, std::vector<int32_t>>{{"x", {1, 2, 3}}, {"y", {4, 5, 6}}});
  DataFrame b({"x", "y"}, {Array::Make(std::vector<double>{0.5, 0.5, 0.5}), Array::Make(std::vector<double>{1.5, 1.5, 1.5})});
  auto c = a + b;  // int32 frame (+) double frame -> double
  REQUIRE(c["x"].dtype() == PDX_FLOAT64);
  REQUIRE((c["y"].values<double>() == std::vector<double>{5.5, 6.5, 7.5}));
  REQUIRE(((a * Scalar(2))["x"].values<long>() == std::vector<long>{2, 4, 6}));
  REQUIRE(a.sum().as<long>() == 21);
  // DataFrame::unary("negate" | "bit_wise_not"), UNARY_FUNCTION(abs | sign | sqrt), pow (src/dataframe.cpp:251-275, 919-935)
  REQUIRE(((-a)["x"].values<long>() == std::vector<long>{-1, -2, -3}));
  REQUIRE(((~a)["y"].values<long>() == std::vector<long>{-5, -6, -7}));
  REQUIRE(((-a).abs()["x"].values<long>() == std::vector<long>{1, 2, 3}));
  REQUIRE(((-a).sign()["y"].values<long>() == std::vector<long>{-1, -1, -1}));
  REQUIRE((a.pow(2.0)["y"].values<double>() == std::vector<double>{16.0, 25.0, 36.0}));
  REQUIRE((DataFrame({"q"}, {Array::Make(std::vector<double>{4.0, 6.25})}).sqrt()["q"].values<double>() == std::vector<double>{2.0, 2.5}));
  REQUIRE(((-Series(Array::Make(std::vector<double>{1.5, -2.0}))).values<double>() == std::vector<double>{-1.5, 2.0}));
  REQUIRE_THROWS(~b);  // bit_wise_not has no float64 kernel
  REQUIRE(((a | Scalar(8))["x"].values<long>() == std::vector<long>{9, 10, 11}));
  REQUIRE(((a & a)["y"].values<long>() == std::vector<long>{4, 5, 6}));
  REQUIRE(((a ^ Scalar(1))["x"].values<long>() == std::vector<long>{0, 3, 2}));
  REQUIRE(((a << Scalar(2))["y"].values<long>() == std::vector<long>{16, 20, 24}));
  REQUIRE((((-a) >> Scalar(1))["x"].values<long>() == std::vector<long>{-1, -1, -2}));  // arithmetic shift
  REQUIRE(((a << Scalar(64))["x"].values<long>() == std::vector<long>{1, 2, 3}));       // out-of-range amount: unchanged
  REQUIRE_THROWS(b | Scalar(1));
  REQUIRE_THROWS(a + DataFrame(std::map<std::string, std::vector<int32_t>>{{"x", {1, 2}}, {"y", {4, 5}}}));
  auto f = a[a["x"] > Scalar(1)];  // DataFrame::where through operator[]
  REQUIRE((f["y"].values<long>() == std::vector<long>{5, 6}));
}

// Series::argsort / sort / n_smallest (src/series.cpp:864-868, 978-992, 1211-1229): stable, index follows the values
static void test_sort() {
  Series s(Array::Make(std::vector<double>{3.0, 1.0, 2.0, 1.0}), Array::Make(std::vector<long>{10, 11, 12, 13}), "s");
  REQUIRE((s.argsort().values<long>() == std::vector<long>{1, 3, 2, 0}));
  Series d = s.sort(false);
  REQUIRE((d.values<double>() == std::vector<double>{3.0, 2.0, 1.0, 1.0}));
  REQUIRE((d.m_index->values_as<long>() == std::vector<long>{10, 12, 11, 13}));
  Series two = s.n_smallest(2);
  REQUIRE(two.size() == 2 && (two.values<double>() == std::vector<double>{1.0, 1.0}));
  REQUIRE((two.m_index->values_as<long>() == std::vector<long>{11, 13}));
  Series withnull(std::vector<double>{2.0, std::nan(""), 1.0});  // NaN -> null on construction: nulls last in both orders
  REQUIRE((withnull.argsort(false).values<long>() == std::vector<long>{0, 2, 1}));
}

// BINARY_OPERATOR_DF(> >= < <= == != && ||) (src/dataframe.cpp:563-577, src/dataframe.h:476-520) and DataFrame / Series::reindex with
// a fill value (src/dataframe.cpp:1139-1186, src/series.cpp:1295-1302; vectors: tests/dataframe_indexing_test.cpp:203-229)
static void test_frame_compare_logical_reindex() {
  DataFrame a({"x", "y"}, {Array::Make(std::vector<double>{1.0, 5.0, 3.0}), Array::Make(std::vector<double>{4.0, 0.5, 3.0})});
  DataFrame b({"x", "y"}, {Array::Make(std::vector<double>{2.0, 5.0, 1.0}), Array::Make(std::vector<double>{4.0, 1.5, std::nan("")})});
  REQUIRE(((a > b)["x"].values<int>() == std::vector<int>{0, 0, 1}));
  REQUIRE(((a >= b)["x"].values<int>() == std::vector<int>{0, 1, 1}));
  auto lt = (a < b)["y"].values<int>(), eq = (a == b)["y"].values<int>();  // (row 2 of b.y is null: only the valid rows are pinned)
  REQUIRE(lt[0] == 0 && lt[1] == 1 && eq[0] == 1 && eq[1] == 0);
  REQUIRE(((a != b)["y"].m_array.valid_flags() == std::vector<bool>{true, true, false}));  // NaN -> null on construction -> null out
  REQUIRE((a > b)["x"].dtype() == PDX_BOOL);
  Series s(std::vector<double>{1.0, 1.0, 3.0});
  REQUIRE(((a <= s)["x"].values<int>() == std::vector<int>{1, 0, 1}));
  REQUIRE(((a > Scalar(2.0))["y"].values<int>() == std::vector<int>{1, 0, 1}));
  DataFrame m = (a > Scalar(2.0)) && (a <= Scalar(4.0));
  REQUIRE((m["x"].values<int>() == std::vector<int>{0, 0, 1}));
  REQUIRE((m["y"].values<int>() == std::vector<int>{1, 0, 1}));
  REQUIRE((((a > Scalar(4.0)) || (a < Scalar(1.0)))["y"].values<int>() == std::vector<int>{0, 1, 0}));
  REQUIRE(((m || Scalar(true))["x"].values<int>() == std::vector<int>{1, 1, 1}));
  REQUIRE(((m && Scalar(false))["y"].values<int>() == std::vector<int>{0, 0, 0}));
  REQUIRE(((m && (a["x"] > Scalar(0.0)))["y"].values<int>() == std::vector<int>{1, 0, 1}));
  REQUIRE_THROWS(a > DataFrame({"x"}, {Array::Make(std::vector<double>{1.0, 2.0, 3.0})}));
  REQUIRE_THROWS(a == Series(std::vector<double>{1.0}));
  REQUIRE_THROWS(a && a);  // "and" has no float64 kernel
  REQUIRE((a[m["y"]]["x"].values<double>() == std::vector<double>{1.0, 3.0}));
  // tests/dataframe_indexing_test.cpp:203-229
  DataFrame in({"col1", "col2"}, {Array::Make(std::vector<long>{1, 2, 3, 4, 5}), Array::Make(std::vector<long>{5, 4, 3, 2, 1})},
               Array::Make(std::vector<long>{1, 2, 3, 4, 5}));
  Array newIndex = Array::Make(std::vector<long>{1, 2, 4, 5, 6});
  DataFrame out = in.reindex(newIndex);
  REQUIRE((out.m_index->values_as<long>() == std::vector<long>{1, 2, 4, 5, 6}));
  REQUIRE((out["col1"].m_array.valid_flags() == std::vector<bool>{true, true, true, true, false}));
  REQUIRE((out["col2"].m_array.valid_flags() == std::vector<bool>{true, true, true, true, false}));
  auto c1 = out["col1"].values<long>(), c2 = out["col2"].values<long>();
  REQUIRE((std::vector<long>(c1.begin(), c1.begin() + 4) == std::vector<long>{1, 2, 4, 5}));
  REQUIRE((std::vector<long>(c2.begin(), c2.begin() + 4) == std::vector<long>{5, 4, 2, 1}));
  // the same with a fill value (src/series.cpp:1295-1302): the absent label 6 takes it, nothing is null
  DataFrame filled = in.reindexAsync(newIndex, Scalar(-1));
  REQUIRE((filled["col1"].values<long>() == std::vector<long>{1, 2, 4, 5, -1}));
  REQUIRE((filled["col2"].values<long>() == std::vector<long>{5, 4, 2, 1, -1}));
  REQUIRE((filled["col1"].m_array.valid_flags() == std::vector<bool>(5, true)));
  Series sv(Array::Make(std::vector<double>{10.0, std::nan(""), 30.0}), Array::Make(std::vector<long>{7, 8, 7}), "v");
  Series r = sv.reindex(Array::Make(std::vector<long>{8, 7, 9}), Scalar(0.5));  // 8: present but null stays null; 7: LAST position; 9: filled
  REQUIRE((r.m_array.valid_flags() == std::vector<bool>{false, true, true}));
  REQUIRE(r.values<double>()[1] == 30.0 && r.values<double>()[2] == 0.5);
  REQUIRE_THROWS(sv.reindex(Array::Make(std::vector<long>{9}), Scalar(1)));  // int64 scalar into a double builder
}

// GroupBy binds a column on its first aggregation (reference: the constructor's processEach, src/dataframe.cpp:1539-1554):
// sum(); mean(); count() as three calls = one sort, one reduce -- the second and third are served from the handle's cache
static void test_groupby_bound_columns() {
  const int n = 50000;
  std::vector<long> key(n);
  std::vector<double> val(n);
  for (int i = 0; i < n; ++i) { key[i] = (i * 7919) % 97; val[i] = 0.25 * (i % 13) - 1.0; }
  DataFrame df({"k", "v"}, {Array::Make(key), Array::Make(val)});
  GroupBy g("k", df);
  auto s1 = g.sum("v").values<double>();
  char plan[256];
  ThrowOnFailure(pdx_groupby_last_plan(g.handle->h, plan, sizeof plan));
  REQUIRE(std::string(plan).find("bound=1 cache=fill") != std::string::npos);
  auto m1 = g.mean("v").values<double>();
  ThrowOnFailure(pdx_groupby_last_plan(g.handle->h, plan, sizeof plan));
  REQUIRE(std::string(plan).find("bound=1 cache=hit") != std::string::npos);
  auto c1 = g.count("v").values<long>();
  REQUIRE(pdx_groupby_bound_bytes(g.handle->h) >= (int64_t)n * 8);
  GroupBy g2("k", df);  // an independent handle gives the same bits
  REQUIRE((g2.sum("v").values<double>() == s1));
  REQUIRE((g2.mean("v").values<double>() == m1));
  long total = 0;
  for (long c : c1) total += c;
  REQUIRE(total == n);
  for (size_t i = 0; i < s1.size(); ++i) REQUIRE(m1[i] == s1[i] / (double)c1[i]);
  ThrowOnFailure(pdx_groupby_unbind(g.handle->h, nullptr));
  REQUIRE(pdx_groupby_bound_bytes(g.handle->h) == 0);
}

// The sharded path through the C ABI from C++ (pdx_dist_*: RCCL opened by the library, no python): one rank, every collective kept on
// the wire (PDX_DIST_FORCE_COLLECTIVES=1), against the single-GPU GroupBy of the same frame -- bit for bit
static void test_sharded_groupby_from_cpp() {
  setenv("PDX_DIST_FORCE_COLLECTIVES", "1", 1);
  const int n = 300007;
  std::vector<long> key(n);
  std::vector<double> val(n);
  uint64_t x = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    key[i] = (long)(x % 4099) * 7919 - 12345;
    val[i] = (double)(x >> 11) * 0x1.0p-53 - 0.5;
  }
  DataFrame df({"k", "v"}, {Array::Make(key), Array::Make(val)});
  dist::Communicator comm(dist::Communicator::unique_id(), 1, 0);
  REQUIRE(comm.world() == 1 && comm.rank() == 0);
  DataFrame res = dist::group_by_sum_mean_count(comm, df, "k", "v", 0);
  GroupBy g("k", df);
  REQUIRE((res.m_index->values_as<long>() == g.unique().values_as<long>()));
  REQUIRE((res["sum"].values<double>() == g.sum("v").values<double>()));
  REQUIRE((res["mean"].values<double>() == g.mean("v").values<double>()));
  REQUIRE((res["count"].values<long>() == g.count("v").values<long>()));
  // the order-free kinds over the "shards" (dense partials + all-gather + fold) == the single-GPU GroupBy
  DataFrame of = dist::group_by_order_free(comm, df, "k", "v", {PDX_AGG_MIN, PDX_AGG_MAX, PDX_AGG_COUNT}, 0);
  REQUIRE((of.m_index->values_as<long>() == g.unique().values_as<long>()));
  REQUIRE((of["min"].values<double>() == g.min("v").values<double>()));
  REQUIRE((of["max"].values<double>() == g.max("v").values<double>()));
  REQUIRE((of["count"].values<long>() == g.count("v").values<long>()));
  Array cat = dist::concat(comm, df.m_columns[1], n);
  REQUIRE(cat.length == n && (cat.values_as<double>() == val));
  Series withnull(std::vector<double>{1.0, std::nan(""), 3.0});
  Array cn = dist::concat(comm, withnull.m_array, 3);
  REQUIRE((cn.valid_flags() == std::vector<bool>{true, false, true}));
  // resample over the "sharded" axis == the single-GPU Resampler (tests/series_resample_test.cpp shape: 1-minute data, 5-minute bins)
  std::vector<long> ts(600);
  std::vector<double> tv(600);
  for (int i = 0; i < 600; ++i) { ts[(size_t)i] = 946684800000000000L + (long)i * 60000000000L; tv[(size_t)i] = 0.5 * (i % 11) - 2.0; }
  Array axis = Array::Make(ts);
  axis.dtype = PDX_TIMESTAMP_NS;
  Series rs = dist::resample_agg(comm, axis, Array::Make(tv), PDX_AGG_MEAN, 300000000000L);
  DataFrame tdf({"v"}, {Array::Make(tv)}, axis);
  DataFrame one = tdf.resample("5T").mean();
  REQUIRE(rs.size() == 120 && (rs.values<double>() == one["v"].values<double>()));
  REQUIRE((rs.m_index->values_as<long>() == one.m_index->values_as<long>()));
  unsetenv("PDX_DIST_FORCE_COLLECTIVES");
}

// DataFrame::toParquet / readParquet (src/dataframe.cpp:646-724) round trip through a file: device columns -> Parquet -> device columns
static void test_parquet_round_trip() {
  std::vector<bool> ok{true, false, true, true, false, true, true};
  DataFrame df({"x", "y", "flag"}, {Array::Make(std::vector<long>{1, -2, 3, 1L << 40, 5, 6, -7}), Array::Make(std::vector<double>{0.5, 1.5, -2.5, 3.5, 4.5, -0.0, 6.5}, &ok),
                                    Array::Make(std::vector<bool>{true, false, true, true, false, false, true})},
               Array::Make(std::vector<long>{10, 11, 12, 13, 14, 15, 16}));
  const std::string path = "/tmp/pdx_facade_test.parquet";
  df.toParquet(path, "idx");
  DataFrame back = DataFrame::readParquet(path);
  REQUIRE((back.m_names == std::vector<std::string>{"x", "y", "flag", "idx"}));
  REQUIRE((back["x"].values<long>() == std::vector<long>{1, -2, 3, 1L << 40, 5, 6, -7}));
  REQUIRE((back["y"].m_array.valid_flags() == ok));
  auto y = back["y"].values<double>();
  REQUIRE(y[0] == 0.5 && y[2] == -2.5 && y[3] == 3.5 && std::signbit(y[5]) && y[6] == 6.5);
  REQUIRE((back["flag"].values<int>() == std::vector<int>{1, 0, 1, 1, 0, 0, 1}));
  REQUIRE((back["idx"].values<long>() == std::vector<long>{10, 11, 12, 13, 14, 15, 16}));
  REQUIRE_THROWS(DataFrame::readParquet("/tmp/pdx_no_such_file.parquet"));
  std::remove(path.c_str());
}

// GroupBy::group / GetKeyByIndex / MakeSubDataFrame / apply / apply_chunk (src/group_by.h:39-77, src/dataframe.cpp:1354-1510) over
// pdx_groupby_groupings; the frame of tests/dataframe_iterator_test.cpp:11-44 (keys 1,3,8,2; sums of b 37,12,3,3)
static void test_groupby_walkers() {
  DataFrame df({"a", "b"}, {Array::Make(std::vector<long>{1, 1, 3, 1, 1, 1, 3, 8, 2, 2}), Array::Make(std::vector<double>{10, 9, 8, 7, 6, 5, 4, 3, 2, 1})});
  auto gb = df.group_by("a");
  REQUIRE(gb.groupSize() == 4);
  REQUIRE(gb.GetKeyByIndex(0).as<long>() == 1 && gb.GetKeyByIndex(1).as<long>() == 3 && gb.GetKeyByIndex(2).as<long>() == 8 && gb.GetKeyByIndex(3).as<long>() == 2);
  REQUIRE((gb.groupings().offsets == std::vector<int64_t>{0, 5, 7, 8, 10}));
  REQUIRE((gb.groupings().rows.values_as<int64_t>() == std::vector<int64_t>{0, 1, 3, 4, 5, 2, 6, 7, 8, 9}));
  DataFrame sub = gb.MakeSubDataFrame(1);
  REQUIRE((sub["b"].values<double>() == std::vector<double>{8, 4}));
  REQUIRE((sub.m_index->values_as<int64_t>() == std::vector<int64_t>{2, 6}));   // the group's rows of the (implicit) index
  REQUIRE((gb.group(8L)[1].values_as<double>() == std::vector<double>{3}));
  {  // groups.at(key) (group_by.h:41-49): std::out_of_range
    bool oor = false;
    try { (void)gb.group(99L); } catch (const std::out_of_range&) { oor = true; }
    REQUIRE(oor);
  }
  Series s = gb.apply([](DataFrame const& f) { return f["b"].sum(); });
  REQUIRE((s.values<double>() == std::vector<double>{37, 12, 3, 3}));
  REQUIRE((s.m_index->values_as<int64_t>() == std::vector<int64_t>{1, 3, 8, 2}));
  Series arr = gb.apply([](DataFrame const& f) { return (f["b"] * Scalar(2.0)).m_array; });
  REQUIRE((arr.values<double>() == std::vector<double>{20, 18, 14, 12, 10, 16, 8, 6, 4, 2}));
  REQUIRE_THROWS(gb.apply([](DataFrame const& f) { return Array::Make(std::vector<double>((size_t)f.num_rows() + 1, 0.0)); }));
  DataFrame pc = gb.apply([](Series const& c) { return c.max(); });
  REQUIRE((pc["b"].values<double>() == std::vector<double>{10, 8, 3, 2}));
  REQUIRE(!pc.m_index);
  DataFrame pa = gb.apply_async([](Series const& c) { return c.max(); });
  REQUIRE((pa.m_index->values_as<int64_t>() == std::vector<int64_t>{1, 3, 8, 2}));
  DataFrame ch = gb.apply_chunk([](DataFrame const& f) { return f * Scalar(1.0); });
  REQUIRE((ch["b"].values<double>() == std::vector<double>{10, 9, 7, 6, 5, 8, 4, 3, 2, 1}));
}

int main() {
  ThrowOnFailure(pdx_init(0));
  test_series_math();
  test_series_where_take();
  test_series_aggregations();
  test_broadcast_unequal_indexes();
  test_groupby();
  test_resample();
  test_concat_and_frame_ops();
  test_sort();
  test_frame_compare_logical_reindex();
  test_groupby_bound_columns();
  test_groupby_walkers();
  test_sharded_groupby_from_cpp();
  test_parquet_round_trip();
  std::printf("%d checks, %d failed\n", g_checks, g_failed);
  return g_failed ? 1 : 0;
}
