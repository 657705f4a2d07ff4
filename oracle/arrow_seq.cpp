// arrow_seq.cpp -- CPU BASELINE HARNESS (test/bench infrastructure, never part of the product).
//
// Replays the reference's group-by call sequence with Apache Arrow C++ itself -- the third-party library the reference forwards
// every numeric operation to (it owns no arithmetic loops on this path) -- so that bench.py's `cpu_baseline` times the real Arrow
// kernels on the GPU box's host cores instead of a restatement:
//
//   GroupBy::makeGroups        (reference src/dataframe.cpp:1571-1600):  Grouper::Make -> Consume -> MakeGroupings -> GetUniques
//   processIndex / processEach (src/dataframe.cpp:1539-1569):            ApplyGroupings of the index and of EVERY column (key, value)
//   GROUPBY_AGG(sum), GROUPBY_NUMERIC_AGG(mean), (count)                 (src/pd_core_macros.h:5-147): one CallFunction per group,
//                                                                        a thread pool over the groups standing in for tbb::parallel_for
//
// What it leaves out flatters the reference: the unordered_map<ScalarPtr, ArrayVector, HashScalar> bookkeeping (one GetScalar +
// map insert per group and column, src/dataframe.cpp:1546-1550) and the builder copies of the results.
// Inputs: the counter-based synthetic workload of SURVEY.md 8d (same generator as the oracle and the HIP library).
// Output: one JSON line on stdout (seconds per phase) and, with --out FILE, the raw result arrays
// [G int64 keys | G f64 sums | G f64 means | G int64 counts] for the bit-exact cross-check in bench.py.
//
// Built by oracle.arrow_seq_build() against the pyarrow wheel's headers and libarrow.so / libarrow_compute.so (Arrow 25.0.0 in
// this image); absent wheel -> no harness, bench.py falls back to the C restatement.
#include <arrow/api.h>
#include <arrow/compute/api.h>
#include <arrow/compute/row/grouper.h>
#include <arrow/util/config.h>
#include <arrow/util/thread_pool.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace cp = arrow::compute;

static uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

#define CHECK_OK(expr)                                                       \
  do {                                                                       \
    auto _st = (expr);                                                       \
    if (!_st.ok()) {                                                         \
      std::fprintf(stderr, "arrow_seq: %s\n", _st.ToString().c_str());       \
      return 2;                                                              \
    }                                                                        \
  } while (0)
template <typename T>
static T Unwrap(arrow::Result<T> r) {
  if (!r.ok()) {
    std::fprintf(stderr, "arrow_seq: %s\n", r.status().ToString().c_str());
    std::exit(2);
  }
  return std::move(r).ValueUnsafe();
}

int main(int argc, char** argv) {
  int64_t n = 1000000, nkeys = 1000;
  int threads = 1;
  std::string out_path;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--rows") && i + 1 < argc) n = (int64_t)std::atof(argv[++i]);
    else if (!std::strcmp(argv[i], "--keys") && i + 1 < argc) nkeys = (int64_t)std::atof(argv[++i]);
    else if (!std::strcmp(argv[i], "--threads") && i + 1 < argc) threads = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--out") && i + 1 < argc) out_path = argv[++i];
  }
  if (threads < 1) threads = 1;
  CHECK_OK(cp::Initialize());  // registers the compute kernels (separate libarrow_compute since Arrow 21)
  CHECK_OK(arrow::SetCpuThreadPoolCapacity(threads));

  // ---- inputs (not timed): key / value columns and the uint64 0..n-1 index the reference materialises (src/ndframe.cpp:100-107)
  auto kbuf = Unwrap(arrow::AllocateBuffer(n * 8));
  auto vbuf = Unwrap(arrow::AllocateBuffer(n * 8));
  auto ibuf = Unwrap(arrow::AllocateBuffer(n * 8));
  {
    auto* k = reinterpret_cast<int64_t*>(kbuf->mutable_data());
    auto* v = reinterpret_cast<double*>(vbuf->mutable_data());
    auto* ix = reinterpret_cast<uint64_t*>(ibuf->mutable_data());
    std::vector<std::thread> gen;
    for (int t = 0; t < threads; ++t)
      gen.emplace_back([&, t] {
        for (int64_t i = n * t / threads; i < n * (t + 1) / threads; ++i) {
          k[i] = (int64_t)(splitmix64((uint64_t)i ^ 0x5EED0001ull) % (uint64_t)nkeys);
          v[i] = (double)(splitmix64((uint64_t)i + 0x5EED0002ull) >> 11) * 0x1.0p-53;
          ix[i] = (uint64_t)i;
        }
      });
    for (auto& th : gen) th.join();
  }
  std::shared_ptr<arrow::Array> key_array = std::make_shared<arrow::Int64Array>(n, std::shared_ptr<arrow::Buffer>(std::move(kbuf)));
  std::shared_ptr<arrow::Array> val_array = std::make_shared<arrow::DoubleArray>(n, std::shared_ptr<arrow::Buffer>(std::move(vbuf)));
  std::shared_ptr<arrow::Array> idx_array = std::make_shared<arrow::UInt64Array>(n, std::shared_ptr<arrow::Buffer>(std::move(ibuf)));

  // ---- GroupBy::makeGroups
  const double t0 = now();
  auto key_batch = Unwrap(cp::ExecBatch::Make(std::vector<arrow::Datum>{key_array}));
  auto grouper = Unwrap(cp::Grouper::Make(key_batch.GetTypes()));
  arrow::Datum id_batch = Unwrap(grouper->Consume(cp::ExecSpan(key_batch)));
  const double t_consume = now();
  auto groupings = Unwrap(cp::Grouper::MakeGroupings(*id_batch.array_as<arrow::UInt32Array>(), grouper->num_groups()));
  auto uniques = Unwrap(grouper->GetUniques());
  std::shared_ptr<arrow::Array> unique_keys = uniques.values[0].make_array();
  const double t_groupings = now();
  auto grouped_index = Unwrap(cp::Grouper::ApplyGroupings(*groupings, *idx_array));  // processIndex
  auto grouped_key = Unwrap(cp::Grouper::ApplyGroupings(*groupings, *key_array));    // processEach: every column, the key column too
  auto grouped_val = Unwrap(cp::Grouper::ApplyGroupings(*groupings, *val_array));
  const int64_t G = grouper->num_groups();
  std::vector<std::shared_ptr<arrow::Array>> group_vals((size_t)G);
  for (int64_t g = 0; g < G; ++g) group_vals[(size_t)g] = grouped_val->value_slice(g);  // groups[key].emplace_back(value_slice)
  const double t_apply = now();

  // ---- sum / mean / count: one scalar CallFunction per group, thread pool over the groups (tbb::parallel_for in the reference)
  std::vector<double> sums((size_t)G), means((size_t)G);
  std::vector<int64_t> counts((size_t)G);
  std::atomic<int64_t> next{0};
  std::atomic<int> failed{0};
  auto worker = [&] {
    const int64_t chunk = 256;
    for (;;) {
      int64_t b = next.fetch_add(chunk);
      if (b >= G) return;
      for (int64_t g = b; g < std::min(G, b + chunk); ++g) {
        auto s = cp::CallFunction("sum", {group_vals[(size_t)g]});
        auto m = cp::CallFunction("mean", {group_vals[(size_t)g]});
        auto c = cp::CallFunction("count", {group_vals[(size_t)g]});
        if (!s.ok() || !m.ok() || !c.ok()) {
          failed = 1;
          return;
        }
        sums[(size_t)g] = s->scalar_as<arrow::DoubleScalar>().value;
        means[(size_t)g] = m->scalar_as<arrow::DoubleScalar>().value;
        counts[(size_t)g] = c->scalar_as<arrow::Int64Scalar>().value;
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
  for (auto& th : pool) th.join();
  const double t_agg = now();
  if (failed) {
    std::fprintf(stderr, "arrow_seq: a per-group CallFunction failed\n");
    return 2;
  }

  if (!out_path.empty()) {
    FILE* f = std::fopen(out_path.c_str(), "wb");
    if (!f) {
      std::perror("arrow_seq: --out");
      return 2;
    }
    auto uk = std::static_pointer_cast<arrow::Int64Array>(unique_keys);
    std::fwrite(uk->raw_values(), 8, (size_t)G, f);
    std::fwrite(sums.data(), 8, (size_t)G, f);
    std::fwrite(means.data(), 8, (size_t)G, f);
    std::fwrite(counts.data(), 8, (size_t)G, f);
    std::fclose(f);
  }
  std::printf("{\"rows\": %lld, \"groups\": %lld, \"threads\": %d, \"arrow_version\": \"%s\", \"seconds\": %.6f, "
              "\"phases\": {\"consume\": %.6f, \"make_groupings\": %.6f, \"apply_groupings_x3\": %.6f, \"per_group_sum_mean_count\": %.6f}}\n",
              (long long)n, (long long)G, threads, ARROW_VERSION_STRING, t_agg - t0, t_consume - t0, t_groupings - t_consume,
              t_apply - t_groupings, t_agg - t_apply);
  return 0;
}
