// arrow_order.cpp -- TEST INFRASTRUCTURE (never part of the product): the group ids arrow::compute::Grouper::Consume hands out -- the call
// GroupBy::makeGroups makes (reference src/dataframe.cpp:1580-1591) -- for a seeded int64 key column, written as raw uint32 so that
// tests/test_oracle_golden_r4.py can hold the SHAPE of Arrow's id order against the first-occurrence order this backend defines
// (include/pdx/abi.h at pdx_groupby_create).  Built like oracle/arrow_seq.cpp against the pyarrow wheel's Arrow C++.
//   arrow_order --rows N --keys K --seed S --out FILE     keys[i] = (mt19937_64(S) % K) * 1000003 - 17
#include <arrow/api.h>
#include <arrow/compute/api.h>
#include <arrow/compute/row/grouper.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

namespace cp = arrow::compute;

int main(int argc, char** argv) {
  int64_t n = 100000, nk = 100;
  uint64_t seed = 1;
  std::string out;
  for (int i = 1; i + 1 < argc; i += 2) {
    if (!std::strcmp(argv[i], "--rows")) n = (int64_t)std::atof(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--keys")) nk = (int64_t)std::atof(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--seed")) seed = (uint64_t)std::atoll(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--out")) out = argv[i + 1];
  }
  if (!cp::Initialize().ok() || out.empty()) return 2;
  std::mt19937_64 rng(seed);
  arrow::Int64Builder kb;
  if (!kb.Reserve(n).ok()) return 2;
  for (int64_t i = 0; i < n; ++i) kb.UnsafeAppend((int64_t)(rng() % (uint64_t)nk) * 1000003 - 17);
  auto keys = kb.Finish().ValueOrDie();
  auto batch = cp::ExecBatch::Make(std::vector<arrow::Datum>{keys}).ValueOrDie();
  auto grouper = cp::Grouper::Make(batch.GetTypes()).ValueOrDie();
  auto ids = grouper->Consume(cp::ExecSpan(batch)).ValueOrDie().array_as<arrow::UInt32Array>();
  FILE* f = std::fopen(out.c_str(), "wb");
  if (!f) return 2;
  std::fwrite(std::static_pointer_cast<arrow::Int64Array>(keys)->raw_values(), 8, (size_t)n, f);
  std::fwrite(ids->raw_values(), 4, (size_t)n, f);
  std::fclose(f);
  std::printf("{\"rows\": %lld, \"groups\": %u, \"arrow_version\": \"%s\"}\n", (long long)n, grouper->num_groups(), ARROW_VERSION_STRING);
  return 0;
}
