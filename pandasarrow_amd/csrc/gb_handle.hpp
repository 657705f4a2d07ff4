// gb_handle.hpp -- part of groupby.hip: the opaque handle behind pdx_groupby (hash group-by or resample segments).
#pragma once
#include <string>

// One column's values in grouped order: what the reference's GroupBy constructor materialises per column (processEach,
// src/dataframe.cpp:1539-1554: MakeGroupings + ApplyGroupings) and every later sum() / mean() / count() reuses
// (src/group_by.h:85-139).  Built by build_layout (gb_layout.hpp) either for ONE pdx_groupby_agg call (a local object) or once
// per bound column (pdx_groupby_bind: kept in the handle, found again by the column's identity).
struct GroupedLayout {
  // identity of the column (bound layouts): Arrow buffers are immutable, the binder promises they outlive the binding
  const void* values = nullptr;
  const void* validity = nullptr;
  int64_t offset = 0;
  int dtype = 0;
  bool bound = false;
  uint64_t last_use = 0;
  size_t bytes = 0;  // device bytes the layout owns
  // (A) fused form: rows stably sorted by the LOW low_bits slot bits only; the top digit of every row in keys8 (narrowing sort) or in
  // keys_sorted (4-byte slots, bit 31 = null flag); run_start[r] = first row of the run with low bits r.  The fused last-digit
  // kernels rank by the top digit and reduce in one pass.
  bool fused = false;
  int low_bits = 0;
  int64_t nruns = 0;
  unsigned int hmax = 0;  // rows of the longest run
  const uint8_t* keys8 = nullptr;
  const uint32_t* fkeys = nullptr;
  const uint64_t* fvals = nullptr;
  const uint32_t* run_start = nullptr;
  // (A') runs longer than max_run rows (a key holding a large share of the rows drags its run along) are skipped by the fused kernels:
  // their rows live once more in a SIDE full form -- (slot, value) of those rows sorted by slot, segment starts for all G groups
  // (empty for every group outside the long runs) -- reduced by the classic kernels BEFORE the fused kernel writes its groups
  unsigned int max_run = 0;             // the limit the layout was built with (0: no run is skipped)
  int64_t side_rows = 0;
  int side_runs = 0;
  const uint64_t* side_vals = nullptr;
  const uint32_t* side_keys = nullptr;  // nullable values: bit 31 = null (as flag_keys)
  const uint32_t* side_seg = nullptr;   // G + 1
  // (B) full form: every group's values contiguous in row order (classic reducers, product / first / last, skewed keys)
  bool full = false;
  const void* vals_sorted = nullptr;
  const uint32_t* flag_keys = nullptr;  // nullable values: bit 31 of the key of every grouped row = null
  const uint32_t* seg_start = nullptr;  // G + 1
  const uint32_t* out_index = nullptr;  // segment -> group id (nullptr: segment order is group order)
  const uint8_t* row_valid = nullptr;   // segments mode: validity is read in place
  // cached per-group results of a bound column (the five standard kinds come from one reduce; later calls copy)
  double* c_sum = nullptr;
  long long* c_isum = nullptr;
  long long* c_count = nullptr;
  void* c_min = nullptr;
  void* c_max = nullptr;
  uint8_t* c_ok = nullptr;
  bool have_pw = false, have_is = false, have_mm = false, have_count = false;
  std::string plan_fused, plan_full;  // how each form was built (pdx_groupby_last_plan reports the one a call used)
  hipStream_t stream = nullptr;
  std::vector<void*> owned;
  std::vector<size_t> owned_bytes;
  template <typename T>
  T* own(size_t count) {
    const size_t b = (count ? count : 1) * sizeof(T);
    T* p = static_cast<T*>(pool_alloc(b));
    if (p) {
      owned.push_back(p);
      owned_bytes.push_back(b);
      bytes += b;
    }
    return p;
  }
  void disown(const void* p) {  // give one block back early (a sort's ping-pong buffer that does not hold the result)
    for (size_t i = 0; i < owned.size(); ++i)
      if (owned[i] == p) {
        StreamNote note(stream);
        pool_free(owned[i]);
        bytes -= owned_bytes[i];
        owned.erase(owned.begin() + (long)i);
        owned_bytes.erase(owned_bytes.begin() + (long)i);
        return;
      }
  }
  GroupedLayout() = default;
  GroupedLayout(const GroupedLayout&) = delete;
  GroupedLayout& operator=(const GroupedLayout&) = delete;
  ~GroupedLayout() {
    StreamNote note(stream);
    pool_free_many(owned.data(), (int)owned.size());
  }
};

struct pdx_groupby {
  int mode = 0;  // 0 = hash group-by, 1 = contiguous segments (resample bins; runs of equal keys when the keys arrive sorted)
  bool resample = false;  // segments are time bins (pdx_resample_create): `bin` / `label_base` are set
  int64_t n = 0, G = 0;
  int key_dtype = PDX_INT64;
  // hash mode
  uint32_t* gid_of_slot = nullptr;  // nslots entries
  int64_t nslots = 0;
  int slot_bits = 0;
  int dense = 0;                    // 1: slots are key - min (dense integer key domain), 0: open-addressing hash table
  // partitioned hash build (slot_of_row == nullptr): rows live in hash-partition order
  int part_bits = 0;                   // hash bits the rows are partitioned by (8, or 8 + digit2_bits after a second level)
  int special_slots = 0;               // 1: the null key and/or the INT64_MIN key occur (their two slots lie past the table)
  uint8_t* digit2 = nullptr;           // second-level digit of every row in FIRST-LEVEL order (very many groups only)
  uint32_t* part_off2 = nullptr;       // its scatter offsets [tiles][1 << digit2_bits]
  int digit2_bits = 0;
  uint8_t* bucket8 = nullptr;          // n, row order: low kPartBits = partition
  uint32_t* part_off = nullptr;     // [tiles][256] scatter offsets of the partition pass
  uint32_t* slot_part = nullptr;    // n, logical slot per partitioned position (non-null = the rows are hash partitioned)
  bool slot_part_ready = true;      // false: the LDS build wrote idx16_part only; ensure_slot_part (groupby.hip) fills slot_part on demand
  uint16_t* idx16_part = nullptr;   // n, slot index inside the bucket's region (LDS build, one partition level): the narrowing sort's key
  uint32_t* rows_part = nullptr;    // n, original row (bit 31: key is null)
  uint32_t* pass0_off = nullptr;    // row-order slots: scanned offsets of the first sort pass (fused into the slot kernel)
  uint32_t* slot_of_row = nullptr;  // n
  uint32_t* occ_slot = nullptr;     // G, slot order
  uint32_t* gid_of_occ = nullptr;   // G
  // both modes
  int64_t* uniques = nullptr;      // G (labels in resample mode)
  uint8_t* unique_ok = nullptr;    // G bytes
  int64_t* first_rows = nullptr;   // G
  long long* sizes = nullptr;      // G: rows per group = count of any column without nulls (gb_acc.hpp fills it on first use)
  bool sizes_ready = false;
  // segments mode
  uint32_t* seg_start = nullptr;   // G + 1
  BinParams bin{};
  long long label_base = 0;
  mutable hipStream_t stream = nullptr;  // the stream of the last call that used the handle (pool frees are ordered behind it)
  // pdx_groupby_create returns without draining its stream (G is known from an earlier read-back; the last small kernels overlap with
  // the caller's preparation of the first aggregation: ~80 us per step): calls on the SAME stream are ordered behind them by the
  // stream, calls on another stream wait for this event first
  hipStream_t create_stream = nullptr;
  hipEvent_t ready = nullptr;
  void use_on(hipStream_t st) const {
    if (ready && st != create_stream) (void)hipStreamWaitEvent(st, ready, 0);
    stream = st;
  }
  // bound columns (pdx_groupby_bind): grouped layouts + cached results, least recently used first out when over the byte limit
  std::vector<std::unique_ptr<GroupedLayout>> bound;
  size_t bind_limit = 0;   // 0 = default (a quarter of the device's memory)
  uint64_t use_clock = 0;
  std::string last_plan;   // the path the last pdx_groupby_agg took (pdx_groupby_last_plan)
  std::vector<void*> owned;
  template <typename T>
  T* own(size_t count) {
    T* p = static_cast<T*>(pool_alloc((count ? count : 1) * sizeof(T)));
    if (p) owned.push_back(p);
    return p;
  }
  ~pdx_groupby() {
    if (ready) (void)hipEventDestroy(ready);
    for (auto& b : bound) b->stream = stream;
    bound.clear();
    StreamNote note(stream);
    pool_free_many(owned.data(), (int)owned.size());
  }
};
