// pdx_common.hpp -- shared plumbing for the gfx950 kernels behind include/pdx/abi.h.
// Error convention, scratch pool, bitmap helpers, launch geometry.  CDNA4 only (wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "pdx/abi.h"

namespace pdx {

// ---------------------------------------------------------------- errors (thread-local message, no exceptions)
void set_error(const std::string& msg);
int fail(int status, const std::string& msg);
int hip_fail(hipError_t e, const char* what);

#define PDX_HIP(expr)                                            \
  do {                                                           \
    hipError_t _e = (expr);                                      \
    if (_e != hipSuccess) return ::pdx::hip_fail(_e, #expr);     \
  } while (0)
#define PDX_TRY(expr)               \
  do {                              \
    int _s = (expr);                \
    if (_s != PDX_OK) return _s;    \
  } while (0)
#define PDX_LAUNCH_CHECK() PDX_HIP(hipGetLastError())

// ---------------------------------------------------------------- scratch pool
// Size-bucketed caching allocator over hipMalloc: group-by needs tens of GB of workspace per call and
// hipMalloc/hipFree of that size costs milliseconds.  Blocks are reused across calls; pdx_trim_pool frees them.
// The pool is per device and stream ordered: a block is returned together with the calling thread's CURRENT stream (the stream
// of the ABI call being served, noted by as_stream) and an event recorded on it; it is handed out again at once to that same
// stream, and to any other stream / thread only once the event has completed.
void* pool_alloc(size_t bytes);  // nullptr on failure (error set)
void pool_free(void* p);
void pool_free_many(void* const* ptrs, int n);  // one event for the whole batch
void pool_trim();
void note_stream(hipStream_t s);  // thread-local: the stream whose queued kernels may still use blocks this thread frees
hipStream_t current_stream();
void* pinned_slot();  // 64 pinned bytes of this host thread (or nullptr)
// thread-local: an orchestration inside the library (dist.hip) that drains its stream itself asks the entry points it calls to leave out
// their trailing hipStreamSynchronize (their outputs are device buffers, stream-ordered): one host wait per stage less
bool defer_sync();
void set_defer_sync(bool on);
// thread-local: the sharded orchestration only wants the records of pdx_groupby_group_values' layout (never the sorted values themselves):
// the layout may stop one sort pass early (gb_partial_tree.hpp, k_flr_emit)
bool fused_emit_wanted();
void set_fused_emit_wanted(bool on);
struct FusedEmitScope {
  bool prev;
  FusedEmitScope() : prev(fused_emit_wanted()) { set_fused_emit_wanted(true); }
  ~FusedEmitScope() { set_fused_emit_wanted(prev); }
};
struct DeferSyncScope {
  bool prev;
  DeferSyncScope() : prev(defer_sync()) { set_defer_sync(true); }
  ~DeferSyncScope() { set_defer_sync(prev); }
};

struct Scratch {  // RAII: everything allocated through it is returned to the pool on scope exit
  static constexpr int kMax = 64;
  void* ptrs[kMax];
  int n = 0;
  bool failed = false;
  ~Scratch() { release(); }
  void release() {
    pool_free_many(ptrs, n);
    n = 0;
  }
  template <typename T>
  T* get(size_t count) {
    if (n >= kMax) {
      fail(PDX_INVALID, "internal: scratch table full");
      failed = true;
      return nullptr;
    }
    void* p = pool_alloc((count ? count : 1) * sizeof(T));
    if (!p) {
      failed = true;
      return nullptr;
    }
    ptrs[n++] = p;
    return static_cast<T*>(p);
  }
};
#define PDX_SCRATCH_CHECK(s) \
  if ((s).failed) return PDX_OOM

// every entry point converts its `void* stream` through here first, which also tells the pool which stream this thread is on
inline hipStream_t as_stream(void* s) {
  hipStream_t st = static_cast<hipStream_t>(s);
  note_stream(st);
  return st;
}
// handles (pdx_groupby, pdx_grouped) outlive the call that made them: they remember their stream and restore it around frees
struct StreamNote {
  hipStream_t prev;
  explicit StreamNote(hipStream_t s) : prev(current_stream()) { note_stream(s); }
  ~StreamNote() { note_stream(prev); }
};

// ---------------------------------------------------------------- optional per-kernel timing (pdx_profile_*)
// When enabled, a HIP event pair brackets the launches inside the scope ON THE LAUNCH STREAM; bench.py reads the
// per-tag totals to report the dominant kernel's achieved bandwidth.  Disabled: one relaxed load per scope.
bool profile_enabled();
struct ProfileScope {
  int slot;
  hipStream_t st;
  ProfileScope(const char* tag, hipStream_t s);
  ~ProfileScope();
};
// Every scope opened by this thread while the object lives reports under `tag` (e.g. the sort of the received partial records: not
// the big per-row scatter passes the bench prices against the roofline)
struct ProfileTagOverride {
  const char* prev;
  explicit ProfileTagOverride(const char* tag);
  ~ProfileTagOverride();
};
#define PDX_CONCAT2(a, b) a##b
#define PDX_CONCAT(a, b) PDX_CONCAT2(a, b)
#define PDX_PROFILE(tag, st) ::pdx::ProfileScope PDX_CONCAT(_pdx_prof_scope_, __LINE__)(tag, st)

// ---------------------------------------------------------------- geometry
constexpr int kWave = 64;
constexpr int kCUs = 256;
// memory-bound grid-stride kernels: enough workgroups to fill 256 CUs x 8 blocks, never more than the work
inline int grid_for(int64_t work_items, int block, int items_per_thread = 1, int max_blocks = kCUs * 8) {
  int64_t per_block = (int64_t)block * items_per_thread;
  int64_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// ---------------------------------------------------------------- device helpers
// Arrow bitmap: bit i lives in byte i>>3, position i&7 (LSB first).
__device__ __forceinline__ bool bit_get(const uint8_t* bits, int64_t i) { return (bits[i >> 3] >> (i & 7)) & 1; }
// 64 consecutive bits starting at bit position `bitpos` (may be unaligned).  Reads up to 9 bytes; callers guarantee
// that reading bytes up to (bitpos+63)>>3 is in bounds or pass `limit_bits` to clamp.
__device__ __forceinline__ uint64_t load_bits64(const uint8_t* bits, int64_t bitpos, int64_t limit_bits) {
  // limit_bits: total number of addressable bits from bit 0 of `bits` (exclusive end); bits past it read as 0
  int64_t byte0 = bitpos >> 3;
  int sh = (int)(bitpos & 7);
  int64_t last_byte = (limit_bits + 7) >> 3;  // exclusive
  uint64_t lo = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int64_t b = byte0 + k;
    uint64_t v = (b < last_byte) ? (uint64_t)bits[b] : 0ull;
    lo |= v << (8 * k);
  }
  uint64_t res = lo >> sh;
  if (sh) {
    int64_t b = byte0 + 8;
    uint64_t v = (b < last_byte) ? (uint64_t)bits[b] : 0ull;
    res |= v << (64 - sh);
  }
  int64_t remain = limit_bits - bitpos;
  if (remain < 64) res &= (remain <= 0) ? 0ull : ((1ull << remain) - 1ull);
  return res;
}
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// device-side error flag block (one per call, in scratch): [0]=code, [1]=payload
struct ErrFlag {
  unsigned long long code;
  long long payload;
};

int check_column(const pdx_column* c, const char* what);
inline bool is_int_like(int dt) { return dt == PDX_INT64 || dt == PDX_UINT64 || dt == PDX_TIMESTAMP_NS; }
inline const uint8_t* validity_or_null(const pdx_column* c) {
  return (c->validity && c->null_count != 0) ? static_cast<const uint8_t*>(c->validity) : nullptr;
}

}  // namespace pdx
