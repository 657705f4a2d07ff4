// runtime.hip -- status/error plumbing, scratch pool, host<->device helpers, synthetic input generators.
#include <string.h>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>
#include "pdx_common.hpp"

namespace pdx {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int status, const std::string& msg) {
  g_last_error = msg;
  return status;
}
int hip_fail(hipError_t e, const char* what) {
  g_last_error = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what;
  return e == hipErrorOutOfMemory ? PDX_OOM : PDX_DEVICE;
}

int check_column(const pdx_column* c, const char* what) {
  if (!c) return fail(PDX_INVALID, std::string(what) + ": null column");
  if (c->length < 0 || c->offset < 0) return fail(PDX_INVALID, std::string(what) + ": negative length/offset");
  if (c->length > 0 && !c->values) return fail(PDX_INVALID, std::string(what) + ": null values pointer");
  return PDX_OK;
}

// ---------------------------------------------------------------- pool
// One pool per device, stream-ordered reuse.  A block goes back with the stream that last used it (the calling thread's
// current stream, see note_stream) and an event recorded on that stream at the moment of the free: kernels still queued on
// the stream may be reading the block.  A later pool_alloc hands it out again
//   - at once when the requesting stream IS that stream (stream order: the new kernels run behind the old ones), or
//   - when the event has completed (hipEventQuery), i.e. every kernel that could touch the block has drained;
// otherwise the block is skipped and, if nothing else fits, a fresh one is hipMalloc'ed.  This is what makes the ABI safe to call
// from several host threads on their own streams (the reference drives this boundary from tbb::parallel_for workers,
// src/pd_core_macros.h:21,56,94,122) and from one thread that switches devices.
namespace {
struct EventCache {  // per device: an event belongs to the device that was current when it was created and can only be recorded there
  std::mutex mu;
  std::map<int, std::vector<hipEvent_t>> free_events;
  hipEvent_t get(int dev) {
    {
      std::lock_guard<std::mutex> lk(mu);
      auto& v = free_events[dev];
      if (!v.empty()) {
        hipEvent_t e = v.back();
        v.pop_back();
        return e;
      }
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    return e;
  }
  void put(int dev, hipEvent_t e) {
    std::lock_guard<std::mutex> lk(mu);
    free_events[dev].push_back(e);
  }
};
EventCache& event_cache() {
  static EventCache* c = new EventCache;  // never destroyed: blocks released during static destruction may still return events
  return *c;
}
// one event shared by every block released together (a Scratch scope, a handle): refcounted, returned to the cache by the last block
struct FreeMark {
  hipEvent_t ev = nullptr;
  int device = 0;
  ~FreeMark() {
    if (ev) event_cache().put(device, ev);
  }
};
struct FreeBlock {
  void* ptr;
  hipStream_t stream;
  std::shared_ptr<FreeMark> mark;  // null: nothing was ever queued against the block (safe for any stream)
};
struct LiveBlock {
  size_t size;
  int device;
};
struct Pool {
  std::mutex mu;
  std::map<int, std::multimap<size_t, FreeBlock>> free_blocks;  // device -> size -> block
  std::map<void*, LiveBlock> live;                              // ptr -> size, device
};
Pool& pool() {
  static Pool* p = new Pool;
  return *p;
}
size_t bucket(size_t bytes) {
  // round up to 256 B below 1 MiB, to 1/8 of the next power of two above (bounded internal fragmentation)
  if (bytes < (1u << 20)) return (bytes + 255) & ~size_t(255);
  size_t p2 = 1;
  while (p2 < bytes) p2 <<= 1;
  size_t step = p2 >> 3;
  return (bytes + step - 1) / step * step;
}
std::atomic<long> g_sync_fallbacks{0};  // pool frees that had to drain the stream because no event could be recorded (should stay 0)
thread_local hipStream_t t_stream = nullptr;  // the stream of the ABI call this thread is serving (note_stream)
bool block_ready(const FreeBlock& b, hipStream_t want) {
  if (!b.mark || !b.mark->ev || b.stream == want) return true;
  hipError_t e = hipEventQuery(b.mark->ev);
  if (e == hipSuccess) return true;
  (void)hipGetLastError();  // hipErrorNotReady is an answer, not a failure: keep it out of the next PDX_LAUNCH_CHECK
  return false;
}
}  // namespace

thread_local bool t_fused_emit = false;
bool fused_emit_wanted() { return t_fused_emit; }
void set_fused_emit_wanted(bool on) { t_fused_emit = on; }
thread_local bool t_defer_sync = false;
bool defer_sync() { return t_defer_sync; }
void set_defer_sync(bool on) { t_defer_sync = on; }
void note_stream(hipStream_t s) { t_stream = s; }
hipStream_t current_stream() { return t_stream; }

// 64 bytes of pinned, device-visible host memory per host thread (never freed): kernels write small results straight into it and
// small device-to-host copies land in it without the staging a pageable target needs.  nullptr when the allocation fails.
void* pinned_slot() {
  static thread_local void* p = nullptr;
  static thread_local bool tried = false;
  if (!tried) {
    tried = true;
    if (hipHostMalloc(&p, 64, hipHostMallocPortable) != hipSuccess) {
      (void)hipGetLastError();
      p = nullptr;
    }
  }
  return p;
}

void* pool_alloc(size_t bytes) {
  size_t sz = bucket(bytes);
  Pool& p = pool();
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    fail(PDX_DEVICE, "no current HIP device (call pdx_init first)");
    return nullptr;
  }
  {
    std::lock_guard<std::mutex> lk(p.mu);
    auto& fb = p.free_blocks[dev];
    int looked = 0;
    for (auto it = fb.lower_bound(sz); it != fb.end() && it->first <= sz + sz / 4 && looked < 16; ++it, ++looked) {
      if (!block_ready(it->second, t_stream)) continue;
      void* ptr = it->second.ptr;
      p.live[ptr] = LiveBlock{it->first, dev};
      fb.erase(it);
      return ptr;
    }
  }
  void* ptr = nullptr;
  hipError_t e = hipMalloc(&ptr, sz);
  if (e != hipSuccess) {
    // give cached blocks back and retry once
    pool_trim();
    (void)hipGetLastError();
    e = hipMalloc(&ptr, sz);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      fail(PDX_OOM, "device allocation of " + std::to_string(sz) + " bytes failed");
      return nullptr;
    }
  }
  std::lock_guard<std::mutex> lk(p.mu);
  p.live[ptr] = LiveBlock{sz, dev};
  return ptr;
}
void pool_free_many(void* const* ptrs, int n) {
  if (n <= 0) return;
  // one event for the whole batch, recorded behind everything this thread has queued on its stream so far
  std::shared_ptr<FreeMark> mark;
  int cur_dev = 0;
  if (hipGetDevice(&cur_dev) != hipSuccess) (void)hipGetLastError();
  hipEvent_t ev = event_cache().get(cur_dev);
  if (ev && hipEventRecord(ev, t_stream) == hipSuccess) {
    mark = std::make_shared<FreeMark>();
    mark->ev = ev;
    mark->device = cur_dev;
  } else {
    // no event: fall back to draining the stream so the blocks are safe for everybody
    (void)hipGetLastError();
    if (ev) event_cache().put(cur_dev, ev);
    g_sync_fallbacks.fetch_add(1, std::memory_order_relaxed);
    (void)hipStreamSynchronize(t_stream);
    (void)hipGetLastError();
  }
  Pool& p = pool();
  std::lock_guard<std::mutex> lk(p.mu);
  for (int i = 0; i < n; ++i) {
    if (!ptrs[i]) continue;
    auto it = p.live.find(ptrs[i]);
    if (it == p.live.end()) continue;
    p.free_blocks[it->second.device].emplace(it->second.size, FreeBlock{ptrs[i], t_stream, mark});
    p.live.erase(it);
  }
}
void pool_free(void* ptr) {
  if (ptr) pool_free_many(&ptr, 1);
}
void pool_trim() {
  Pool& p = pool();
  std::vector<std::pair<int, void*>> to_free;
  {
    std::lock_guard<std::mutex> lk(p.mu);
    for (auto& dv : p.free_blocks) {
      for (auto& kv : dv.second) to_free.emplace_back(dv.first, kv.second.ptr);
      dv.second.clear();
    }
  }
  // hipFree waits for the device to go idle, so blocks whose event has not completed yet are safe to release here
  int cur = 0;
  bool have_cur = hipGetDevice(&cur) == hipSuccess;
  for (auto& q : to_free) {
    if (have_cur && q.first != cur) (void)hipSetDevice(q.first);
    (void)hipFree(q.second);
    if (have_cur && q.first != cur) (void)hipSetDevice(cur);
  }
  (void)hipGetLastError();
}

// ---------------------------------------------------------------- profiling
namespace {
struct ProfRecord {
  const char* tag;
  hipEvent_t start, stop;
};
struct Profiler {
  std::mutex mu;
  bool enabled = false;
  std::vector<ProfRecord> records;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events;
};
Profiler& profiler() {
  static Profiler p;
  return p;
}
}  // namespace
bool profile_enabled() { return profiler().enabled; }
static thread_local const char* t_profile_tag_override = nullptr;
ProfileTagOverride::ProfileTagOverride(const char* tag) : prev(t_profile_tag_override) { t_profile_tag_override = tag; }
ProfileTagOverride::~ProfileTagOverride() { t_profile_tag_override = prev; }
ProfileScope::ProfileScope(const char* tag, hipStream_t s) : slot(-1), st(s) {
  Profiler& p = profiler();
  if (!p.enabled) return;
  std::lock_guard<std::mutex> lk(p.mu);
  ProfRecord r;
  r.tag = t_profile_tag_override ? t_profile_tag_override : tag;
  if (!p.free_events.empty()) {
    r.start = p.free_events.back().first;
    r.stop = p.free_events.back().second;
    p.free_events.pop_back();
  } else {
    if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
  }
  (void)hipEventRecord(r.start, st);
  slot = (int)p.records.size();
  p.records.push_back(r);
}
ProfileScope::~ProfileScope() {
  if (slot < 0) return;
  Profiler& p = profiler();
  std::lock_guard<std::mutex> lk(p.mu);
  (void)hipEventRecord(p.records[slot].stop, st);
}

// ---------------------------------------------------------------- synthetic generators
__global__ void k_synth_keys(int64_t start, int64_t n, uint64_t num_keys, int64_t* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
    uint64_t i = (uint64_t)(start + k);
    out[k] = (int64_t)(splitmix64(i ^ 0x5EED0001ull) % num_keys);
  }
}
__global__ void k_synth_vals(int64_t start, int64_t n, uint64_t seed_off, double* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
    uint64_t i = (uint64_t)(start + k);
    out[k] = (double)(splitmix64(i + 0x5EED0002ull + seed_off) >> 11) * 0x1.0p-53;
  }
}
__global__ void k_synth_ts(int64_t start, int64_t n, int64_t t0, int64_t step, int64_t* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) out[k] = t0 + (start + k) * step;
}

}  // namespace pdx

using namespace pdx;

extern "C" {

int pdx_abi_version(void) { return PDX_ABI_VERSION; }
#ifndef PDX_SOURCE_HASH
#define PDX_SOURCE_HASH "unknown"
#endif
const char* pdx_build_info(void) { return "pdx-hip abi " "1" " gfx950 sources " PDX_SOURCE_HASH; }

int pdx_init(int device) {
  int count = 0;
  PDX_HIP(hipGetDeviceCount(&count));
  if (count <= 0) return fail(PDX_DEVICE, "no HIP device visible: libpdx_hip needs an MI355X (gfx950)");
  if (device < 0 || device >= count) return fail(PDX_INVALID, "pdx_init: device index out of range");
  PDX_HIP(hipSetDevice(device));
  return PDX_OK;
}
int pdx_shutdown(void) {
  pool_trim();
  return PDX_OK;
}
const char* pdx_last_error(void) { return g_last_error.c_str(); }

int pdx_malloc(void** dptr, size_t bytes) {
  if (!dptr) return fail(PDX_INVALID, "pdx_malloc: null out pointer");
  PDX_HIP(hipMalloc(dptr, bytes ? bytes : 1));
  return PDX_OK;
}
int pdx_free(void* dptr) {
  if (dptr) PDX_HIP(hipFree(dptr));
  return PDX_OK;
}
int pdx_to_device(void* dst, const void* src, size_t bytes, void* stream) {
  if (bytes) PDX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
  PDX_HIP(hipStreamSynchronize(as_stream(stream)));
  return PDX_OK;
}
int pdx_to_host(void* dst, const void* src, size_t bytes, void* stream) {
  if (bytes) PDX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
  PDX_HIP(hipStreamSynchronize(as_stream(stream)));
  return PDX_OK;
}
int pdx_stream_synchronize(void* stream) {
  PDX_HIP(hipStreamSynchronize(as_stream(stream)));
  return PDX_OK;
}
int pdx_trim_pool(void) {
  pool_trim();
  return PDX_OK;
}

int pdx_profile_enable(int on) {
  Profiler& p = profiler();
  std::lock_guard<std::mutex> lk(p.mu);
  p.enabled = on != 0;
  return PDX_OK;
}
int pdx_profile_reset(void) {
  Profiler& p = profiler();
  std::lock_guard<std::mutex> lk(p.mu);
  for (auto& r : p.records) p.free_events.emplace_back(r.start, r.stop);
  p.records.clear();
  return PDX_OK;
}
// Writes "tag count total_ms\n" lines (aggregated per tag) into buf; synchronises the device first.
int pdx_profile_report(char* buf, size_t buf_len) {
  if (!buf || buf_len == 0) return fail(PDX_INVALID, "pdx_profile_report: null buffer");
  PDX_HIP(hipDeviceSynchronize());
  Profiler& p = profiler();
  std::lock_guard<std::mutex> lk(p.mu);
  std::map<std::string, std::pair<long, double>> agg;
  for (auto& r : p.records) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess) continue;
    auto& a = agg[r.tag];
    a.first += 1;
    a.second += ms;
  }
  std::string out;
  for (auto& kv : agg) out += kv.first + " " + std::to_string(kv.second.first) + " " + std::to_string(kv.second.second) + "\n";
  // (diagnostic: frees that could not record an event and drained their stream instead -- a lost-asynchrony signal, expected 0)
  if (const long fb = g_sync_fallbacks.load(std::memory_order_relaxed)) out += "pool_sync_fallback " + std::to_string(fb) + " 0.0\n";
  if (out.size() + 1 > buf_len) return fail(PDX_INVALID, "pdx_profile_report: buffer too small");
  memcpy(buf, out.c_str(), out.size() + 1);
  return PDX_OK;
}

int pdx_synth_keys(int64_t start, int64_t n, int64_t num_keys, int64_t* out, void* stream) {
  if (n < 0 || num_keys <= 0) return fail(PDX_INVALID, "pdx_synth_keys: bad arguments");
  if (n == 0) return PDX_OK;
  hipLaunchKernelGGL(k_synth_keys, dim3(grid_for(n, 256, 4)), dim3(256), 0, as_stream(stream), start, n, (uint64_t)num_keys, out);
  PDX_LAUNCH_CHECK();
  return PDX_OK;
}
int pdx_synth_vals(int64_t start, int64_t n, uint64_t seed_off, double* out, void* stream) {
  if (n < 0) return fail(PDX_INVALID, "pdx_synth_vals: bad arguments");
  if (n == 0) return PDX_OK;
  hipLaunchKernelGGL(k_synth_vals, dim3(grid_for(n, 256, 4)), dim3(256), 0, as_stream(stream), start, n, seed_off, out);
  PDX_LAUNCH_CHECK();
  return PDX_OK;
}
int pdx_synth_ts(int64_t start, int64_t n, int64_t t0_ns, int64_t step_ns, int64_t* out, void* stream) {
  if (n < 0) return fail(PDX_INVALID, "pdx_synth_ts: bad arguments");
  if (n == 0) return PDX_OK;
  hipLaunchKernelGGL(k_synth_ts, dim3(grid_for(n, 256, 4)), dim3(256), 0, as_stream(stream), start, n, t0_ns, step_ns, out);
  PDX_LAUNCH_CHECK();
  return PDX_OK;
}

}  // extern "C"
