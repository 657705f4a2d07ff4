// dist.hip -- the row-range sharded group-by and concat behind a C ABI: one process per GPU, RCCL
// over xGMI called DIRECTLY (librccl is opened at run time: no torch, no python in the loop), so that the reference's C++ host code
// can shard the path through the same thin shim it uses for the single-GPU calls.
//
// The reference has no distributed code; the contract is the single-process result of
//   df.group_by(key).{sum, mean, count}(col)   (src/group_by.h:85-139, src/pd_core_macros.h:5-147)
// bit for bit, with the rows split into row ranges in rank order.  Protocol (the partial-tree exchange, see include/pdx/abi.h at
// pdx_groupby_group_values): local dictionary -> all-gather(v) of the local uniques, regrouped into the global first-occurrence
// dictionary -> all-gather of the per-group row counts (every rank learns the global rank interval of its share of each group)
// -> per group the boundary-leaf fragments + aligned subtree nodes of the local share, emitted in global-id order so that they are
// already partitioned by owner -> ONE all-to-all(v) to the owners of contiguous global-id ranges -> replay through Arrow's binary
// counter -> all-gather(v) of the owners' sums.  count = the all-gathered counts, mean = sum / count.
//
// Transport: two primitives (all-gather of equal byte counts, all-to-all with per-peer byte ranges) behind a function table.  The
// built-in table is RCCL (ncclAllGather, grouped ncclSend / ncclRecv); pdx_dist_init_custom takes the caller's own (the tests run the
// same orchestration across three processes that share one GPU, where RCCL refuses duplicate devices).
#include <dlfcn.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <string>
#include <vector>
#include "compact.hpp"
#include "pairwise.hpp"
#include "pdx_common.hpp"
#include "scan.hpp"

namespace pdx {
namespace {

// ---------------------------------------------------------------- RCCL, resolved at run time
struct Rccl {
  struct Id128 {  // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
    char b[128];
  };
  typedef int (*GetUniqueId_t)(void*);
  typedef int (*CommInitRank_t)(void**, int, Id128, int);
  typedef int (*CommDestroy_t)(void*);
  typedef int (*AllGather_t)(const void*, void*, size_t, int, void*, hipStream_t);
  typedef int (*SendRecv_t)(void*, size_t, int, int, void*, hipStream_t);
  typedef int (*Group_t)(void);
  typedef const char* (*ErrStr_t)(int);
  GetUniqueId_t GetUniqueId = nullptr;
  CommInitRank_t CommInitRank = nullptr;
  CommDestroy_t CommDestroy = nullptr;
  AllGather_t AllGather = nullptr;
  SendRecv_t Send = nullptr, Recv = nullptr;
  Group_t GroupStart = nullptr, GroupEnd = nullptr;
  ErrStr_t GetErrorString = nullptr;
  std::string error;
  bool ok = false;
};
Rccl& rccl() {
  static Rccl* r = [] {
    Rccl* x = new Rccl;
    void* h = nullptr;
    // a process that already holds an RCCL (PyTorch ships one) gets that copy back by soname
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) {
      x->error = std::string("librccl.so could not be opened: ") + (dlerror() ? dlerror() : "?");
      return x;
    }
    auto sym = [&](const char* n) {
      void* p = dlsym(h, n);
      if (!p && x->error.empty()) x->error = std::string("librccl.so lacks ") + n;
      return p;
    };
    x->GetUniqueId = (Rccl::GetUniqueId_t)sym("ncclGetUniqueId");
    x->CommInitRank = (Rccl::CommInitRank_t)sym("ncclCommInitRank");
    x->CommDestroy = (Rccl::CommDestroy_t)sym("ncclCommDestroy");
    x->AllGather = (Rccl::AllGather_t)sym("ncclAllGather");
    x->Send = (Rccl::SendRecv_t)sym("ncclSend");
    x->Recv = (Rccl::SendRecv_t)sym("ncclRecv");
    x->GroupStart = (Rccl::Group_t)sym("ncclGroupStart");
    x->GroupEnd = (Rccl::Group_t)sym("ncclGroupEnd");
    x->GetErrorString = (Rccl::ErrStr_t)sym("ncclGetErrorString");
    x->ok = x->error.empty();
    return x;
  }();
  return *r;
}
int nccl_fail(int rc, const char* what) {
  Rccl& r = rccl();
  return fail(PDX_DEVICE, std::string("RCCL error in ") + what + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "?"));
}
#define PDX_NCCL(expr)                                   \
  do {                                                   \
    const int _r = (expr);                               \
    if (_r != 0) return nccl_fail(_r, #expr);            \
  } while (0)
constexpr int kNcclInt8 = 0;

struct RcclCtx {
  void* comm = nullptr;
  int world = 1, rank = 0;
  bool force = false;  // PDX_DIST_FORCE_COLLECTIVES=1: the exchange with oneself goes over the wire too (tests at world size 1)
};
int rccl_all_gather(void* vctx, const void* send, void* recv, size_t bytes, void* stream) {
  RcclCtx* c = static_cast<RcclCtx*>(vctx);
  if (bytes == 0) return PDX_OK;
  PDX_NCCL(rccl().AllGather(send, recv, bytes, kNcclInt8, c->comm, static_cast<hipStream_t>(stream)));
  return PDX_OK;
}
int rccl_all_to_all_v(void* vctx, const void* send, const size_t* send_off, const size_t* send_bytes, void* recv, const size_t* recv_off,
                      const size_t* recv_bytes, void* stream) {
  RcclCtx* c = static_cast<RcclCtx*>(vctx);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const uint8_t* s = static_cast<const uint8_t*>(send);
  uint8_t* r = static_cast<uint8_t*>(recv);
  Rccl& n = rccl();
  PDX_NCCL(n.GroupStart());
  // a Send / Recv that fails must not leave the thread's RCCL group open (every later RCCL call of the thread -- torch's own process
  // group included -- would queue into it and never launch): remember the first error, always close the group, then report
  int first_err = 0;
  const char* where = "";
  for (int p = 0; p < c->world && !first_err; ++p) {
    if (p == c->rank && !c->force) continue;
    if (send_bytes[p] && (first_err = n.Send(const_cast<uint8_t*>(s + send_off[p]), send_bytes[p], kNcclInt8, p, c->comm, st)) != 0) where = "ncclSend";
    if (!first_err && recv_bytes[p] && (first_err = n.Recv(r + recv_off[p], recv_bytes[p], kNcclInt8, p, c->comm, st)) != 0) where = "ncclRecv";
  }
  const int end_err = n.GroupEnd();
  if (first_err) return nccl_fail(first_err, where);
  if (end_err) return nccl_fail(end_err, "ncclGroupEnd");
  if (!c->force && send_bytes[c->rank])
    PDX_HIP(hipMemcpyAsync(r + recv_off[c->rank], s + send_off[c->rank], send_bytes[c->rank], hipMemcpyDeviceToDevice, st));
  return PDX_OK;
}

// ---------------------------------------------------------------- glue kernels (everything that was torch in pandasarrow_amd/dist.py)
// bit `off + i` of a validity bitmap (nullptr: all valid) as one 0 / 1 word per row: the form validity travels in
__global__ void k_bits_to_i64(const uint8_t* __restrict__ bits, int64_t off, int64_t n, int64_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = bits ? (int64_t)bit_get(bits, off + i) : 1;
}
__global__ void k_iota_i64(int64_t n, int64_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = i;
}
__global__ void k_i64_to_bits(const int64_t* __restrict__ v, int64_t n, uint8_t* __restrict__ bits) {
  const int64_t nb = (n + 7) / 8, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += stride) {
    uint8_t w = 0;
    for (int k = 0; k < 8; ++k) {
      const int64_t i = b * 8 + k;
      if (i < n && v[i]) w |= (uint8_t)(1u << k);
    }
    bits[b] = w;
  }
}
__global__ void k_add_const(int64_t* __restrict__ v, int64_t n, int64_t c) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) v[i] += c;
}
__global__ void k_map_from_ids(const uint32_t* __restrict__ gid_cat, int64_t off, int64_t n, int64_t* __restrict__ my_map) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) my_map[i] = (int64_t)gid_cat[off + i];
}
__global__ void k_gather_i64(const int64_t* __restrict__ src, const int64_t* __restrict__ idx, int64_t n, int64_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = src[idx[i]];
}
__global__ void k_scatter_i64(const int64_t* __restrict__ src, const int64_t* __restrict__ idx, int64_t n, int64_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[idx[i]] = src[i];
}
__global__ void k_scatter_iota(const int64_t* __restrict__ idx, int64_t n, int64_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[idx[i]] = i;
}
// the counts travel as 32-bit words (a rank holds < 2^31 rows): half the bytes of the one collective that sits between the sort and the emission
__global__ void k_scatter_i64_to_u32(const int64_t* __restrict__ src, const int64_t* __restrict__ idx, int64_t n, uint32_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[idx[i]] = (uint32_t)src[i];
}
// allc[W][G] -> rows of group g on lower ranks, rows of group g on all ranks
template <typename CT>
__global__ void k_count_prefix(const CT* __restrict__ allc, int W, int rank, int64_t G, int64_t* __restrict__ prefix, int64_t* __restrict__ total) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < G; g += stride) {
    int64_t p = 0, t = 0;
    for (int s = 0; s < W; ++s) {
      const int64_t c = (int64_t)allc[(int64_t)s * G + g];
      if (s < rank) p += c;
      t += c;
    }
    prefix[g] = p;
    total[g] = t;
  }
}
// first record whose key is >= bound_d * 64, d = 0..W, bound_d = G * d / W: the owners' contiguous global-id ranges (records are sorted by
// key = global id * 64 + level + 1)
__global__ void k_record_cuts(const int64_t* __restrict__ rec_key, int64_t m, int64_t G, int W, int64_t* __restrict__ cuts) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d > W) return;
  const int64_t want = (G * d / W) * 64;
  int64_t lo = 0, hi = m;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (rec_key[mid] < want) lo = mid + 1;
    else hi = mid;
  }
  cuts[d] = lo;
}
// ---- the owners' replay WITHOUT a sort of the records.  Rank p emits its records in global-id order, so what an owner receives from p is
// the records of its groups gid_lo, gid_lo + 1, ... one after the other, and how many each group has follows from the counts every rank
// holds anyway (allc[p][g] rows of group g on rank p: a = the rows on lower ranks, c = its own): the segment of (p, g) starts at the
// exclusive sum of the record counts in (p, g) order -- which is also the order of the received buffer.  (The records used to be sorted by
// group with three radix passes: 0.36 of the 3.9 ms per-rank step.)
template <typename CT>
__global__ void k_record_counts(const CT* __restrict__ allc, int W, int64_t G, int64_t gid_lo, int64_t n_own, int64_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, total = (int64_t)W * n_own;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int p = (int)(i / n_own);
    const int64_t g = gid_lo + i % n_own;
    int64_t a = 0;
    for (int q = 0; q < p; ++q) a += (int64_t)allc[(int64_t)q * G + g];
    out[i] = partial_record_count(a, (int64_t)allc[(int64_t)p * G + g]);
  }
}
// A thread per owned group walks its W segments.  VALUES ONLY travel: the codes of rank p's records of group g follow from the rows it holds -- a = rows on lower ranks, c = its
// own: the rows up to the first leaf boundary one by one, the aligned blocks of the leaves inside [a, a + c), the sum of the rows that begin
// the last leaf -- exactly the order the emitting kernels use.  Halves the bytes of the record exchange (one all-to-all instead of two).
template <typename CT>
__global__ void __launch_bounds__(256) k_replay_ranked_values(const double* __restrict__ rec_val, int64_t m, const int64_t* __restrict__ seg,
                                                              const CT* __restrict__ counts /* [W][G] */, int W, int64_t G, int64_t gid_lo,
                                                              int64_t n_own, double* __restrict__ out, unsigned int* __restrict__ bad) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_own; j += stride) {
    PairwiseCounter cn;
    cn.init();
    double acc = 0.0;
    int fill = 0;
    bool any = false;
    int64_t a = 0;
    for (int p = 0; p < W; ++p) {
      const int64_t c = (int64_t)counts[(int64_t)p * G + gid_lo + j];
      const int64_t at = (int64_t)p * n_own + j;
      int64_t i = seg[at];
      const int64_t i1 = seg[at + 1];
      if (i < 0 || i1 > m || i1 < i || i1 - i != partial_record_count(a, c)) {
        atomicExch(bad, 4u);
        break;
      }
      if (c > 0) {
        any = true;
        const int64_t b = a + c, kf = (a + 15) >> 4, kl = b >> 4;
        const int64_t raw = kf > kl ? c : 16 * kf - a;  // rows that continue a leaf begun on a lower rank
        for (int64_t q = 0; q < raw; ++q) {
          acc = pw_leaf_add(acc, rec_val[i++]);
          if (++fill == 16) {
            cn.push(acc, 0);
            acc = 0.0;
            fill = 0;
          }
        }
        if (kf <= kl) {
          if (fill != 0) atomicExch(bad, 2u);  // the nodes start on a leaf boundary
          for (int64_t sidx = kf; sidx < kl;) {
            const int lvl = (int)aligned_block_level(sidx, kl);
            cn.push(rec_val[i++], lvl);
            sidx += (int64_t)1 << lvl;
          }
          if (b > 16 * kl) {  // the first rows of the next leaf, summed in order by their rank
            acc = rec_val[i++];
            fill = (int)(b - 16 * kl);
          }
        }
      }
      a += c;
      if (p == W - 1 && j == n_own - 1 && i1 != m) atomicExch(bad, 5u);  // (the received buffer holds exactly what the counts announce)
    }
    if (fill) cn.push(acc, 0);
    if (!any) atomicExch(bad, 3u);
    out[j] = any ? cn.finish() : 0.0;
  }
}
__global__ void k_means(const double* __restrict__ sums, const int64_t* __restrict__ counts, int64_t G, double* __restrict__ means) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < G; g += stride) { const double sg = sums[g]; means[g] = sg == sg ? sg / (double)counts[g] : __longlong_as_double(__double_as_longlong(sg) | 0x0008000000000000ll); }  // (x86: a NaN dividend comes back quieted)
}
// rows of a sorted int64 axis below `edge` (<= edge when inclusive): one thread, binary search
__global__ void k_count_below(const long long* __restrict__ t, int64_t n, long long edge, int inclusive, int64_t* __restrict__ out) {
  if (blockIdx.x || threadIdx.x) return;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (inclusive ? t[mid] <= edge : t[mid] < edge) lo = mid + 1;
    else hi = mid;
  }
  *out = lo;
}
// The reduce-by-key of SURVEY 8(e) 3a for the ORDER-FREE kinds: every rank's dense per-group partials part[p][a][g] (a = 0 valid count,
// 1 rows, 2 min, 3 max, 4 int64 sum; absent arrays are never read) folded in rank order.  min keeps the first of tied values (strict <),
// max the first -- the last when the group holds a null on any rank (minmax.hpp); NaN partials (a share of NaNs only) are skipped
// unless every share is one; int64 sums wrap.
template <typename T>
__global__ void k_fold_partials(const int64_t* __restrict__ part, int W, int A, int64_t G, int a_min, int a_max, int a_sum, int a_rows, T* __restrict__ omin,
                                T* __restrict__ omax, long long* __restrict__ osum, long long* __restrict__ ocnt, int64_t* __restrict__ ok) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < G; g += stride) {
    long long cnt = 0, rows = 0;
    for (int p = 0; p < W; ++p) {
      cnt += part[((int64_t)p * A + 0) * G + g];
      if (a_rows >= 0) rows += part[((int64_t)p * A + a_rows) * G + g];
    }
    const bool has_null = a_rows >= 0 && rows > cnt;
    bool have_mn = false, have_mx = false;
    T mn = T(0), mx = T(0);
    unsigned long long sm = 0;
    for (int p = 0; p < W; ++p) {
      if (part[((int64_t)p * A + 0) * G + g] <= 0) continue;  // no valid value of the group on this rank
      if (a_sum >= 0) sm += (unsigned long long)part[((int64_t)p * A + a_sum) * G + g];
      if (a_min >= 0) {
        const T x = reinterpret_cast<const T*>(part)[((int64_t)p * A + a_min) * G + g];
        if (x == x && (!have_mn || x < mn)) { mn = x; have_mn = true; }
      }
      if (a_max >= 0) {
        const T x = reinterpret_cast<const T*>(part)[((int64_t)p * A + a_max) * G + g];
        if (x == x && (!have_mx || x > mx || (has_null && x == mx))) { mx = x; have_mx = true; }
      }
    }
    T none = T(0);
    if constexpr (__is_same(T, double)) none = __builtin_nan("");
    if (omin) omin[g] = have_mn ? mn : none;
    if (omax) omax[g] = have_mx ? mx : none;
    if (osum) osum[g] = (long long)sm;
    if (ocnt) ocnt[g] = cnt;
    if (ok) ok[g] = cnt > 0;
  }
}
__global__ void k_fill_i64(int64_t* __restrict__ v, int64_t n, int64_t x) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) v[i] = x;
}
// recv = every rank's [keys | first rows | valid words] block back to back (rank p: 3 * n_p words from 3 * off[p]) -> the three
// concatenations in rank order
constexpr int kMaxPackRanks = 64;
struct RankOffsets {
  int64_t off[kMaxPackRanks + 1];
};
__global__ void k_unpack3(const int64_t* __restrict__ recv, RankOffsets ro, int W, int64_t total, int64_t* __restrict__ a, int64_t* __restrict__ b,
                          int64_t* __restrict__ c) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int p = 0;
    while (p + 1 < W && ro.off[p + 1] <= i) ++p;
    const int64_t np = ro.off[p + 1] - ro.off[p], j = i - ro.off[p];
    const int64_t* blk = recv + 3 * ro.off[p];
    a[i] = blk[j];
    b[i] = blk[np + j];
    c[i] = blk[2 * np + j];
  }
}
// any i with t[i] > t[i + 1]?  (a shard's own sortedness, checked before the ranks exchange rows)
__global__ void k_any_descent(const long long* __restrict__ t, int64_t n, unsigned int* __restrict__ flag) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i + 1 < n; i += stride) bad = bad || t[i] > t[i + 1];
  if (bad) atomicOr(flag, 1u);
}
struct InvPred {
  const int64_t* inv;
  __device__ bool operator()(int64_t g) const { return inv[g] >= 0; }
};
struct InvEmit {
  const int64_t* inv;
  int64_t* order;
  __device__ void operator()(int64_t pos, int64_t g) const { order[pos] = inv[g]; }
};

// ---------------------------------------------------------------- in-process transport: W virtual ranks = W host threads on ONE device
// (pdx_groupby_sum_mean_count_chunked: inputs beyond 2^31 rows are cut into chunks that play the ranks of the exchange above)
struct LocalShared {
  int W = 1;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  bool aborted = false;
  std::vector<const uint8_t*> send;
  std::vector<const size_t*> soff, sbytes;
  // returns false when a rank has failed: everybody unwinds instead of waiting for it
  bool barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (aborted) return false;
    const uint64_t g = generation;
    if (++arrived == W) {
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != g || aborted; });
    }
    return !aborted;
  }
  void abort() {
    std::lock_guard<std::mutex> lk(mu);
    aborted = true;
    cv.notify_all();
  }
};
struct LocalCtx {
  LocalShared* sh;
  int rank;
};
int local_fail() { return fail(PDX_DEVICE, "chunked group-by: another chunk failed"); }
int local_all_gather(void* vctx, const void* send, void* recv, size_t bytes, void* stream) {
  LocalCtx* c = static_cast<LocalCtx*>(vctx);
  LocalShared* sh = c->sh;
  hipStream_t st = static_cast<hipStream_t>(stream);
  PDX_HIP(hipStreamSynchronize(st));  // my contribution is complete before a peer reads it
  sh->send[(size_t)c->rank] = static_cast<const uint8_t*>(send);
  if (!sh->barrier()) return local_fail();
  for (int p = 0; p < sh->W && bytes; ++p)
    PDX_HIP(hipMemcpyAsync(static_cast<uint8_t*>(recv) + (size_t)p * bytes, sh->send[(size_t)p], bytes, hipMemcpyDeviceToDevice, st));
  PDX_HIP(hipStreamSynchronize(st));
  if (!sh->barrier()) return local_fail();  // nobody reuses its send buffer before every peer has copied it
  return PDX_OK;
}
int local_all_to_all_v(void* vctx, const void* send, const size_t* send_off, const size_t* send_bytes, void* recv, const size_t* recv_off,
                       const size_t* recv_bytes, void* stream) {
  LocalCtx* c = static_cast<LocalCtx*>(vctx);
  LocalShared* sh = c->sh;
  hipStream_t st = static_cast<hipStream_t>(stream);
  PDX_HIP(hipStreamSynchronize(st));
  sh->send[(size_t)c->rank] = static_cast<const uint8_t*>(send);
  sh->soff[(size_t)c->rank] = send_off;
  sh->sbytes[(size_t)c->rank] = send_bytes;
  if (!sh->barrier()) return local_fail();
  for (int p = 0; p < sh->W; ++p) {
    const size_t b = sh->sbytes[(size_t)p][c->rank];
    if (b != recv_bytes[p]) {
      sh->abort();
      return fail(PDX_DEVICE, "chunked group-by: internal: send and receive counts disagree");
    }
    if (b) PDX_HIP(hipMemcpyAsync(static_cast<uint8_t*>(recv) + recv_off[p], sh->send[(size_t)p] + sh->soff[(size_t)p][c->rank], b, hipMemcpyDeviceToDevice, st));
  }
  PDX_HIP(hipStreamSynchronize(st));
  if (!sh->barrier()) return local_fail();
  return PDX_OK;
}

}  // namespace
}  // namespace pdx

using namespace pdx;

struct pdx_dist {
  int world = 1, rank = 0;
  pdx_dist_transport tr{};
  std::unique_ptr<RcclCtx> rccl_ctx;  // built-in transport
  bool force = false;
  // a second stream of this communicator: the local value sort of a sharded group-by runs on it while the dictionary exchange (collectives
  // and the regrouping of W dictionaries) occupies the caller's stream -- neither needs the other's result.  Created on first use.
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool side_ok() {
    if (side) return true;
    if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      if (ev_fork) (void)hipEventDestroy(ev_fork);
      if (ev_join) (void)hipEventDestroy(ev_join);
      if (side) (void)hipStreamDestroy(side);
      side = nullptr;
      ev_fork = ev_join = nullptr;
      return false;
    }
    return true;
  }
  ~pdx_dist() {
    if (side) {
      (void)hipStreamSynchronize(side);
      (void)hipEventDestroy(ev_fork);
      (void)hipEventDestroy(ev_join);
      (void)hipStreamDestroy(side);
    }
    if (rccl_ctx && rccl_ctx->comm && rccl().CommDestroy) (void)rccl().CommDestroy(rccl_ctx->comm);
  }
};

struct pdx_dist_groupby {
  int64_t G = 0, records = 0;
  int key_dtype = PDX_INT64;
  int64_t* keys = nullptr;
  int64_t* keys_ok = nullptr;  // 0 / 1 per key (the null key)
  int64_t* first_rows = nullptr;
  double* sums = nullptr;
  double* means = nullptr;
  int64_t* counts = nullptr;
  hipStream_t stream = nullptr;
  std::vector<void*> owned;
  template <typename T>
  T* own(size_t count) {
    T* p = static_cast<T*>(pool_alloc((count ? count : 1) * sizeof(T)));
    if (p) owned.push_back(p);
    return p;
  }
  ~pdx_dist_groupby() {
    StreamNote note(stream);
    pool_free_many(owned.data(), (int)owned.size());
  }
};

struct pdx_dist_agg {  // pdx_dist_groupby_order_free: the global dictionary + one column per requested kind
  int64_t G = 0;
  int key_dtype = PDX_INT64;
  int64_t *keys = nullptr, *keys_ok = nullptr, *first_rows = nullptr;
  std::vector<int> dtypes;
  std::vector<uint64_t*> vals;
  std::vector<int64_t*> oks;  // 0 / 1 per group (values with nulls), or nullptr
  hipStream_t stream = nullptr;
  std::vector<void*> owned;
  template <typename T>
  T* own(size_t count) {
    T* p = static_cast<T*>(pool_alloc((count ? count : 1) * sizeof(T)));
    if (p) owned.push_back(p);
    return p;
  }
  ~pdx_dist_agg() {
    StreamNote note(stream);
    pool_free_many(owned.data(), (int)owned.size());
  }
};

struct pdx_dist_resampled {
  int64_t G = 0;
  int nk = 0;
  int64_t* labels = nullptr;
  std::vector<int> dtypes;
  std::vector<uint64_t*> vals;
  std::vector<int64_t*> oks;  // 0 / 1 per bin, or nullptr when no shard produced a null
  hipStream_t stream = nullptr;
  std::vector<void*> owned;
  template <typename T>
  T* own(size_t count) {
    T* p = static_cast<T*>(pool_alloc((count ? count : 1) * sizeof(T)));
    if (p) owned.push_back(p);
    return p;
  }
  ~pdx_dist_resampled() {
    StreamNote note(stream);
    pool_free_many(owned.data(), (int)owned.size());
  }
};

namespace pdx {
namespace {

// k int64 values per rank -> [W][k] on the host (one small all-gather + one sync)
int gather_host(pdx_dist* d, const int64_t* mine, int k, std::vector<int64_t>* all, Scratch& s, hipStream_t st) {
  all->assign((size_t)d->world * k, 0);
  if (d->world == 1 && !d->force) {
    for (int i = 0; i < k; ++i) (*all)[(size_t)i] = mine[i];
    return PDX_OK;
  }
  int64_t* dsend = s.get<int64_t>((size_t)k);
  int64_t* drecv = s.get<int64_t>((size_t)k * d->world);
  PDX_SCRATCH_CHECK(s);
  // (small payloads go through this thread's pinned slot: the copy is ordered on the stream and needs no host wait of its own -- the
  //  one synchronisation of this function is the read-back below)
  void* pin = (size_t)k * sizeof(int64_t) <= 64 ? pinned_slot() : nullptr;
  if (pin) {
    memcpy(pin, mine, sizeof(int64_t) * k);
    PDX_HIP(hipMemcpyAsync(dsend, pin, sizeof(int64_t) * k, hipMemcpyHostToDevice, st));
  } else {
    PDX_HIP(hipMemcpyAsync(dsend, mine, sizeof(int64_t) * k, hipMemcpyHostToDevice, st));
    PDX_HIP(hipStreamSynchronize(st));  // `mine` is pageable host memory
  }
  PDX_TRY(d->tr.all_gather(d->tr.ctx, dsend, drecv, sizeof(int64_t) * k, st));
  PDX_HIP(hipMemcpyAsync(all->data(), drecv, sizeof(int64_t) * k * d->world, hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}
// the same for k int64 values that already live on the device: [W][k] on the host with ONE synchronisation
int gather_device(pdx_dist* d, const int64_t* dmine, int k, std::vector<int64_t>* all, Scratch& s, hipStream_t st) {
  all->assign((size_t)d->world * k, 0);
  if (d->world == 1 && !d->force) {
    PDX_HIP(hipMemcpyAsync(all->data(), dmine, sizeof(int64_t) * k, hipMemcpyDeviceToHost, st));
    PDX_HIP(hipStreamSynchronize(st));
    return PDX_OK;
  }
  int64_t* drecv = s.get<int64_t>((size_t)k * d->world);
  PDX_SCRATCH_CHECK(s);
  PDX_TRY(d->tr.all_gather(d->tr.ctx, dmine, drecv, sizeof(int64_t) * k, st));
  PDX_HIP(hipMemcpyAsync(all->data(), drecv, sizeof(int64_t) * k * d->world, hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}
// concatenation of every rank's `mine` (sizes[r] elements of `elem` bytes) in rank order: the all-gather(v) of SURVEY 8e
int all_gather_v(pdx_dist* d, const void* mine, const std::vector<int64_t>& sizes, size_t elem, void* out, hipStream_t st) {
  const int W = d->world;
  if (W == 1 && !d->force) {
    if (sizes[0]) PDX_HIP(hipMemcpyAsync(out, mine, (size_t)sizes[0] * elem, hipMemcpyDeviceToDevice, st));
    return PDX_OK;
  }
  std::vector<size_t> so((size_t)W, 0), sb((size_t)W, (size_t)sizes[(size_t)d->rank] * elem), ro((size_t)W), rb((size_t)W);
  size_t at = 0;
  for (int p = 0; p < W; ++p) {
    ro[(size_t)p] = at;
    rb[(size_t)p] = (size_t)sizes[(size_t)p] * elem;
    at += rb[(size_t)p];
  }
  return d->tr.all_to_all_v(d->tr.ctx, mine, so.data(), sb.data(), out, ro.data(), rb.data(), st);
}


// PDX_DIST_TIMING=1 (diagnostic): host time between the stages of the sharded group-by, printed per call on stderr
struct StageTimer {
  bool on;
  std::chrono::steady_clock::time_point t0;
  std::string line;
  StageTimer() : on([] { const char* e = getenv("PDX_DIST_TIMING"); return e && e[0] == '1'; }()), t0(std::chrono::steady_clock::now()) {}
  void mark(const char* what) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    line += std::string(what) + "=" + std::to_string((int)std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count()) + "us ";
    t0 = t1;
  }
  ~StageTimer() {
    if (on) fprintf(stderr, "[pdx_dist] %s\n", line.c_str());
  }
};

// A rank whose shard fails a local precondition (nulls in its values, an unsorted stretch, a create that fails) must not return while
// its peers wait for it inside the next collective: the local status travels with the first small all-gather of the call, and every
// rank returns an error before any further collective.  The failing rank keeps its own message; the others name the rank.
int gate_status(const std::vector<int64_t>& info, int words, int status_word, int W, int r, int local_rc, const char* what) {
  for (int p = 0; p < W; ++p) {
    const int64_t code = info[(size_t)p * words + status_word];
    if (code == PDX_OK) continue;
    if (p == r && local_rc != PDX_OK) return local_rc;  // (thread-local message of the failing call is still in place)
    return fail((int)code, std::string(what) + ": rank " + std::to_string(p) + " failed on its shard (status " + std::to_string(code) +
                               "); no rank entered the exchange");
  }
  return PDX_OK;
}

// Steps 1-2 of every sharded group-by: local dictionary -> all-gather(v) of (key, first row, valid) in rank order -> regrouped keeping the
// first occurrence = the single-process dictionary.  own(count) allocates int64 arrays that live as long as the result.
struct GlobalDict {
  pdx_groupby *gb = nullptr, *gb_cat = nullptr;
  int64_t Gl = 0, G = 0;
  int64_t* my_map = nullptr;  // local group id -> global group id (scratch)
  int64_t *keys = nullptr, *keys_ok = nullptr, *first_rows = nullptr;  // G, result lifetime
  ~GlobalDict() {
    if (gb_cat) pdx_groupby_destroy(gb_cat);
    if (gb) pdx_groupby_destroy(gb);
  }
};
// after_create (may be empty): called once the local handle exists and before the first collective -- work that needs only the local handle
// (the value sort) is started there on the communicator's side stream; its status joins the gate like the create's own
template <typename Own>
int build_dictionary(pdx_dist* d, const pdx_column* keys, int local_rc, const char* what, int64_t row_offset, Scratch& s, hipStream_t st, Own&& own,
                     GlobalDict* D, const std::function<int(pdx_groupby*)>& after_create = {}) {
  const int W = d->world, r = d->rank;
  const bool solo = W == 1 && !d->force;
  // ---- 1. local dictionary (a failure here is reported through the gate below, not by leaving)
  StageTimer tm;
  if (local_rc == PDX_OK) local_rc = pdx_groupby_create(keys, st, &D->gb);
  if (local_rc == PDX_OK && after_create) local_rc = after_create(D->gb);
  tm.mark("local_create");
  const int64_t Gl = local_rc == PDX_OK ? pdx_groupby_num_groups(D->gb) : 0;
  D->Gl = Gl;
  int64_t mine[2] = {Gl, local_rc};
  std::vector<int64_t> info;
  const int grc = gather_host(d, mine, 2, &info, s, st);
  if (grc != PDX_OK) return local_rc != PDX_OK ? local_rc : grc;
  PDX_TRY(gate_status(info, 2, 1, W, r, local_rc, what));
  tm.mark("gate");
  std::vector<int64_t> sizes((size_t)W);
  for (int p = 0; p < W; ++p) sizes[(size_t)p] = info[(size_t)p * 2];
  // the three dictionary columns of this rank side by side -- [keys | first rows | valid words] -- so that they travel in ONE exchange
  int64_t* pack = s.get<int64_t>((size_t)3 * (size_t)std::max<int64_t>(Gl, 1));
  uint8_t* uk_bits = s.get<uint8_t>((size_t)(Gl + 7) / 8 + 16);
  PDX_SCRATCH_CHECK(s);
  int64_t *uk = pack, *fr = pack + Gl, *uok = pack + 2 * Gl;
  {
    pdx_mut_column m{};
    m.dtype = keys->dtype;
    m.length = Gl;
    m.values = uk;
    m.validity = uk_bits;
    PDX_TRY(pdx_groupby_unique_keys(D->gb, &m, st));
    PDX_TRY(pdx_groupby_first_rows(D->gb, fr, st));
    if (Gl) {
      hipLaunchKernelGGL(k_bits_to_i64, dim3(grid_for(Gl, 256)), dim3(256), 0, st, uk_bits, (int64_t)0, Gl, uok);
      hipLaunchKernelGGL(k_add_const, dim3(grid_for(Gl, 256)), dim3(256), 0, st, fr, Gl, row_offset);
    }
    PDX_LAUNCH_CHECK();
  }
  // ---- 2. global dictionary
  int64_t total_u = 0, off = 0;
  for (int p = 0; p < W; ++p) {
    if (p < r) off += sizes[(size_t)p];
    total_u += sizes[(size_t)p];
  }
  int64_t G = Gl;
  D->my_map = s.get<int64_t>((size_t)Gl);
  PDX_SCRATCH_CHECK(s);
  if (solo) {
    D->keys = own((size_t)G);
    D->keys_ok = own((size_t)G);
    D->first_rows = own((size_t)G);
    if (!D->keys || !D->keys_ok || !D->first_rows) return PDX_OOM;
    if (G) {
      PDX_HIP(hipMemcpyAsync(D->keys, uk, (size_t)G * 8, hipMemcpyDeviceToDevice, st));
      PDX_HIP(hipMemcpyAsync(D->keys_ok, uok, (size_t)G * 8, hipMemcpyDeviceToDevice, st));
      PDX_HIP(hipMemcpyAsync(D->first_rows, fr, (size_t)G * 8, hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(k_iota_i64, dim3(grid_for(G, 256)), dim3(256), 0, st, G, D->my_map);  // one rank: local ids ARE the global ids
    }
    PDX_LAUNCH_CHECK();
  } else {
    int64_t* cat_keys = s.get<int64_t>((size_t)total_u);
    int64_t* cat_first = s.get<int64_t>((size_t)total_u);
    int64_t* cat_ok = s.get<int64_t>((size_t)total_u);
    uint8_t* cat_bits = s.get<uint8_t>((size_t)(total_u + 7) / 8 + 16);
    uint32_t* gid_cat = s.get<uint32_t>((size_t)total_u);
    PDX_SCRATCH_CHECK(s);
    if (W <= kMaxPackRanks) {  // one all-gather(v) of 24 bytes per local group, unpacked into the three concatenations
      int64_t* recv = s.get<int64_t>((size_t)3 * (size_t)std::max<int64_t>(total_u, 1));
      PDX_SCRATCH_CHECK(s);
      PDX_TRY(all_gather_v(d, pack, sizes, 24, recv, st));
      RankOffsets ro{};
      for (int p = 0; p <= W; ++p) ro.off[p] = p ? ro.off[p - 1] + sizes[(size_t)p - 1] : 0;
      if (total_u) hipLaunchKernelGGL(k_unpack3, dim3(grid_for(total_u, 256)), dim3(256), 0, st, recv, ro, W, total_u, cat_keys, cat_first, cat_ok);
      PDX_LAUNCH_CHECK();
    } else {
      PDX_TRY(all_gather_v(d, uk, sizes, 8, cat_keys, st));
      PDX_TRY(all_gather_v(d, fr, sizes, 8, cat_first, st));
      PDX_TRY(all_gather_v(d, uok, sizes, 8, cat_ok, st));
    }
    if (total_u) hipLaunchKernelGGL(k_i64_to_bits, dim3(grid_for((total_u + 7) / 8, 256)), dim3(256), 0, st, cat_ok, total_u, cat_bits);
    PDX_LAUNCH_CHECK();
    pdx_column cc{};
    cc.dtype = keys->dtype;
    cc.length = total_u;
    cc.null_count = -1;
    cc.validity = cat_bits;
    cc.values = cat_keys;
    tm.mark("dict_exchange");
    PDX_TRY(pdx_groupby_create(&cc, st, &D->gb_cat));
    tm.mark("cat_create");
    G = pdx_groupby_num_groups(D->gb_cat);
    D->keys = own((size_t)G);
    D->keys_ok = own((size_t)G);
    D->first_rows = own((size_t)G);
    uint8_t* gk_bits = s.get<uint8_t>((size_t)(G + 7) / 8 + 16);
    int64_t* cat_first_rows = s.get<int64_t>((size_t)G);
    if (!D->keys || !D->keys_ok || !D->first_rows) return PDX_OOM;
    PDX_SCRATCH_CHECK(s);
    pdx_mut_column gm{};
    gm.dtype = keys->dtype;
    gm.length = G;
    gm.values = D->keys;
    gm.validity = gk_bits;
    PDX_TRY(pdx_groupby_unique_keys(D->gb_cat, &gm, st));
    PDX_TRY(pdx_groupby_first_rows(D->gb_cat, cat_first_rows, st));
    if (total_u) PDX_TRY(pdx_groupby_group_ids(D->gb_cat, gid_cat, st));
    if (G) {
      hipLaunchKernelGGL(k_bits_to_i64, dim3(grid_for(G, 256)), dim3(256), 0, st, gk_bits, (int64_t)0, G, D->keys_ok);
      hipLaunchKernelGGL(k_gather_i64, dim3(grid_for(G, 256)), dim3(256), 0, st, cat_first, cat_first_rows, G, D->first_rows);
    }
    if (Gl) hipLaunchKernelGGL(k_map_from_ids, dim3(grid_for(Gl, 256)), dim3(256), 0, st, gid_cat, off, Gl, D->my_map);
    PDX_LAUNCH_CHECK();
    tm.mark("cat_fetch");
  }
  D->G = G;
  return PDX_OK;
}

}  // namespace
}  // namespace pdx

extern "C" {

int pdx_dist_unique_id(void* out_id128) {
  if (!out_id128) return fail(PDX_INVALID, "pdx_dist_unique_id: null output");
  Rccl& r = rccl();
  if (!r.ok) return fail(PDX_DEVICE, "pdx_dist_unique_id: " + r.error);
  PDX_NCCL(r.GetUniqueId(out_id128));
  return PDX_OK;
}

int pdx_dist_init(const void* id128, int world, int rank, pdx_dist** out) {
  if (!out || !id128) return fail(PDX_INVALID, "pdx_dist_init: null argument");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return fail(PDX_INVALID, "pdx_dist_init: rank outside the world");
  Rccl& r = rccl();
  if (!r.ok) return fail(PDX_DEVICE, "pdx_dist_init: " + r.error);
  std::unique_ptr<pdx_dist> d(new pdx_dist());
  d->world = world;
  d->rank = rank;
  d->force = [] { const char* e = getenv("PDX_DIST_FORCE_COLLECTIVES"); return e && e[0] == '1'; }();
  d->rccl_ctx.reset(new RcclCtx());
  d->rccl_ctx->world = world;
  d->rccl_ctx->rank = rank;
  d->rccl_ctx->force = d->force;
  Rccl::Id128 id;
  memcpy(id.b, id128, 128);
  PDX_NCCL(r.CommInitRank(&d->rccl_ctx->comm, world, id, rank));  // on the calling thread's current device (pdx_init)
  d->tr.ctx = d->rccl_ctx.get();
  d->tr.all_gather = rccl_all_gather;
  d->tr.all_to_all_v = rccl_all_to_all_v;
  *out = d.release();
  return PDX_OK;
}

int pdx_dist_init_custom(const pdx_dist_transport* transport, int world, int rank, pdx_dist** out) {
  if (!out || !transport || !transport->all_gather || !transport->all_to_all_v) return fail(PDX_INVALID, "pdx_dist_init_custom: null argument");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return fail(PDX_INVALID, "pdx_dist_init_custom: rank outside the world");
  std::unique_ptr<pdx_dist> d(new pdx_dist());
  d->world = world;
  d->rank = rank;
  d->force = [] { const char* e = getenv("PDX_DIST_FORCE_COLLECTIVES"); return e && e[0] == '1'; }();
  d->tr = *transport;
  *out = d.release();
  return PDX_OK;
}
int pdx_dist_destroy(pdx_dist* d) {
  delete d;
  return PDX_OK;
}
int pdx_dist_world(const pdx_dist* d) { return d ? d->world : -1; }
int pdx_dist_rank(const pdx_dist* d) { return d ? d->rank : -1; }

int pdx_dist_groupby_sum_mean_count(pdx_dist* d, const pdx_column* keys, const pdx_column* values, int64_t row_offset, void* stream,
                                    pdx_dist_groupby** out) {
  if (!d || !out) return fail(PDX_INVALID, "pdx_dist_groupby_sum_mean_count: null argument");
  *out = nullptr;
  // local preconditions: collected, not returned -- they travel with the dictionary's first all-gather (build_dictionary)
  int lrc = check_column(keys, "pdx_dist_groupby_sum_mean_count");
  if (lrc == PDX_OK) lrc = check_column(values, "pdx_dist_groupby_sum_mean_count");
  if (lrc == PDX_OK && (values->dtype != PDX_FLOAT64 || validity_or_null(values)))
    lrc = fail(PDX_NOT_IMPLEMENTED, "pdx_dist_groupby_sum_mean_count: float64 values without nulls (the partial-tree exchange); other kinds / columns: pdx_dist_groupby_order_free, or route rows");
  if (lrc == PDX_OK && values->length != keys->length) lrc = fail(PDX_INVALID, "pdx_dist_groupby_sum_mean_count: keys and values differ in length");
  hipStream_t st = as_stream(stream);
  const int W = d->world, r = d->rank;
  const bool solo = W == 1 && !d->force;
  Scratch s;
  std::unique_ptr<pdx_dist_groupby> res(new pdx_dist_groupby());
  res->stream = st;
  res->key_dtype = lrc == PDX_OK ? keys->dtype : PDX_INT64;
  GlobalDict D;
  struct Handles {  // RAII for the intermediate handles (declared after D: the grouped values go before the handle they came from)
    pdx_grouped* gv = nullptr;
    ~Handles() {
      if (gv) pdx_grouped_destroy(gv);
    }
  } h;
  // the local value sort starts on the side stream as soon as the local handle exists; the caller's stream joins it before the counts
  // (PDX_DIST_OVERLAP=0: everything on the caller's stream, one stage after the other)
  static const bool overlap_on = [] { const char* e = getenv("PDX_DIST_OVERLAP"); return !(e && e[0] == '0'); }();
  struct SideJoin {  // declared after D and h: runs before their destructors, so nothing they free is still read on the side stream
    pdx_dist* d;
    hipStream_t st;
    bool pending = false;
    void join() {
      if (pending) (void)hipStreamWaitEvent(st, d->ev_join, 0);
      pending = false;
    }
    ~SideJoin() { join(); }
  } side_join{d, st};
  std::function<int(pdx_groupby*)> start_sort;
  if (!solo && overlap_on && d->side_ok())
    start_sort = [&](pdx_groupby* gb) -> int {
      if (hipEventRecord(d->ev_fork, st) != hipSuccess || hipStreamWaitEvent(d->side, d->ev_fork, 0) != hipSuccess) {
        (void)hipGetLastError();
        return PDX_OK;  // (no fork: the sort runs on the caller's stream below)
      }
      int rc;
      {
        StreamNote on_side(d->side);  // (this thread's frees inside the call are ordered behind the side stream; restored on leaving)
        DeferSyncScope no_wait;
        FusedEmitScope records_only;  // (two sort passes: the last digit is left to the kernel that emits the records)
        rc = pdx_groupby_group_values(gb, values, d->side, &h.gv);
      }
      note_stream(st);
      if (hipEventRecord(d->ev_join, d->side) == hipSuccess) side_join.pending = true;
      else {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(d->side);
      }
      return rc;
    };
  PDX_TRY(build_dictionary(d, keys, lrc, "pdx_dist_groupby_sum_mean_count", row_offset, s, st, [&](size_t c) { return res->own<int64_t>(c); }, &D, start_sort));
  res->keys = D.keys;
  res->keys_ok = D.keys_ok;
  res->first_rows = D.first_rows;
  const int64_t Gl = D.Gl, G = D.G;
  int64_t* my_map = D.my_map;
  res->G = G;
  res->sums = res->own<double>((size_t)G);
  res->means = res->own<double>((size_t)G);
  res->counts = res->own<int64_t>((size_t)G);
  if (!res->sums || !res->means || !res->counts) return PDX_OOM;
  // ---- 3. grouped values + rows per local group
  StageTimer tm;
  DeferSyncScope no_stage_waits;  // (the stages below hand device buffers on; the host waits where it reads a size back, and at the end)
  if (h.gv) {
    side_join.join();  // sorted on the side stream meanwhile
  } else {
    FusedEmitScope records_only;
    PDX_TRY(pdx_groupby_group_values(D.gb, values, st, &h.gv));
  }
  tm.mark("group_values");
  int64_t* cnt_local = s.get<int64_t>((size_t)Gl);
  int64_t* prefix_local = s.get<int64_t>((size_t)Gl);
  PDX_SCRATCH_CHECK(s);
  PDX_TRY(pdx_grouped_counts(h.gv, cnt_local, st));
  int64_t* order = nullptr;
  const uint32_t* counts_by_rank = nullptr;  // [W][G] rows of every group on every rank (null on one rank: the local counts, local id = global id)
  if (solo) {
    if (G) {
      PDX_HIP(hipMemcpyAsync(res->counts, cnt_local, (size_t)G * 8, hipMemcpyDeviceToDevice, st));
      PDX_HIP(hipMemsetAsync(prefix_local, 0, (size_t)G * 8, st));
    }
  } else {
    // ---- 4. rows per (global group, rank): dense count vectors, all-gathered; prefix over the lower ranks
    uint32_t* dense = s.get<uint32_t>((size_t)G);
    uint32_t* allc = s.get<uint32_t>((size_t)G * W);
    int64_t* prefix_g = s.get<int64_t>((size_t)G);
    int64_t* inv = s.get<int64_t>((size_t)G);
    order = s.get<int64_t>((size_t)Gl);
    PDX_SCRATCH_CHECK(s);
    if (G) {
      PDX_HIP(hipMemsetAsync(dense, 0, (size_t)G * 4, st));
      PDX_HIP(hipMemsetAsync(inv, 0xFF, (size_t)G * 8, st));
    }
    if (Gl) {
      hipLaunchKernelGGL(k_scatter_i64_to_u32, dim3(grid_for(Gl, 256)), dim3(256), 0, st, cnt_local, my_map, Gl, dense);
      hipLaunchKernelGGL(k_scatter_iota, dim3(grid_for(Gl, 256)), dim3(256), 0, st, my_map, Gl, inv);
    }
    PDX_LAUNCH_CHECK();
    PDX_TRY(d->tr.all_gather(d->tr.ctx, dense, allc, (size_t)G * 4, st));
    counts_by_rank = allc;
    if (G) hipLaunchKernelGGL((k_count_prefix<uint32_t>), dim3(grid_for(G, 256)), dim3(256), 0, st, allc, W, r, G, prefix_g, res->counts);
    if (Gl) hipLaunchKernelGGL(k_gather_i64, dim3(grid_for(Gl, 256)), dim3(256), 0, st, prefix_g, my_map, Gl, prefix_local);
    PDX_LAUNCH_CHECK();
    // records are emitted group by group in GLOBAL-id order, so they leave the kernel already partitioned by owner rank:
    // order = the local group ids sorted by their global id = an ordered compaction of the inverse map
    // (every local group has exactly one global id -- my_map is a restriction of the dictionary's own ids -- so the compaction emits Gl
    //  entries; the count is not read back: no host wait here)
    PDX_TRY(compact_indices(G, InvPred{inv}, InvEmit{inv, order}, nullptr, s, st));
  }
  // ---- 5. partial records of the local share of every group
  tm.mark("counts_exchange");
  int64_t nrec = 0;
  PDX_TRY(pdx_grouped_partial_plan(h.gv, prefix_local, order, &nrec, st));
  tm.mark("partial_plan");
  // values only (PDX_DIST_REPLAY_SORT=1, the sorting replay, needs the keys as well)
  static const bool ranked_replay = [] { const char* e = getenv("PDX_DIST_REPLAY_SORT"); return !(e && e[0] == '1'); }();
  int64_t* rec_key = ranked_replay ? nullptr : s.get<int64_t>((size_t)nrec);
  double* rec_val = s.get<double>((size_t)nrec);
  PDX_SCRATCH_CHECK(s);
  PDX_TRY(pdx_grouped_partial_fill(h.gv, my_map, rec_key, rec_val, st));
  res->records = nrec;
  tm.mark("partial_fill");
  // ---- 6. one all-to-all(v) to the owners of contiguous global-id ranges
  std::vector<int64_t> bounds((size_t)W + 1);
  for (int p = 0; p <= W; ++p) bounds[(size_t)p] = G * p / W;
  const int64_t n_own = bounds[(size_t)r + 1] - bounds[(size_t)r];
  int64_t* rk = rec_key;
  double* rv = rec_val;
  int64_t m = nrec;
  if (!solo) {
    int64_t* dcuts = s.get<int64_t>((size_t)W + 1);
    PDX_SCRATCH_CHECK(s);
    if (rec_key) {
      hipLaunchKernelGGL(k_record_cuts, dim3((unsigned)ceil_div(W + 1, 64)), dim3(64), 0, st, rec_key, nrec, G, W, dcuts);
      PDX_LAUNCH_CHECK();
    } else {
      PDX_TRY(pdx_grouped_record_cuts(h.gv, my_map, G, W, dcuts, st));
    }
    // every rank's cut points (what I send and what I receive) straight from the device buffers: one all-gather, one host wait
    std::vector<int64_t> all_cuts;
    PDX_TRY(gather_device(d, dcuts, W + 1, &all_cuts, s, st));
    std::vector<int64_t> my_cuts(all_cuts.begin() + (size_t)r * (W + 1), all_cuts.begin() + (size_t)(r + 1) * (W + 1));
    std::vector<size_t> so((size_t)W), sb((size_t)W), ro((size_t)W), rb((size_t)W);
    size_t at = 0;
    for (int p = 0; p < W; ++p) {
      so[(size_t)p] = (size_t)my_cuts[(size_t)p] * 8;
      sb[(size_t)p] = (size_t)(my_cuts[(size_t)p + 1] - my_cuts[(size_t)p]) * 8;
      const int64_t* row = &all_cuts[(size_t)p * (W + 1)];
      ro[(size_t)p] = at;
      rb[(size_t)p] = (size_t)(row[r + 1] - row[r]) * 8;
      at += rb[(size_t)p];
    }
    m = (int64_t)(at / 8);
    rk = rec_key ? s.get<int64_t>((size_t)m) : nullptr;
    rv = s.get<double>((size_t)m);
    PDX_SCRATCH_CHECK(s);
    if (rec_key) PDX_TRY(d->tr.all_to_all_v(d->tr.ctx, rec_key, so.data(), sb.data(), rk, ro.data(), rb.data(), st));
    PDX_TRY(d->tr.all_to_all_v(d->tr.ctx, rec_val, so.data(), sb.data(), rv, ro.data(), rb.data(), st));
  }
  // ---- 7. owners replay their groups' records in (source rank, emission) order; 8. all-gather(v) of the sums
  tm.mark("record_exchange");
  double* sums_own = solo ? res->sums : s.get<double>((size_t)n_own);
  PDX_SCRATCH_CHECK(s);
  if (!ranked_replay) {
    PDX_TRY(pdx_replay_partials(rk, rv, m, bounds[(size_t)r], n_own, sums_own, st));  // (sorts the records by group first)
  } else if (n_own > 0) {
    const int64_t cells = (int64_t)W * n_own;
    int64_t* seg = s.get<int64_t>((size_t)cells + 1);
    int64_t* seg_total = s.get<int64_t>(1);
    unsigned int* bad = s.get<unsigned int>(1);
    PDX_SCRATCH_CHECK(s);
    PDX_HIP(hipMemsetAsync(bad, 0, sizeof(unsigned int), st));
    if (counts_by_rank)
      hipLaunchKernelGGL((k_record_counts<uint32_t>), dim3(grid_for(cells, 256)), dim3(256), 0, st, counts_by_rank, W, G, bounds[(size_t)r], n_own, seg);
    else
      hipLaunchKernelGGL((k_record_counts<int64_t>), dim3(grid_for(cells, 256)), dim3(256), 0, st, cnt_local, W, G, bounds[(size_t)r], n_own, seg);
    PDX_LAUNCH_CHECK();
    PDX_TRY((device_exclusive_scan<int64_t, SumOp>(seg, seg, cells, seg_total, s, st)));
    PDX_HIP(hipMemcpyAsync(seg + cells, seg_total, sizeof(int64_t), hipMemcpyDeviceToDevice, st));
    {
      PDX_PROFILE("replay_partials", st);
      if (counts_by_rank)
        hipLaunchKernelGGL((k_replay_ranked_values<uint32_t>), dim3(grid_for(n_own, 256)), dim3(256), 0, st, rv, m, seg, counts_by_rank, W, G,
                           bounds[(size_t)r], n_own, sums_own, bad);
      else
        hipLaunchKernelGGL((k_replay_ranked_values<int64_t>), dim3(grid_for(n_own, 256)), dim3(256), 0, st, rv, m, seg, cnt_local, W, G, bounds[(size_t)r],
                           n_own, sums_own, bad);
    }
    PDX_LAUNCH_CHECK();
    unsigned int hbad = 0;
    unsigned int* pin = static_cast<unsigned int*>(pinned_slot());
    PDX_HIP(hipMemcpyAsync(pin ? pin : &hbad, bad, sizeof(hbad), hipMemcpyDeviceToHost, st));
    PDX_HIP(hipStreamSynchronize(st));
    if (pin) hbad = *reinterpret_cast<volatile unsigned int*>(pin);
    if (hbad) return fail(PDX_DEVICE, "pdx_dist_groupby_sum_mean_count: the received records do not match the exchanged counts (code " + std::to_string(hbad) + ")");
  }
  tm.mark("replay");
  if (!solo) {
    std::vector<int64_t> own_sizes((size_t)W);
    for (int p = 0; p < W; ++p) own_sizes[(size_t)p] = bounds[(size_t)p + 1] - bounds[(size_t)p];
    PDX_TRY(all_gather_v(d, sums_own, own_sizes, 8, res->sums, st));
  }
  if (G) hipLaunchKernelGGL(k_means, dim3(grid_for(G, 256)), dim3(256), 0, st, res->sums, res->counts, G, res->means);
  PDX_LAUNCH_CHECK();
  PDX_HIP(hipStreamSynchronize(st));
  tm.mark("gather_sums+drain");
  *out = res.release();
  return PDX_OK;
}

int64_t pdx_dist_groupby_num_groups(const pdx_dist_groupby* g) { return g ? g->G : -1; }
int64_t pdx_dist_groupby_num_records(const pdx_dist_groupby* g) { return g ? g->records : -1; }
int pdx_dist_groupby_destroy(pdx_dist_groupby* g) {
  delete g;
  return PDX_OK;
}
int pdx_dist_groupby_fetch(const pdx_dist_groupby* g, pdx_mut_column* keys, int64_t* first_rows, double* sums, double* means, int64_t* counts, void* stream) {
  if (!g) return fail(PDX_INVALID, "pdx_dist_groupby_fetch: null handle");
  hipStream_t st = as_stream(stream);
  const size_t b = (size_t)g->G * 8;
  if (keys) {
    if (keys->length < g->G || (g->G && !keys->values)) return fail(PDX_INVALID, "pdx_dist_groupby_fetch: key output too small");
    keys->length = g->G;
    keys->null_count = -1;
    if (b) PDX_HIP(hipMemcpyAsync(keys->values, g->keys, b, hipMemcpyDeviceToDevice, st));
    if (keys->validity && g->G)
      hipLaunchKernelGGL(k_i64_to_bits, dim3(grid_for((g->G + 7) / 8, 256)), dim3(256), 0, st, g->keys_ok, g->G, static_cast<uint8_t*>(keys->validity));
  }
  if (b) {
    if (first_rows) PDX_HIP(hipMemcpyAsync(first_rows, g->first_rows, b, hipMemcpyDeviceToDevice, st));
    if (sums) PDX_HIP(hipMemcpyAsync(sums, g->sums, b, hipMemcpyDeviceToDevice, st));
    if (means) PDX_HIP(hipMemcpyAsync(means, g->means, b, hipMemcpyDeviceToDevice, st));
    if (counts) PDX_HIP(hipMemcpyAsync(counts, g->counts, b, hipMemcpyDeviceToDevice, st));
  }
  PDX_LAUNCH_CHECK();
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}

// df.group_by(key).{min, max, count}(col) and the int64 sum (src/group_by.h:85-139, GROUPBY_AGG / GROUPBY_NUMERIC_AGG
// src/pd_core_macros.h:5-147) over row-range shards: these kinds do not depend on the order of a group's rows, so every rank reduces its
// shard (pdx_groupby_agg on the local dictionary: the accumulate path of gb_acc.hpp), scatters the results into dense G-length partial
// arrays indexed by GLOBAL group id and ONE all-gather + a fold in rank order finishes -- SURVEY 8(e) 3a's reduce-by-key.  Values may
// carry nulls.  kinds: PDX_AGG_MIN / MAX / COUNT, PDX_AGG_SUM for int64 values (float64 sums are order dependent:
// pdx_dist_groupby_sum_mean_count).  Every rank ends with the full result.
int pdx_dist_groupby_order_free(pdx_dist* d, const pdx_column* keys, const pdx_column* values, const int* kinds, int nk, int64_t row_offset, void* stream,
                                pdx_dist_agg** out) {
  if (!d || !out || !kinds || nk <= 0 || nk > 8) return fail(PDX_INVALID, "pdx_dist_groupby_order_free: null argument / more than 8 kinds");
  *out = nullptr;
  int lrc = check_column(keys, "pdx_dist_groupby_order_free");
  if (lrc == PDX_OK) lrc = check_column(values, "pdx_dist_groupby_order_free");
  if (lrc == PDX_OK && values->dtype != PDX_FLOAT64 && values->dtype != PDX_INT64) lrc = fail(PDX_NOT_IMPLEMENTED, "pdx_dist_groupby_order_free: values must be int64 or float64");
  if (lrc == PDX_OK && values->length != keys->length) lrc = fail(PDX_INVALID, "pdx_dist_groupby_order_free: keys and values differ in length");
  bool want_min = false, want_max = false, want_sum = false;
  for (int k = 0; k < nk && lrc == PDX_OK; ++k) {
    if (kinds[k] == PDX_AGG_MIN) want_min = true;
    else if (kinds[k] == PDX_AGG_MAX) want_max = true;
    else if (kinds[k] == PDX_AGG_COUNT) continue;
    else if (kinds[k] == PDX_AGG_SUM && values->dtype == PDX_INT64) want_sum = true;
    else lrc = fail(PDX_NOT_IMPLEMENTED, "pdx_dist_groupby_order_free: min / max / count, and sum of int64 values (order-dependent kinds: pdx_dist_groupby_sum_mean_count)");
  }
  hipStream_t st = as_stream(stream);
  const int W = d->world;
  const bool solo = W == 1 && !d->force;
  Scratch s;
  std::unique_ptr<pdx_dist_agg> res(new pdx_dist_agg());
  res->stream = st;
  res->key_dtype = lrc == PDX_OK ? keys->dtype : PDX_INT64;
  GlobalDict D;
  PDX_TRY(build_dictionary(d, keys, lrc, "pdx_dist_groupby_order_free", row_offset, s, st, [&](size_t c) { return res->own<int64_t>(c); }, &D));
  res->keys = D.keys;
  res->keys_ok = D.keys_ok;
  res->first_rows = D.first_rows;
  const int64_t Gl = D.Gl, G = D.G;
  res->G = G;
  const bool is_f = values->dtype == PDX_FLOAT64;
  const bool nullable = validity_or_null(values) != nullptr;
  // ---- the shard's own aggregates (local group order).  The rows per group ride along when a maximum must know whether the group
  // holds a null anywhere (its tie rule); nullability is a per-rank fact, so every rank sends the array (a shard without nulls: rows == count)
  const bool need_rows = want_max && is_f;
  int a = 1;
  const int a_rows = need_rows ? a++ : -1, a_min = want_min ? a++ : -1, a_max = want_max ? a++ : -1, a_sum = want_sum ? a++ : -1, A = a;
  int64_t* local = s.get<int64_t>((size_t)A * (size_t)std::max<int64_t>(Gl, 1));
  uint8_t* lbits = nullable ? s.get<uint8_t>((size_t)(Gl + 7) / 8 + 16) : nullptr;
  PDX_SCRATCH_CHECK(s);
  if (Gl) {
    int lk[4];
    pdx_mut_column lo[4];
    int m = 0;
    auto add = [&](int kind, int slot, int dt) {
      lk[m] = kind;
      lo[m] = pdx_mut_column{};
      lo[m].dtype = dt;
      lo[m].length = Gl;
      lo[m].values = local + (size_t)slot * Gl;
      lo[m].validity = (nullable && kind != PDX_AGG_COUNT) ? lbits : nullptr;  // (group validity is re-derived from the counts)
      ++m;
    };
    add(PDX_AGG_COUNT, 0, PDX_INT64);
    if (want_min) add(PDX_AGG_MIN, a_min, values->dtype);
    if (want_max) add(PDX_AGG_MAX, a_max, values->dtype);
    if (want_sum) add(PDX_AGG_SUM, a_sum, PDX_INT64);
    PDX_TRY(pdx_groupby_agg(D.gb, values, lk, m, lo, st));
    if (need_rows) {
      if (nullable) {  // rows per group = the count of the same column without its validity
        pdx_column all = *values;
        all.validity = nullptr;
        all.null_count = 0;
        int ck = PDX_AGG_COUNT;
        pdx_mut_column co{};
        co.dtype = PDX_INT64;
        co.length = Gl;
        co.values = local + (size_t)a_rows * Gl;
        PDX_TRY(pdx_groupby_agg(D.gb, &all, &ck, 1, &co, st));
      } else {
        PDX_HIP(hipMemcpyAsync(local + (size_t)a_rows * Gl, local, (size_t)Gl * 8, hipMemcpyDeviceToDevice, st));
      }
    }
  }
  // ---- dense partials by global id, one all-gather, fold
  int64_t* dense = s.get<int64_t>((size_t)A * (size_t)std::max<int64_t>(G, 1));
  int64_t* all_parts = solo ? dense : s.get<int64_t>((size_t)W * A * (size_t)std::max<int64_t>(G, 1));
  PDX_SCRATCH_CHECK(s);
  if (G) PDX_HIP(hipMemsetAsync(dense, 0, (size_t)A * G * 8, st));  // count 0 = "this rank has no valid value of the group": the other arrays are not read
  for (int j = 0; j < A && Gl; ++j)
    hipLaunchKernelGGL(k_scatter_i64, dim3(grid_for(Gl, 256)), dim3(256), 0, st, local + (size_t)j * Gl, D.my_map, Gl, dense + (size_t)j * G);
  PDX_LAUNCH_CHECK();
  if (!solo && G) PDX_TRY(d->tr.all_gather(d->tr.ctx, dense, all_parts, (size_t)A * G * 8, st));
  uint64_t *omin = nullptr, *omax = nullptr, *osum = nullptr, *ocnt = nullptr;
  int64_t* ok = nullable ? res->own<int64_t>((size_t)G) : nullptr;
  if (nullable && !ok) return PDX_OOM;
  for (int k = 0; k < nk; ++k) {
    uint64_t** slot = kinds[k] == PDX_AGG_MIN ? &omin : kinds[k] == PDX_AGG_MAX ? &omax : kinds[k] == PDX_AGG_SUM ? &osum : &ocnt;
    if (!*slot) *slot = res->own<uint64_t>((size_t)G);
    if (!*slot) return PDX_OOM;
    res->vals.push_back(*slot);
    res->dtypes.push_back(kinds[k] == PDX_AGG_COUNT || kinds[k] == PDX_AGG_SUM ? PDX_INT64 : values->dtype);
    res->oks.push_back(kinds[k] == PDX_AGG_COUNT ? nullptr : ok);
  }
  if (G) {
    const int Wf = solo ? 1 : W;
    if (is_f)
      hipLaunchKernelGGL((k_fold_partials<double>), dim3(grid_for(G, 256)), dim3(256), 0, st, all_parts, Wf, A, G, a_min, a_max, a_sum, a_rows,
                         reinterpret_cast<double*>(omin), reinterpret_cast<double*>(omax), reinterpret_cast<long long*>(osum), reinterpret_cast<long long*>(ocnt), ok);
    else
      hipLaunchKernelGGL((k_fold_partials<long long>), dim3(grid_for(G, 256)), dim3(256), 0, st, all_parts, Wf, A, G, a_min, a_max, a_sum, a_rows,
                         reinterpret_cast<long long*>(omin), reinterpret_cast<long long*>(omax), reinterpret_cast<long long*>(osum), reinterpret_cast<long long*>(ocnt), ok);
    PDX_LAUNCH_CHECK();
  }
  PDX_HIP(hipStreamSynchronize(st));
  *out = res.release();
  return PDX_OK;
}
int64_t pdx_dist_agg_num_groups(const pdx_dist_agg* g) { return g ? g->G : -1; }
int pdx_dist_agg_destroy(pdx_dist_agg* g) {
  delete g;
  return PDX_OK;
}
// keys / first_rows may be NULL; outs[k]: the k-th requested kind (dtype as pdx_groupby_agg; a validity buffer is filled when given)
int pdx_dist_agg_fetch(const pdx_dist_agg* g, pdx_mut_column* keys, int64_t* first_rows, pdx_mut_column* outs, void* stream) {
  if (!g) return fail(PDX_INVALID, "pdx_dist_agg_fetch: null handle");
  hipStream_t st = as_stream(stream);
  const size_t b = (size_t)g->G * 8;
  if (keys) {
    if (keys->length < g->G || (g->G && !keys->values)) return fail(PDX_INVALID, "pdx_dist_agg_fetch: key output too small");
    keys->length = g->G;
    keys->null_count = -1;
    if (b) PDX_HIP(hipMemcpyAsync(keys->values, g->keys, b, hipMemcpyDeviceToDevice, st));
    if (keys->validity && g->G)
      hipLaunchKernelGGL(k_i64_to_bits, dim3(grid_for((g->G + 7) / 8, 256)), dim3(256), 0, st, g->keys_ok, g->G, static_cast<uint8_t*>(keys->validity));
  }
  if (b && first_rows) PDX_HIP(hipMemcpyAsync(first_rows, g->first_rows, b, hipMemcpyDeviceToDevice, st));
  for (size_t k = 0; outs && k < g->vals.size(); ++k) {
    pdx_mut_column& o = outs[k];
    if (o.length < g->G || (g->G && !o.values)) return fail(PDX_INVALID, "pdx_dist_agg_fetch: output too small");
    if (o.dtype != g->dtypes[k]) return fail(PDX_INVALID, "pdx_dist_agg_fetch: output dtype does not match the aggregate's result type");
    o.length = g->G;
    if (b) PDX_HIP(hipMemcpyAsync(o.values, g->vals[k], b, hipMemcpyDeviceToDevice, st));
    if (g->oks[k]) {
      if (!o.validity) return fail(PDX_INVALID, "pdx_dist_agg_fetch: the result carries nulls but an output has no validity buffer");
      if (g->G) hipLaunchKernelGGL(k_i64_to_bits, dim3(grid_for((g->G + 7) / 8, 256)), dim3(256), 0, st, g->oks[k], g->G, static_cast<uint8_t*>(o.validity));
      o.null_count = -1;
    } else {
      if (o.validity && g->G) PDX_HIP(hipMemsetAsync(o.validity, 0xFF, (size_t)(g->G + 7) / 8, st));
      o.null_count = 0;
    }
  }
  PDX_LAUNCH_CHECK();
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}

// pd::concat of row-range shards (src/concat.cpp:116-190 over the shards' results): the all-gather(v) merge in rank order.
// part: this rank's rows (int64 / uint64 / timestamp / float64); out: capacity >= the total; validity stitched from 0/1 words.
int pdx_dist_concat(pdx_dist* d, const pdx_column* part, pdx_mut_column* out, void* stream) {
  if (!d || !out) return fail(PDX_INVALID, "pdx_dist_concat: null argument");
  PDX_TRY(check_column(part, "pdx_dist_concat"));
  if (part->dtype == PDX_BOOL) return fail(PDX_NOT_IMPLEMENTED, "pdx_dist_concat: boolean columns are not supported (8-byte value columns only)");
  if (out->dtype != part->dtype) return fail(PDX_INVALID, "pdx_dist_concat: output dtype differs");
  hipStream_t st = as_stream(stream);
  Scratch s;
  std::vector<int64_t> meta, mine{part->length, validity_or_null(part) ? 1 : 0};
  PDX_TRY(gather_host(d, mine.data(), 2, &meta, s, st));
  std::vector<int64_t> sizes((size_t)d->world);
  int64_t total = 0;
  bool any_nulls = false;
  for (int p = 0; p < d->world; ++p) {
    sizes[(size_t)p] = meta[(size_t)p * 2];
    total += sizes[(size_t)p];
    any_nulls = any_nulls || meta[(size_t)p * 2 + 1] != 0;
  }
  if (out->length < total || (total && !out->values)) return fail(PDX_INVALID, "pdx_dist_concat: output too small");
  if (any_nulls && !out->validity) return fail(PDX_INVALID, "pdx_dist_concat: a shard carries nulls but the output has no validity buffer");
  out->length = total;
  out->null_count = any_nulls ? -1 : 0;
  const uint64_t* vin = static_cast<const uint64_t*>(part->values) + part->offset;
  PDX_TRY(all_gather_v(d, vin, sizes, 8, out->values, st));
  if (any_nulls) {
    // validity travels as one 0/1 word per row: bit offsets differ per shard, so the bits are re-packed on the receiver
    const int64_t n = part->length;
    int64_t* okw = s.get<int64_t>((size_t)n);
    int64_t* all_ok = s.get<int64_t>((size_t)total);
    PDX_SCRATCH_CHECK(s);
    if (n) hipLaunchKernelGGL(k_bits_to_i64, dim3(grid_for(n, 256)), dim3(256), 0, st, validity_or_null(part), part->offset, n, okw);
    PDX_LAUNCH_CHECK();
    PDX_TRY(all_gather_v(d, okw, sizes, 8, all_ok, st));
    if (total) hipLaunchKernelGGL(k_i64_to_bits, dim3(grid_for((total + 7) / 8, 256)), dim3(256), 0, st, all_ok, total, static_cast<uint8_t*>(out->validity));
    PDX_LAUNCH_CHECK();
  }
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}


// pd::resample(df, rule).{kinds}(col) over an axis sharded by row ranges in rank order (src/resample.h:91-122, src/group_by.h:255-299).
// Sorted timestamps make the shards time ranges.  Bins are made whole before any arithmetic: the leading rows of a shard that fall
// into a bin already open on an earlier rank move to that bin's first rank (ONE all-to-all(v); usually a few hundred rows per cut),
// every rank resamples its rows on the WHOLE axis' grid (pdx_resample_grid of the all-gathered extremes), and the (label, value)
// rows are all-gathered in rank order == label order.  The reference's whole-axis errors (unsorted input, values outside the
// bins, upsampling) are raised on every rank.
int pdx_dist_resample(pdx_dist* d, const pdx_column* ts, const pdx_column* values, const int* kinds, int nk, int64_t freq_ns, int closed_right,
                      int label_right, int origin_type, int64_t origin_custom_ns, int64_t offset_ns, void* stream, pdx_dist_resampled** out) {
  if (!d || !out || !kinds || nk <= 0) return fail(PDX_INVALID, "pdx_dist_resample: null argument");
  *out = nullptr;
  // local preconditions are collected and travel with the first all-gather: a rank that fails on its shard must not return while its
  // peers wait for it in a collective (every rank then returns the error)
  int lrc = check_column(ts, "pdx_dist_resample");
  if (lrc == PDX_OK) lrc = check_column(values, "pdx_dist_resample");
  if (lrc == PDX_OK && ts->dtype != PDX_TIMESTAMP_NS && ts->dtype != PDX_INT64) lrc = fail(PDX_INVALID, "axis must be a TimestampArray");
  if (lrc == PDX_OK && validity_or_null(ts)) lrc = fail(PDX_NOT_IMPLEMENTED, "pdx_dist_resample: null timestamps are not supported");
  if (lrc == PDX_OK && values->dtype != PDX_FLOAT64 && values->dtype != PDX_INT64) lrc = fail(PDX_NOT_IMPLEMENTED, "pdx_dist_resample: values must be int64 or float64");
  if (lrc == PDX_OK && values->length != ts->length) lrc = fail(PDX_INVALID, "pdx_dist_resample: axis and values differ in length");
  hipStream_t st = as_stream(stream);
  const int W = d->world, r = d->rank;
  const bool solo = W == 1 && !d->force;
  Scratch s;
  const int64_t n_loc = lrc == PDX_OK ? ts->length : 0;
  const long long* tv = lrc == PDX_OK ? static_cast<const long long*>(ts->values) + ts->offset : nullptr;
  const uint64_t* vv = lrc == PDX_OK ? static_cast<const uint64_t*>(values->values) + values->offset : nullptr;
  // ---- every shard's (rows, first, last, has nulls, local status): the whole axis' extremes and grid on every rank
  int64_t mine[5] = {n_loc, 0, 0, (lrc == PDX_OK && validity_or_null(values)) ? 1 : 0, lrc};
  if (n_loc) {
    // is the shard sorted INSIDE?  (its boundaries are compared across ranks below; an unsorted stretch inside would only surface in
    // this rank's pdx_resample_create, after the row exchange)
    unsigned int* dflag = s.get<unsigned int>(1);
    if (s.failed) mine[4] = lrc = PDX_OOM;
    else {
      unsigned int hflag = 0;
      hipError_t e = hipMemsetAsync(dflag, 0, sizeof(unsigned int), st);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(k_any_descent, dim3(grid_for(n_loc, 256, 4)), dim3(256), 0, st, tv, n_loc, dflag);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipMemcpyAsync(&mine[1], tv, 8, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipMemcpyAsync(&mine[2], tv + (n_loc - 1), 8, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipMemcpyAsync(&hflag, dflag, sizeof(unsigned int), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e != hipSuccess) mine[4] = lrc = hip_fail(e, "pdx_dist_resample");
      else if (hflag) mine[4] = lrc = fail(PDX_INVALID, "pdx_resample_create: timestamps must be sorted ascending");
    }
  }
  std::vector<int64_t> info;
  {
    const int grc = gather_host(d, mine, 5, &info, s, st);
    if (grc != PDX_OK) return lrc != PDX_OK ? lrc : grc;
  }
  PDX_TRY(gate_status(info, 5, 4, W, r, lrc, "pdx_dist_resample"));
  auto I = [&](int q, int k) { return info[(size_t)q * 5 + k]; };
  std::vector<int> live;
  int64_t N = 0;
  bool any_nulls = false;
  for (int q = 0; q < W; ++q) {
    if (I(q, 0) > 0) live.push_back(q);
    N += I(q, 0);
    any_nulls = any_nulls || I(q, 3) != 0;
  }
  std::unique_ptr<pdx_dist_resampled> res(new pdx_dist_resampled());
  res->stream = st;
  res->nk = nk;
  if (N == 0) {
    res->labels = res->own<int64_t>(1);
    for (int k = 0; k < nk; ++k) {
      res->dtypes.push_back(PDX_FLOAT64);
      res->vals.push_back(res->own<uint64_t>(1));
      res->oks.push_back(nullptr);
    }
    *out = res.release();
    return PDX_OK;
  }
  for (size_t i = 0; i + 1 < live.size(); ++i)
    if (I(live[i], 2) > I(live[i + 1], 1)) return fail(PDX_INVALID, "pdx_resample_create: timestamps must be sorted ascending");
  int64_t first_edge = 0, nbins = 0;
  PDX_TRY(pdx_resample_grid(I(live.front(), 1), I(live.back(), 2), freq_ns, closed_right, origin_type, origin_custom_ns, offset_ns, &first_edge, &nbins));
  if (N < nbins) return fail(PDX_INVALID, "upSampling is not implemented.");  // GroupInfo::upsampling on the whole axis (src/resample.h:14-17)
  auto floor_div = [](int64_t a, int64_t b) { int64_t q = a / b, m = a % b; return (m != 0 && ((m < 0) != (b < 0))) ? q - 1 : q; };
  auto bin_of = [&](int64_t t) { return floor_div(t - first_edge - (closed_right ? 1 : 0), freq_ns); };  // [e_k, e_k + f) or (e_k, e_k + f]
  // ---- the first rank that holds rows of my leading bin; the rows of that bin move there
  int send_to = r;
  int64_t m = 0;
  if (n_loc) {
    const int64_t b0 = bin_of(I(r, 1));
    for (int i = (int)live.size() - 1; i >= 0; --i) {
      const int q = live[(size_t)i];
      if (q >= r) continue;
      if (bin_of(I(q, 2)) != b0) break;
      send_to = q;
      if (bin_of(I(q, 1)) != b0) break;
    }
    if (send_to != r) {
      int64_t* dm = s.get<int64_t>(1);
      PDX_SCRATCH_CHECK(s);
      hipLaunchKernelGGL(k_count_below, dim3(1), dim3(1), 0, st, tv, n_loc, (long long)(first_edge + (b0 + 1) * freq_ns), closed_right ? 1 : 0, dm);
      PDX_LAUNCH_CHECK();
      PDX_HIP(hipMemcpyAsync(&m, dm, 8, hipMemcpyDeviceToHost, st));
      PDX_HIP(hipStreamSynchronize(st));
    }
  }
  // ---- the exchange of the moved rows (timestamps, values, validity words) and the shard as it is resampled
  const long long* ts2 = tv;
  const uint64_t* vals2 = vv;
  const uint8_t* valid2 = validity_or_null(values);
  int64_t valid2_off = values->offset, n2 = n_loc;
  if (!solo) {
    std::vector<int64_t> moves;  // moves[q] = (rows q sends, destination of q)
    int64_t mv[2] = {m, send_to};
    PDX_TRY(gather_host(d, mv, 2, &moves, s, st));
    int64_t recv_rows = 0;
    std::vector<size_t> so((size_t)W, 0), sb((size_t)W, 0), ro((size_t)W, 0), rb((size_t)W, 0);
    for (int q = 0; q < W; ++q) {
      if (q == r) continue;
      if (moves[(size_t)q * 2 + 1] == r) {
        ro[(size_t)q] = (size_t)recv_rows * 8;
        rb[(size_t)q] = (size_t)moves[(size_t)q * 2] * 8;
        recv_rows += moves[(size_t)q * 2];
      }
    }
    if (send_to != r) sb[(size_t)send_to] = (size_t)m * 8;
    n2 = n_loc - m + recv_rows;
    long long* t2 = s.get<long long>((size_t)n2);
    uint64_t* v2 = s.get<uint64_t>((size_t)n2);
    int64_t* okw = any_nulls ? s.get<int64_t>((size_t)n_loc) : nullptr;
    int64_t* ok2 = any_nulls ? s.get<int64_t>((size_t)n2) : nullptr;
    uint8_t* bits2 = any_nulls ? s.get<uint8_t>((size_t)(n2 + 7) / 8 + 16) : nullptr;
    PDX_SCRATCH_CHECK(s);
    const int64_t keep = n_loc - m;
    if (keep) {
      PDX_HIP(hipMemcpyAsync(t2, tv + m, (size_t)keep * 8, hipMemcpyDeviceToDevice, st));
      PDX_HIP(hipMemcpyAsync(v2, vv + m, (size_t)keep * 8, hipMemcpyDeviceToDevice, st));
    }
    PDX_TRY(d->tr.all_to_all_v(d->tr.ctx, tv, so.data(), sb.data(), t2 + keep, ro.data(), rb.data(), st));
    PDX_TRY(d->tr.all_to_all_v(d->tr.ctx, vv, so.data(), sb.data(), v2 + keep, ro.data(), rb.data(), st));
    if (any_nulls) {
      if (n_loc) hipLaunchKernelGGL(k_bits_to_i64, dim3(grid_for(n_loc, 256)), dim3(256), 0, st, validity_or_null(values), values->offset, n_loc, okw);
      PDX_LAUNCH_CHECK();
      if (keep) PDX_HIP(hipMemcpyAsync(ok2, okw + m, (size_t)keep * 8, hipMemcpyDeviceToDevice, st));
      PDX_TRY(d->tr.all_to_all_v(d->tr.ctx, okw, so.data(), sb.data(), ok2 + keep, ro.data(), rb.data(), st));
      if (n2) hipLaunchKernelGGL(k_i64_to_bits, dim3(grid_for((n2 + 7) / 8, 256)), dim3(256), 0, st, ok2, n2, bits2);
      PDX_LAUNCH_CHECK();
    }
    ts2 = t2;
    vals2 = v2;
    valid2 = bits2;
    valid2_off = 0;
  }
  // ---- every shard bins on the whole axis' grid, anchored at its FIRST EDGE (with a negative first offset the reference's grid starts
  // at the first timestamp itself, src/resample.cpp:85-178)
  struct Handle {
    pdx_groupby* gb = nullptr;
    ~Handle() { if (gb) pdx_groupby_destroy(gb); }
  } h;
  pdx_column tc{};
  tc.dtype = PDX_TIMESTAMP_NS;
  tc.length = n2;
  tc.values = ts2;
  pdx_column vc{};
  vc.dtype = values->dtype;
  vc.length = n2;
  vc.offset = solo ? values->offset : 0;
  vc.values = solo ? values->values : (const void*)vals2;
  vc.validity = valid2;
  vc.null_count = valid2 ? -1 : 0;
  (void)valid2_off;
  int64_t Gl = 0;
  int crc = PDX_OK;  // the shard's own create: its status travels with the bin counts (same gate as above)
  if (n2) {
    crc = pdx_resample_create(&tc, freq_ns, closed_right, label_right, PDX_ORIGIN_CUSTOM | PDX_ORIGIN_SHARD, first_edge, 0, st, &h.gb);
    if (crc == PDX_OK) Gl = pdx_groupby_num_groups(h.gb);
  }
  std::vector<int64_t> sizes((size_t)W), ginfo;
  {
    int64_t gm[2] = {Gl, crc};
    const int grc = gather_host(d, gm, 2, &ginfo, s, st);
    if (grc != PDX_OK) return crc != PDX_OK ? crc : grc;
  }
  PDX_TRY(gate_status(ginfo, 2, 1, W, r, crc, "pdx_dist_resample"));
  int64_t G = 0;
  for (int q = 0; q < W; ++q) {
    sizes[(size_t)q] = ginfo[(size_t)q * 2];
    G += sizes[(size_t)q];
  }
  res->G = G;
  res->labels = res->own<int64_t>((size_t)G);
  if (!res->labels) return PDX_OOM;
  int64_t* lab_l = s.get<int64_t>((size_t)Gl);
  PDX_SCRATCH_CHECK(s);
  std::vector<pdx_mut_column> outs((size_t)nk);
  std::vector<uint8_t*> out_bits((size_t)nk, nullptr);
  for (int k = 0; k < nk; ++k) {
    const int kind = kinds[k];
    if (kind < PDX_AGG_SUM || kind > PDX_AGG_LAST) return fail(PDX_NOT_IMPLEMENTED, "pdx_dist_resample: aggregate kind not supported on shards");
    const int dt = (kind == PDX_AGG_MEAN || kind == PDX_AGG_VARIANCE || kind == PDX_AGG_STDDEV) ? PDX_FLOAT64 : kind == PDX_AGG_COUNT ? PDX_INT64 : values->dtype;
    res->dtypes.push_back(dt);
    res->vals.push_back(res->own<uint64_t>((size_t)G));
    res->oks.push_back(any_nulls && kind != PDX_AGG_COUNT ? res->own<int64_t>((size_t)G) : nullptr);
    if (!res->vals.back() || (any_nulls && kind != PDX_AGG_COUNT && !res->oks.back())) return PDX_OOM;
    pdx_mut_column& mcol = outs[(size_t)k];
    mcol = pdx_mut_column{};
    mcol.dtype = dt;
    mcol.length = Gl;
    mcol.values = s.get<uint64_t>((size_t)Gl);
    out_bits[(size_t)k] = valid2 ? s.get<uint8_t>((size_t)(Gl + 7) / 8 + 16) : nullptr;
    mcol.validity = out_bits[(size_t)k];
    PDX_SCRATCH_CHECK(s);
  }
  if (n2) {
    pdx_mut_column lm{};
    lm.dtype = PDX_TIMESTAMP_NS;
    lm.length = Gl;
    lm.values = lab_l;
    PDX_TRY(pdx_groupby_unique_keys(h.gb, &lm, st));
    PDX_TRY(pdx_groupby_agg(h.gb, &vc, kinds, nk, outs.data(), st));
  }
  PDX_TRY(all_gather_v(d, lab_l, sizes, 8, res->labels, st));
  int64_t* okl = any_nulls ? s.get<int64_t>((size_t)Gl) : nullptr;
  PDX_SCRATCH_CHECK(s);
  for (int k = 0; k < nk; ++k) {
    PDX_TRY(all_gather_v(d, outs[(size_t)k].values, sizes, 8, res->vals[(size_t)k], st));
    if (res->oks[(size_t)k]) {
      // (a shard without nulls contributes all-valid words)
      if (Gl) hipLaunchKernelGGL(k_bits_to_i64, dim3(grid_for(Gl, 256)), dim3(256), 0, st, (const uint8_t*)out_bits[(size_t)k], (int64_t)0, Gl, okl);
      PDX_LAUNCH_CHECK();
      PDX_TRY(all_gather_v(d, okl, sizes, 8, res->oks[(size_t)k], st));
    }
  }
  PDX_HIP(hipStreamSynchronize(st));
  *out = res.release();
  return PDX_OK;
}
int64_t pdx_dist_resampled_num_bins(const pdx_dist_resampled* g) { return g ? g->G : -1; }
int pdx_dist_resampled_destroy(pdx_dist_resampled* g) {
  delete g;
  return PDX_OK;
}
// labels: PDX_TIMESTAMP_NS, capacity >= num_bins; outs[k]: the k-th aggregate (dtype as pdx_groupby_agg; a validity buffer is filled when given)
int pdx_dist_resampled_fetch(const pdx_dist_resampled* g, pdx_mut_column* labels, pdx_mut_column* outs, void* stream) {
  if (!g) return fail(PDX_INVALID, "pdx_dist_resampled_fetch: null handle");
  hipStream_t st = as_stream(stream);
  const size_t b = (size_t)g->G * 8;
  if (labels) {
    if (labels->length < g->G || (g->G && !labels->values)) return fail(PDX_INVALID, "pdx_dist_resampled_fetch: label output too small");
    labels->length = g->G;
    labels->null_count = 0;
    if (b) PDX_HIP(hipMemcpyAsync(labels->values, g->labels, b, hipMemcpyDeviceToDevice, st));
  }
  for (int k = 0; outs && k < g->nk; ++k) {
    pdx_mut_column& o = outs[k];
    if (o.length < g->G || (g->G && !o.values)) return fail(PDX_INVALID, "pdx_dist_resampled_fetch: output too small");
    if (o.dtype != g->dtypes[(size_t)k]) return fail(PDX_INVALID, "pdx_dist_resampled_fetch: output dtype does not match the aggregate's result type");
    o.length = g->G;
    if (b) PDX_HIP(hipMemcpyAsync(o.values, g->vals[(size_t)k], b, hipMemcpyDeviceToDevice, st));
    if (g->oks[(size_t)k]) {
      if (!o.validity) return fail(PDX_INVALID, "pdx_dist_resampled_fetch: the result carries nulls but an output has no validity buffer");
      if (g->G) hipLaunchKernelGGL(k_i64_to_bits, dim3(grid_for((g->G + 7) / 8, 256)), dim3(256), 0, st, g->oks[(size_t)k], g->G, static_cast<uint8_t*>(o.validity));
      o.null_count = -1;
    } else {
      if (o.validity && g->G) PDX_HIP(hipMemsetAsync(o.validity, 0xFF, (size_t)(g->G + 7) / 8, st));
      o.null_count = 0;
    }
  }
  PDX_LAUNCH_CHECK();
  PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}

// Inputs of more than 2^31 - 1 rows (the per-call limit of the 32-bit row ids inside pdx_groupby_create): the rows are cut into chunks,
// every chunk plays one rank of the exchange above -- a host thread with its own stream on the SAME device, the collectives are
// device-to-device copies between the threads' buffers -- and the partial-tree records make the merged sums bit-identical to one
// pairwise tree over the whole column.  chunk_rows = 0: the largest chunk the limit allows.
int pdx_groupby_sum_mean_count_chunked(const pdx_column* keys, const pdx_column* values, int64_t chunk_rows, void* stream, pdx_dist_groupby** out) {
  if (!out) return fail(PDX_INVALID, "pdx_groupby_sum_mean_count_chunked: null output");
  *out = nullptr;
  PDX_TRY(check_column(keys, "pdx_groupby_sum_mean_count_chunked"));
  PDX_TRY(check_column(values, "pdx_groupby_sum_mean_count_chunked"));
  if (values->length != keys->length) return fail(PDX_INVALID, "pdx_groupby_sum_mean_count_chunked: keys and values differ in length");
  if (values->dtype != PDX_FLOAT64 || validity_or_null(values))
    return fail(PDX_NOT_IMPLEMENTED, "pdx_groupby_sum_mean_count_chunked: float64 values without nulls");
  const int64_t n = keys->length, kMaxChunk = 0x7FFFF000ll;
  if (chunk_rows <= 0 || chunk_rows > kMaxChunk) chunk_rows = kMaxChunk;
  const int W = (int)std::max<int64_t>(1, ceil_div(n, chunk_rows));
  if (W > 64) return fail(PDX_INVALID, "pdx_groupby_sum_mean_count_chunked: more than 64 chunks");
  int device = 0;
  PDX_HIP(hipGetDevice(&device));
  (void)stream;  // (every chunk runs on a stream of its own; the call returns when all of them have finished)
  LocalShared sh;
  sh.W = W;
  sh.send.assign((size_t)W, nullptr);
  sh.soff.assign((size_t)W, nullptr);
  sh.sbytes.assign((size_t)W, nullptr);
  std::vector<int> rcs((size_t)W, PDX_OK);
  std::vector<std::string> errs((size_t)W);
  std::vector<pdx_dist_groupby*> results((size_t)W, nullptr);
  auto work = [&](int r) {
    int rc = PDX_OK;
    hipStream_t st = nullptr;
    pdx_dist* d = nullptr;
    LocalCtx ctx{&sh, r};
    do {
      if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
        rc = fail(PDX_DEVICE, "chunked group-by: no stream for a chunk");
        break;
      }
      pdx_dist_transport tr{&ctx, local_all_gather, local_all_to_all_v};
      if ((rc = pdx_dist_init_custom(&tr, W, r, &d)) != PDX_OK) break;
      const int64_t lo = (int64_t)r * chunk_rows, len = std::min<int64_t>(chunk_rows, n - lo);
      pdx_column k = *keys, v = *values;
      k.offset += lo;
      k.length = len;
      v.offset += lo;
      v.length = len;
      rc = pdx_dist_groupby_sum_mean_count(d, &k, &v, lo, st, &results[(size_t)r]);
    } while (false);
    if (rc != PDX_OK) {
      errs[(size_t)r] = pdx_last_error();
      sh.abort();
    }
    rcs[(size_t)r] = rc;
    if (d) pdx_dist_destroy(d);
    if (st) {
      (void)hipStreamSynchronize(st);
      if (results[(size_t)r]) results[(size_t)r]->stream = nullptr;  // the chunk's stream dies with this thread: later frees go to the default stream
      (void)hipStreamDestroy(st);
    }
  };
  if (W == 1) {
    work(0);
  } else {
    std::vector<std::thread> threads;
    for (int r = 0; r < W; ++r) threads.emplace_back(work, r);
    for (auto& t : threads) t.join();
  }
  int rc = PDX_OK;
  for (int r = 0; r < W; ++r)
    if (rcs[(size_t)r] != PDX_OK && (rc == PDX_OK || errs[(size_t)r].find("another chunk failed") == std::string::npos)) {
      rc = rcs[(size_t)r];
      set_error(errs[(size_t)r]);
    }
  for (int r = 1; r < W; ++r) delete results[(size_t)r];  // every chunk holds the full result: keep the first
  if (rc != PDX_OK) {
    delete results[0];
    return rc;
  }
  *out = results[0];
  return PDX_OK;
}

// The order-free kinds (min / max / count, int64 sum) for inputs of more than 2^31 - 1 rows: chunks play the ranks of pdx_dist_groupby_order_free on
// host threads of their own (as pdx_groupby_sum_mean_count_chunked does for the partial-tree exchange): every chunk reduces its rows without a value
// sort, the dense per-group partials are folded in chunk order.  chunk_rows = 0: the largest chunk the 32-bit row ids allow.
int pdx_groupby_order_free_chunked(const pdx_column* keys, const pdx_column* values, const int* kinds, int nk, int64_t chunk_rows, void* stream, pdx_dist_agg** out) {
  if (!out || !kinds || nk <= 0) return fail(PDX_INVALID, "pdx_groupby_order_free_chunked: null argument");
  *out = nullptr;
  PDX_TRY(check_column(keys, "pdx_groupby_order_free_chunked"));
  PDX_TRY(check_column(values, "pdx_groupby_order_free_chunked"));
  if (values->length != keys->length) return fail(PDX_INVALID, "pdx_groupby_order_free_chunked: keys and values differ in length");
  const int64_t n = keys->length, kMaxChunk = 0x7FFFF000ll;
  if (chunk_rows <= 0 || chunk_rows > kMaxChunk) chunk_rows = kMaxChunk;
  const int W = (int)std::max<int64_t>(1, ceil_div(n, chunk_rows));
  if (W > 64) return fail(PDX_INVALID, "pdx_groupby_order_free_chunked: more than 64 chunks");
  int device = 0;
  PDX_HIP(hipGetDevice(&device));
  (void)stream;  // (every chunk runs on a stream of its own; the call returns when all of them have finished)
  LocalShared sh;
  sh.W = W;
  sh.send.assign((size_t)W, nullptr);
  sh.soff.assign((size_t)W, nullptr);
  sh.sbytes.assign((size_t)W, nullptr);
  std::vector<int> rcs((size_t)W, PDX_OK);
  std::vector<std::string> errs((size_t)W);
  std::vector<pdx_dist_agg*> results((size_t)W, nullptr);
  auto work = [&](int r) {
    int rc = PDX_OK;
    hipStream_t st = nullptr;
    pdx_dist* d = nullptr;
    LocalCtx ctx{&sh, r};
    do {
      if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
        rc = fail(PDX_DEVICE, "chunked group-by: no stream for a chunk");
        break;
      }
      pdx_dist_transport tr{&ctx, local_all_gather, local_all_to_all_v};
      if ((rc = pdx_dist_init_custom(&tr, W, r, &d)) != PDX_OK) break;
      const int64_t lo = (int64_t)r * chunk_rows, len = std::min<int64_t>(chunk_rows, n - lo);
      pdx_column k = *keys, v = *values;
      k.offset += lo;
      k.length = len;
      v.offset += lo;
      v.length = len;
      rc = pdx_dist_groupby_order_free(d, &k, &v, kinds, nk, lo, st, &results[(size_t)r]);
    } while (false);
    if (rc != PDX_OK) {
      errs[(size_t)r] = pdx_last_error();
      sh.abort();
    }
    rcs[(size_t)r] = rc;
    if (d) pdx_dist_destroy(d);
    if (st) {
      (void)hipStreamSynchronize(st);
      if (results[(size_t)r]) results[(size_t)r]->stream = nullptr;  // the chunk's stream dies with this thread
      (void)hipStreamDestroy(st);
    }
  };
  if (W == 1) {
    work(0);
  } else {
    std::vector<std::thread> threads;
    for (int r = 0; r < W; ++r) threads.emplace_back(work, r);
    for (auto& t : threads) t.join();
  }
  int rc = PDX_OK;
  for (int r = 0; r < W; ++r)
    if (rcs[(size_t)r] != PDX_OK && (rc == PDX_OK || errs[(size_t)r].find("another chunk failed") == std::string::npos)) {
      rc = rcs[(size_t)r];
      set_error(errs[(size_t)r]);
    }
  for (int r = 1; r < W; ++r) delete results[(size_t)r];  // every chunk holds the full result: keep the first
  if (rc != PDX_OK) {
    delete results[0];
    return rc;
  }
  *out = results[0];
  return PDX_OK;
}


}  // extern "C"
