// Microbenchmark (round 4): how fast can 1e9 rows update G-length per-group tables WITHOUT a sort?
// Variants: global atomics (add u32 / add u64 / min u64), filtered min (plain load, atomic only on improvement),
// and the LDS form (rows already partitioned so that a workgroup's slots fit LDS).  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__host__ __device__ inline uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ void k_gen(uint32_t* slot, double* val, int64_t n, uint32_t G) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    slot[i] = (uint32_t)(mix((uint64_t)i ^ 0x5EED0001ull) % G);
    val[i] = (double)(mix((uint64_t)i + 0x5EED0002ull) >> 11) * 0x1.0p-53;
  }
}

__device__ inline uint64_t ord(double d) {
  uint64_t b = __double_as_longlong(d);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// 4 rows per thread per trip, 16-byte slot loads, grid-stride
template <int MODE>
__global__ __launch_bounds__(256) void k_atomic(const uint32_t* __restrict__ slot, const double* __restrict__ val, int64_t n, uint32_t* __restrict__ cnt,
                                                 unsigned long long* __restrict__ tab, unsigned long long* __restrict__ tab2) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    const uint4 s = *reinterpret_cast<const uint4*>(slot + i);
    uint32_t ss[4] = {s.x, s.y, s.z, s.w};
    double v[4];
    if (MODE != 0) {
      const double2 a = *reinterpret_cast<const double2*>(val + i), b = *reinterpret_cast<const double2*>(val + i + 2);
      v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE == 0) atomicAdd(&cnt[ss[j]], 1u);                                         // count
      if (MODE == 1) atomicAdd(&tab[ss[j]], (unsigned long long)__double_as_longlong(v[j]));  // int64 sum
      if (MODE == 2) atomicMin(&tab[ss[j]], (unsigned long long)ord(v[j]));              // min, unconditional
      if (MODE == 3) {                                                                   // min, filtered by a plain load
        const unsigned long long o = ord(v[j]);
        if (o < __builtin_nontemporal_load(&tab[ss[j]])) atomicMin(&tab[ss[j]], o);
      }
      if (MODE == 4) {                                                                   // min + max filtered + count
        const unsigned long long o = ord(v[j]);
        if (o < tab[ss[j]]) atomicMin(&tab[ss[j]], o);
        if (o > tab2[ss[j]]) atomicMax(&tab2[ss[j]], o);
        atomicAdd(&cnt[ss[j]], 1u);
      }
      if (MODE == 5) {                                                                   // plain random load only (gather rate)
        const unsigned long long o = tab[ss[j]];
        if (o == 0x1234567ull) cnt[0] = 1;
      }
      if (MODE == 6) {                                                                   // min + max filtered, no count
        const unsigned long long o = ord(v[j]);
        if (o < tab[ss[j]]) atomicMin(&tab[ss[j]], o);
        if (o > tab2[ss[j]]) atomicMax(&tab2[ss[j]], o);
      }
    }
  }
}

// LDS form: rows partitioned into B buckets of consecutive positions; bucket b's slots map to [0, S) local indexes (here: slot / B).
// One 1024-thread workgroup per (bucket, chunk): accumulate in LDS, flush with global atomics (chunks > 1) or plain stores.
template <int MODE>
__global__ __launch_bounds__(1024) void k_lds_acc(const uint16_t* __restrict__ key16, const double* __restrict__ val, const int64_t* __restrict__ bstart,
                                                  int S, int chunks, unsigned long long* __restrict__ tab, unsigned long long* __restrict__ tab2,
                                                  uint32_t* __restrict__ cnt, int B) {
  extern __shared__ unsigned long long lds[];
  unsigned long long* lmin = lds;
  unsigned long long* lmax = lds + S;
  const int b = blockIdx.x / chunks, c = blockIdx.x % chunks;
  for (int i = threadIdx.x; i < S; i += 1024) {
    lmin[i] = ~0ull;
    if (MODE == 1) lmax[i] = 0;
  }
  __syncthreads();
  const int64_t lo = bstart[b], hi = bstart[b + 1];
  const int64_t per = ((hi - lo + chunks - 1) / chunks + 3) & ~3ll;
  const int64_t s0 = lo + per * c, s1 = s0 + per < hi ? s0 + per : hi;
  for (int64_t i = s0 + threadIdx.x * 4; i < s1; i += 4096) {
    if (i + 4 <= s1 && ((i & 3) == 0)) {
      const uint2 kk = *reinterpret_cast<const uint2*>(key16 + i);
      const double2 a = *reinterpret_cast<const double2*>(val + i), bb = *reinterpret_cast<const double2*>(val + i + 2);
      const uint32_t k[4] = {kk.x & 0xFFFF, kk.x >> 16, kk.y & 0xFFFF, kk.y >> 16};
      const double v[4] = {a.x, a.y, bb.x, bb.y};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned long long o = ord(v[j]);
        atomicMin(&lmin[k[j]], o);
        if (MODE == 1) atomicMax(&lmax[k[j]], o);
      }
    } else {
      for (int64_t r = i; r < s1 && r < i + 4; ++r) {
        const unsigned long long o = ord(val[r]);
        atomicMin(&lmin[key16[r]], o);
        if (MODE == 1) atomicMax(&lmax[key16[r]], o);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < S; i += 1024) {
    const int64_t g = (int64_t)i * B + b;
    if (chunks == 1) {
      tab[g] = lmin[i];
      if (MODE == 1) tab2[g] = lmax[i];
    } else {
      if (lmin[i] != ~0ull) atomicMin(&tab[g], lmin[i]);
      if (MODE == 1 && lmax[i]) atomicMax(&tab2[g], lmax[i]);
    }
  }
}

__global__ void k_fill16(const uint32_t* slot, uint16_t* k16, int64_t n, int S) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) k16[i] = (uint16_t)(slot[i] % (uint32_t)S);
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000000ll;
  const uint32_t G = argc > 2 ? (uint32_t)atoll(argv[2]) : 1000000u;
  uint32_t *slot, *cnt;
  double* val;
  unsigned long long *tab, *tab2;
  CK(hipMalloc(&slot, n * 4 + 64));
  CK(hipMalloc(&val, n * 8 + 64));
  CK(hipMalloc(&cnt, (size_t)G * 4));
  CK(hipMalloc(&tab, (size_t)G * 8));
  CK(hipMalloc(&tab2, (size_t)G * 8));
  hipLaunchKernelGGL(k_gen, dim3(256 * 16), dim3(256), 0, 0, slot, val, n, G);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const char* names[] = {"count: atomicAdd u32", "int64 sum: atomicAdd u64", "min: atomicMin u64", "min: nt-load filter + atomicMin", "min+max filtered + count",
                         "random 8-B load only", "min+max filtered"};
  for (int wgs = 8; wgs <= 32; wgs *= 2)
    for (int mode = 0; mode < 7; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(cnt, 0, (size_t)G * 4, 0));
        CK(hipMemsetAsync(tab, mode == 1 ? 0 : 0xFF, (size_t)G * 8, 0));
        CK(hipMemsetAsync(tab2, 0, (size_t)G * 8, 0));
        CK(hipEventRecord(e0, 0));
        const dim3 grid(256 * wgs), blk(256);
        switch (mode) {
          case 0: hipLaunchKernelGGL(k_atomic<0>, grid, blk, 0, 0, slot, val, n, cnt, tab, tab2); break;
          case 1: hipLaunchKernelGGL(k_atomic<1>, grid, blk, 0, 0, slot, val, n, cnt, tab, tab2); break;
          case 2: hipLaunchKernelGGL(k_atomic<2>, grid, blk, 0, 0, slot, val, n, cnt, tab, tab2); break;
          case 3: hipLaunchKernelGGL(k_atomic<3>, grid, blk, 0, 0, slot, val, n, cnt, tab, tab2); break;
          case 4: hipLaunchKernelGGL(k_atomic<4>, grid, blk, 0, 0, slot, val, n, cnt, tab, tab2); break;
          case 5: hipLaunchKernelGGL(k_atomic<5>, grid, blk, 0, 0, slot, val, n, cnt, tab, tab2); break;
          case 6: hipLaunchKernelGGL(k_atomic<6>, grid, blk, 0, 0, slot, val, n, cnt, tab, tab2); break;
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("wgs/CU %2d  %-34s %8.3f ms  %7.1f Grows/s\n", wgs, names[mode], best, n / best * 1e-6);
      fflush(stdout);
    }
  // LDS form on synthetic partitioned input: B buckets, local key = random < S
  for (int B : {128, 64}) {
    const int S = (int)((G + B - 1) / B);
    uint16_t* key16;
    int64_t* bstart;
    CK(hipMalloc(&key16, n * 2 + 64));
    CK(hipMalloc(&bstart, (B + 1) * 8));
    std::vector<int64_t> hb(B + 1);
    for (int b = 0; b <= B; ++b) hb[b] = ((n / B) & ~3ll) * b;
    hb[B] = n;
    CK(hipMemcpy(bstart, hb.data(), (B + 1) * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fill16, dim3(4096), dim3(256), 0, 0, slot, key16, n, S);
    CK(hipDeviceSynchronize());
    for (int mode = 0; mode < 2; ++mode)
      for (int chunks : {2, 4, 8, 16}) {
        const size_t lds = (size_t)S * 8 * (mode == 1 ? 2 : 1);
        if (lds > 160 * 1024) continue;
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipMemsetAsync(tab, 0xFF, (size_t)G * 8, 0));
          CK(hipMemsetAsync(tab2, 0, (size_t)G * 8, 0));
          CK(hipEventRecord(e0, 0));
          if (mode == 0) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds_acc<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_lds_acc<0>, dim3(B * chunks), dim3(1024), lds, 0, key16, val, bstart, S, chunks, tab, tab2, cnt, B);
          } else {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds_acc<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_lds_acc<1>, dim3(B * chunks), dim3(1024), lds, 0, key16, val, bstart, S, chunks, tab, tab2, cnt, B);
          }
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        printf("LDS form B=%3d S=%5d chunks=%2d %s  %8.3f ms  (%.0f KB LDS)\n", B, S, chunks, mode ? "min+max" : "min    ", best, lds / 1024.0);
        fflush(stdout);
      }
    CK(hipFree(key16));
    CK(hipFree(bstart));
  }
  return 0;
}
