// flr_wave.hpp -- fused last digit, one WAVE per run (included by groupby.hip after SegOut / seg_to_f64 / flr_sqdev).
//
// Same job as k_flr_reduce (the final 6-bit sort pass and Arrow's leaf / binary-counter reduce in one kernel, 9 B/row read once),
// re-cut so that nothing in it waits on a workgroup barrier: after the LSD passes a run (equal low slot bits) holds <= 64 groups
// interleaved in row order, and 64 is the wave width -- so ONE wave owns a run from start to end, lane g owns group g, and every
// step below is ordered by the wave's own in-order LDS queue.  Per 512-row tile:
//   A  every row ORs its lane bit into match[step][digit]               (8 LDS atomics per lane, issued back to back, ONE wait --
//                                                                        the step-by-step form paid three LDS round trips per step)
//   B  lane d scans its digit's 8 match words: rows of digit d in front of every (half-)step, the digit's total, exclusive scan
//      over the 64 digits = staging offsets, written back as base[step][d][half]  (conflict-free: consecutive lanes and words)
//   C  every row: place = base[step][digit][half] + popcount(match[step][digit][half] & lower lanes)   (two 4-byte reads)
//   D  values (and null flags) staged in LDS in group-major, row order; the next tile's global loads are issued
//   E  lane g walks its group's rows in order: Arrow's 16-value sequential leaf, a null row closes the open leaf; min / max /
//      integer sum ride along; finished leaf sums are written back over the consumed rows
//   F  lane g pushes its finished leaves through Arrow's binary counter, one LDS column per lane.  ALL levels live in LDS (their
//      number comes from the longest run, via the launch's dynamic LDS size): a private array for the rare high levels went to
//      scratch memory, and a scratch access waits for vmcnt(0) -- i.e. for the next tile's loads still in flight -- on every push.
// No binary search for leaf owners, no cross-wave prefix, no idle waves during the pushes; nullable values, min / max and int64
// sums take the same path as the dense sum (the walk handles them), so a 5 %-null column no longer falls back to a per-lane
// replay that leaves three of four waves idle.
// LDS per wave: ~13.5 KB (dense, 13 counter levels) -> 12 waves per CU, limited by LDS, so up to 168 VGPRs are free.
#pragma once

namespace pdx {

constexpr int kFwItems = 8;                   // steps of 64 rows per tile
constexpr int kFwTile = 64 * kFwItems;        // 512 rows

// LDS carve-up of one wave (64-thread workgroup); all offsets in bytes, 8-byte aligned
constexpr int kFwStageBytes = (kFwTile + 64) * 8;                 // phases D-F: values as 64-bit patterns, group g's rows at dstart[g] + g
                                                                  // (skewed by one element per group so the 64 walking lanes spread over the
                                                                  // banks); phases A-C: the match words [kFwItems][64] alias its first 4 KB
constexpr int kFwBaseOff = kFwStageBytes;                         // base[kFwItems][64][2] uint32; at tile load: the 512 key bytes
constexpr int kFwNullOff = kFwBaseOff + kFwItems * 64 * 8;        // snull[kFwTile + 64] bytes (nullable instantiations only)
__host__ __device__ constexpr int fw_csum_off(bool nullable) { return nullable ? kFwNullOff + kFwTile + 64 : kFwNullOff; }
__host__ __device__ constexpr int fw_lds_bytes(bool nullable, int levels) { return fw_csum_off(nullable) + levels * 64 * 8; }

// Arrow's binary counter, one LDS column per lane
__device__ __forceinline__ void fw_counter_push(double* csum /* [levels][64] */, int lane, unsigned long long& cmask, int& root, double leaf) {
  int cur = 0;
  unsigned long long m = 1;
  double v = pw_merge(csum[lane], leaf);
  cmask ^= m;
  while ((cmask & m) == 0) {
    csum[cur * 64 + lane] = 0.0;
    ++cur;
    m <<= 1;
    v = pw_merge(csum[cur * 64 + lane], v);
    cmask ^= m;
  }
  csum[cur * 64 + lane] = v;
  root = cur > root ? cur : root;
}

// KT: uint8 (narrowing sort: the top digit alone, bit 7 = the value's null flag) or uint32 slots (digit at low_bits, bit 31 = null flag)
// levels: counter levels in LDS; the host derives it from the longest run (a group cannot outgrow its run)
template <typename T, typename KT, bool NULLABLE, bool PW_ONLY>
__global__ void __launch_bounds__(64) k_flr_wave(const KT* __restrict__ keys, const T* __restrict__ vals, int64_t n, const uint32_t* __restrict__ run_start,
                                                 int64_t nruns, int low_bits, const uint32_t* __restrict__ gid_of_slot, SegOut out,
                                                 uint8_t* __restrict__ ok, int want_pw, int want_mm, int want_is,
                                                 const double* __restrict__ sqdev_mean, int levels, unsigned int max_run) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fw_lds[];
  unsigned long long* const match = reinterpret_cast<unsigned long long*>(fw_lds);  // [kFwItems][64]: both half-step words of a digit
  uint32_t* const match32 = reinterpret_cast<uint32_t*>(fw_lds);                    // [kFwItems][64][2]
  unsigned long long* const stage = reinterpret_cast<unsigned long long*>(fw_lds);
  uint32_t* const base = reinterpret_cast<uint32_t*>(fw_lds + kFwBaseOff);           // [kFwItems][64][2]
  uint8_t* const keybytes = fw_lds + kFwBaseOff;
  uint8_t* const snull = fw_lds + kFwNullOff;
  double* const csum = reinterpret_cast<double*>(fw_lds + fw_csum_off(NULLABLE));
  const int lane = threadIdx.x;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  constexpr bool kByteKeys = sizeof(KT) == 1;
  for (int s = 0; s < kFwItems; ++s) match[s * 64 + lane] = 0ull;
  __builtin_amdgcn_wave_barrier();

  for (int64_t run = blockIdx.x; run < nruns; run += gridDim.x) {
    const int64_t rs = run_start[run], re = run_start[run + 1];
    if (rs == re || re - rs > (int64_t)max_run) continue;  // (longer runs: the layout's side form)
    // per-group state (lane = top digit)
    double acc = 0.0, mu = 0.0;
    int pos = 0, root = 0;
    unsigned long long cmask = 0, isum = 0;
    long long nvalid = 0, nrows = 0;
    T vmn = T(0), vmx = T(0);
    int zneg = -1;
    bool has = false, mu_known = false;
    for (int l = 0; l < levels; ++l) csum[l * 64 + lane] = 0.0;

    // tiles start at multiples of 64 rows (absolute), so every load below is aligned; rows in front of the run are inactive
    const int64_t a0 = rs & ~(int64_t)63;
    T val[kFwItems];
    uint32_t kraw[kFwItems];
    uint4 kvec = make_uint4(0, 0, 0, 0);
    auto issue_loads = [&](int64_t t0) {
#pragma unroll
      for (int s = 0; s < kFwItems; ++s) {
        const int64_t r = t0 + s * 64 + lane;
        val[s] = (r >= rs && r < re) ? vals[r] : T(0);
      }
      if constexpr (kByteKeys) {
        // 512 key bytes = 32 lanes x 16 bytes (t0 is a multiple of 64; bytes outside [rs, re) belong to the neighbouring runs of the
        // same n-byte key array, except at its very end, where the tail is fetched byte by byte)
        if (lane < kFwTile / 16) {
          const int64_t b = t0 + (int64_t)lane * 16;
          if (b + 16 <= n) kvec = *reinterpret_cast<const uint4*>(keys + b);
          else {
            uint32_t w[4] = {0, 0, 0, 0};
            for (int q = 0; q < 16; ++q)
              if (b + q < re) w[q >> 2] |= (uint32_t)keys[b + q] << (8 * (q & 3));
            kvec = make_uint4(w[0], w[1], w[2], w[3]);
          }
        }
      } else {
#pragma unroll
        for (int s = 0; s < kFwItems; ++s) {
          const int64_t r = t0 + s * 64 + lane;
          kraw[s] = (r >= rs && r < re) ? (uint32_t)keys[r] : 0u;
        }
      }
    };
    issue_loads(a0);

    for (int64_t t0 = a0; t0 < re; t0 += kFwTile) {
      // ---- this tile's digits / flags into registers
      uint32_t dig[kFwItems];
      uint32_t nullbits = 0, active = 0;
      if constexpr (kByteKeys) {
        if (lane < kFwTile / 16) reinterpret_cast<uint4*>(keybytes)[lane] = kvec;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < kFwItems; ++s) kraw[s] = keybytes[s * 64 + lane];
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int s = 0; s < kFwItems; ++s) {
        const int64_t r = t0 + s * 64 + lane;
        const bool act = r >= rs && r < re;
        active |= (uint32_t)act << s;
        if constexpr (kByteKeys) {
          dig[s] = kraw[s] & 63u;
          if (NULLABLE) nullbits |= ((kraw[s] >> 7) & 1u) << s;
        } else {
          dig[s] = ((kraw[s] & kSortKeyMask) >> low_bits) & 63u;
          if (NULLABLE) nullbits |= (kraw[s] >> 31) << s;
        }
      }
      // ---- A: match-any, all steps at once.  The 64 rows of a step are two half-steps of 32: a row ORs bit (lane & 31) into the
      // 32-bit word match[step][digit][lane >> 5] -- 4-byte LDS atomics and 4-byte reads in phase C suffer about half the bank
      // conflicts of 8-byte ones, and the owner lane still fetches / clears both halves of a digit with one 8-byte access
#pragma unroll
      for (int s = 0; s < kFwItems; ++s)
        if ((active >> s) & 1u)
          __hip_atomic_fetch_or(&match32[(s * 64 + dig[s]) * 2 + (lane >> 5)], 1u << (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // ---- B: lane d = digit d: rows of the digit in front of every half-step, total, staging offsets (conflict-free accesses)
      uint32_t tot = 0;
      uint32_t before[2 * kFwItems];
      {
        unsigned long long w[kFwItems];
#pragma unroll
        for (int s = 0; s < kFwItems; ++s) w[s] = __hip_atomic_load(&match[s * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#pragma unroll
        for (int s = 0; s < kFwItems; ++s) {
          before[2 * s] = tot;
          tot += (uint32_t)__popc((uint32_t)w[s]);
          before[2 * s + 1] = tot;
          tot += (uint32_t)__popc((uint32_t)(w[s] >> 32));
        }
      }
      const uint32_t inc = wave_inclusive_scan(tot, SumOp());
      const uint32_t my_start = inc - tot + (uint32_t)lane;  // skewed start of group `lane`
#pragma unroll
      for (int s = 0; s < kFwItems; ++s)  // base[step][digit][half] = staging position of the digit's first row of that half-step
        reinterpret_cast<uint2*>(base)[s * 64 + lane] = make_uint2(my_start + before[2 * s], my_start + before[2 * s + 1]);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // ---- C: every row's place in group-major order: two 4-byte reads
      uint32_t place[kFwItems];
      const uint32_t lt32 = (1u << (lane & 31)) - 1u;
#pragma unroll
      for (int s = 0; s < kFwItems; ++s) {
        const int e = (s * 64 + (int)dig[s]) * 2 + (lane >> 5);
        const uint32_t peers = __hip_atomic_load(&match32[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        place[s] = base[e] + (uint32_t)__popc(peers & lt32);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the match words are dead: the staging area may overwrite them
      // ---- D: stage
#pragma unroll
      for (int s = 0; s < kFwItems; ++s)
        if ((active >> s) & 1u) {
          unsigned long long bits;
          __builtin_memcpy(&bits, &val[s], 8);
          stage[place[s]] = bits;
          if (NULLABLE) snull[place[s]] = (uint8_t)((nullbits >> s) & 1u);
        }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (t0 + kFwTile < re) issue_loads(t0 + kFwTile);  // in flight during the walk
      // ---- E: lane g walks its rows
      const int c = (int)tot;
      nrows += c;
      if (sqdev_mean && c > 0 && !mu_known) {
        mu = sqdev_mean[gid_of_slot[((uint32_t)lane << low_bits) | (uint32_t)run]];
        mu_known = true;
      }
      const int i0 = (int)my_start;
      int nleaf = 0;
      for (int ib = 0; ib < c; ib += 8) {
        unsigned long long xb[8];
        uint8_t nb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = ib + u < c ? ib + u : c - 1;
          xb[u] = stage[i0 + i];
          nb[u] = NULLABLE ? snull[i0 + i] : (uint8_t)0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (ib + u >= c) break;
          T x;
          __builtin_memcpy(&x, &xb[u], 8);
          bool close = false;
          if (!NULLABLE || nb[u] == 0) {
            ++nvalid;
            if (PW_ONLY || want_pw) {
              acc = pw_leaf_add(pos == 0 ? 0.0 : acc, sqdev_mean ? flr_sqdev(seg_to_f64(x), mu) : seg_to_f64(x));
              close = ++pos == 16;
            }
            if constexpr (!PW_ONLY) {
              if (want_is) isum += (unsigned long long)x;
              if (want_mm && x == x) {
                if (!has) { vmn = vmx = x; has = true; }
                else {
                  if (x < vmn) vmn = x;
                  if (x > vmx) vmx = x;
                }
                if constexpr (__is_same(T, double)) {
                  if (x == 0.0) zneg = __double_as_longlong(x) < 0 ? 1 : 0;  // the LAST zero of the group (minmax.hpp)
                }
              }
            }
          } else {
            close = (PW_ONLY || want_pw) && pos > 0;  // a null row closes the open leaf
          }
          if (close) {
            reinterpret_cast<double*>(stage)[i0 + nleaf++] = acc;  // over a row already consumed (a leaf has >= 1 row)
            pos = 0;
          }
        }
      }
      // ---- F: the finished leaves through the counter
      for (int j = 0; j < nleaf; ++j) fw_counter_push(csum, lane, cmask, root, reinterpret_cast<const double*>(stage)[i0 + j]);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // the staging area becomes the (all-zero) match words of the next tile
#pragma unroll
      for (int s = 0; s < kFwItems; ++s) match[s * 64 + lane] = 0ull;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }

    if (nrows > 0) {
      const uint32_t slot = ((uint32_t)lane << low_bits) | (uint32_t)run;
      const uint32_t oi = gid_of_slot[slot];
      if (PW_ONLY || want_pw) {
        if (pos > 0) fw_counter_push(csum, lane, cmask, root, acc);
        double total = 0.0;
        if (nvalid > 0) {
          double a = csum[lane];
          for (int i = 1; i <= root; ++i) a = pw_merge(csum[i * 64 + lane], a);
          total = a;
        }
        if (out.sum_f) out.sum_f[oi] = total;
        if (out.mean) out.mean[oi] = nvalid ? pw_mean(total, (double)nvalid) : 0.0;
      }
      if constexpr (!PW_ONLY) {
        if (want_is && out.sum_i) out.sum_i[oi] = (long long)isum;
        if (want_mm) {
          T nanv = T(0);
          if constexpr (__is_same(T, double)) {
            nanv = __builtin_nan("");
            if (has && zneg >= 0 && vmx == 0.0 && nvalid < nrows) vmx = zneg ? -0.0 : 0.0;  // a group WITH nulls keeps the last tied zero
          }
          if (out.vmin) static_cast<T*>(out.vmin)[oi] = has ? vmn : nanv;
          if (out.vmax) static_cast<T*>(out.vmax)[oi] = has ? vmx : nanv;
        }
      }
      if (out.count) out.count[oi] = nvalid;
      if (ok) ok[oi] = nvalid > 0;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace pdx
