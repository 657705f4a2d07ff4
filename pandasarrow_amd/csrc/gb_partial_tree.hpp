// gb_partial_tree.hpp -- part of groupby.hip (textually included there, in its namespace context; split by stage, kernels unchanged):
// exact multi-GPU fp64 sum: partial-tree exchange (pdx_grouped_*, pdx_replay_partials).
#pragma once

// =====================================================================================================================
// Exact multi-GPU fp64 sum: partial-tree exchange (see include/pdx/abi.h).  Thread-per-group kernels: every group's values
// are contiguous (grouped), a thread walks its group once.  Uncoalesced but short: c ~ rows/group/rank.
// =====================================================================================================================
namespace pdx {

// (aligned_block_level / partial_record_count: pairwise.hpp, shared with the owners' replay in dist.hip)

__global__ void k_grouped_counts(const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ gid_of_occ, int64_t G,
                                 int64_t* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < G; k += stride)
    out[gid_of_occ[k]] = (int64_t)seg_start[k + 1] - (int64_t)seg_start[k];
}
// emission order: record block j belongs to local group order[j] (order == nullptr: group j); occ_of_gid maps a local group id to
// its position in slot order (where its values live)
__global__ void k_occ_of_gid(const uint32_t* __restrict__ gid_of_occ, int64_t G, uint32_t* __restrict__ occ_of_gid) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < G; k += stride) occ_of_gid[gid_of_occ[k]] = (uint32_t)k;
}
__global__ void k_partial_plan(const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ occ_of_gid, const int64_t* __restrict__ order,
                               int64_t G, const int64_t* __restrict__ prefix, int64_t* __restrict__ rec_cnt) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < G; j += stride) {
    const int64_t lg = order ? order[j] : j;
    const uint32_t k = occ_of_gid[lg];
    rec_cnt[j] = partial_record_count(prefix[lg], (int64_t)seg_start[k + 1] - (int64_t)seg_start[k]);
  }
}
__global__ void __launch_bounds__(256) k_partial_fill(const double* __restrict__ vals, const uint32_t* __restrict__ seg_start,
                                                      const uint32_t* __restrict__ occ_of_gid, const int64_t* __restrict__ order, int64_t G,
                                                      const int64_t* __restrict__ prefix, const int64_t* __restrict__ gid_map,
                                                      const int64_t* __restrict__ rec_off, int64_t* __restrict__ rec_key,
                                                      double* __restrict__ rec_val, int wave_form_too) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < G; j += stride) {
    const int64_t lg = order ? order[j] : j;
    const uint32_t k = occ_of_gid[lg];
    const double* v = vals + seg_start[k];
    const int64_t c = (int64_t)seg_start[k + 1] - (int64_t)seg_start[k];
    if (c <= 0) continue;
    const int64_t a = prefix[lg], b = a + c;
    const int64_t kf = (a + 15) >> 4, kl = b >> 4;
    if (wave_form_too && (kf > kl || kl - kf <= 64)) continue;  // k_partial_fill_wave emits this group
    const int64_t gkey = gid_map[lg] * 64;
    int64_t pos = rec_off[j];
    if (kf > kl) {
      for (int64_t i = 0; i < c; ++i) { if (rec_key) rec_key[pos] = gkey; rec_val[pos] = v[i]; ++pos; }
      continue;
    }
    const int64_t h = 16 * kf - a;
    for (int64_t i = 0; i < h; ++i) { if (rec_key) rec_key[pos] = gkey; rec_val[pos] = v[i]; ++pos; }
    for (int64_t sidx = kf; sidx < kl;) {
      const int lvl = (int)aligned_block_level(sidx, kl);
      const int64_t nleaf = (int64_t)1 << lvl;
      // perfect tree over leaves [sidx, sidx + nleaf): replay the counter, all merges happen inside the block
      PairwiseCounter cn;
      cn.init();
      const double* lv = v + (16 * sidx - a);
      for (int64_t q = 0; q < nleaf; ++q) {
        double acc = 0.0;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc += lv[q * 16 + e];
        if (acc != acc) acc = pw_leaf_redo(16, [&](int e) { return lv[q * 16 + e]; });
        cn.push(acc, 0);
      }
      if (rec_key) rec_key[pos] = gkey + lvl + 1;
      rec_val[pos] = cn.sum[lvl];
      ++pos;
      sidx += nleaf;
    }
    if (b > 16 * kl) {  // the rows that begin the last leaf: their sequential sum
      double acc = 0.0;
      for (int64_t i = 16 * kl - a; i < c; ++i) acc = pw_leaf_add(acc, v[i]);
      if (rec_key) rec_key[pos] = gkey + kPartialLeafCode + (b - 16 * kl);
      rec_val[pos] = acc;
    }
  }
}

// Wave-per-group form of k_partial_fill for groups with at most 64 interior leaves (<= ~1050 rows; longer ones keep the thread form): lane l
// sums interior leaf l (16 contiguous values), six shuffle steps build every aligned perfect subtree at once (t[k] at lane r = the tree
// over leaves [r, r + 2^k), left + right as the counter merges them), the boundary-leaf fragments are copied by the lanes.  The thread
// form walks each group with one thread and a 64-entry counter in scratch memory (2.2 ms per 5e8 rows).
// LPG lanes per group: 64 (one wave per group, <= 64 interior leaves), or 16 -- four groups per wave for the short groups of a sharded
// run (8 GPUs x 1e6 keys: ~125 rows = 6-8 interior leaves per group and rank; the 64-lane form left 7/8 of every wave idle: 0.71 ms per
// 1.25e8 rows).  A group whose interior leaves do not fit LPG lanes is left to the wider form (LPG = 16: `wide_too`) or to the thread form.
template <int LPG>
__global__ void __launch_bounds__(256) k_partial_fill_wave(const double* __restrict__ vals, const uint32_t* __restrict__ seg_start,
                                                           const uint32_t* __restrict__ occ_of_gid, const int64_t* __restrict__ order, int64_t G,
                                                           const int64_t* __restrict__ prefix, const int64_t* __restrict__ gid_map,
                                                           const int64_t* __restrict__ rec_off, int64_t* __restrict__ rec_key,
                                                           double* __restrict__ rec_val) {
  static_assert(LPG == 16 || LPG == 64, "head / tail fragments (<= 15 rows) take one lane each");
  constexpr int kLevels = LPG == 64 ? 6 : 4, kPerBlock = 256 / LPG;
  const int lane = threadIdx.x & (LPG - 1);
  const int64_t nw = (int64_t)gridDim.x * kPerBlock;
  for (int64_t j = (int64_t)blockIdx.x * kPerBlock + (threadIdx.x / LPG); j < G; j += nw) {
    const int64_t lg = order ? order[j] : j;
    const uint32_t k = occ_of_gid[lg];
    const double* v = vals + seg_start[k];
    const int64_t c = (int64_t)seg_start[k + 1] - (int64_t)seg_start[k];
    if (c <= 0) continue;
    const int64_t a = prefix[lg], b = a + c;
    const int64_t kf = (a + 15) >> 4, kl = b >> 4;
    const int64_t gkey = gid_map[lg] * 64;
    const int64_t pos0 = rec_off[j];
    if (kf > kl) {  // the whole range lies inside one leaf (< 31 rows): fragments only
      if (LPG == 64) continue;  // (emitted by the 16-lane form, which always runs)
      for (int64_t i = lane; i < c; i += LPG) {
        if (rec_key) rec_key[pos0 + i] = gkey;
        rec_val[pos0 + i] = v[i];
      }
      continue;
    }
    const int nint = (int)(kl - kf);
    if (kl - kf > LPG) continue;                    // wider form / long group: k_partial_fill
    if (LPG == 64 && kl - kf <= 16) continue;       // (the 16-lane form's)
    const int h = (int)(16 * kf - a);
    if (lane < h) {
      if (rec_key) rec_key[pos0 + lane] = gkey;
      rec_val[pos0 + lane] = v[lane];
    }
    double t[kLevels + 1];
    t[0] = 0.0;
    if (lane < nint) {
      const double* lv = v + h + 16 * lane;
      double acc = 0.0;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc += lv[e];
      if (acc != acc) acc = pw_leaf_redo(16, [&](int e) { return lv[e]; });
      t[0] = acc;
    }
    // (a source lane outside the group's LPG lanes hands back the reader's own value: those subtrees reach past the last leaf and are
    //  never emitted.  All lanes of the wave take part in the shuffles of their own group: the loop above is left only by whole groups,
    //  and __shfl_down with a width reads inside the width-aligned lane group)
#pragma unroll
    for (int q = 0; q < kLevels; ++q) t[q + 1] = pw_merge(t[q], __shfl_down(t[q], 1 << q, LPG));
    int64_t pos = pos0 + h;
    for (int64_t sidx = kf; sidx < kl;) {
      const int lvl = (int)aligned_block_level(sidx, kl);
      if (lane == (int)(sidx - kf)) {
        double val = t[0];
#pragma unroll
        for (int q = 1; q <= kLevels; ++q) val = lvl == q ? t[q] : val;
        if (rec_key) rec_key[pos] = gkey + lvl + 1;
        rec_val[pos] = val;
      }
      ++pos;
      sidx += (int64_t)1 << lvl;
    }
    const int ntail = (int)(b - 16 * kl);
    if (lane == 0 && ntail > 0) {  // the rows that begin the last leaf: their sequential sum, one record
      double acc = 0.0;
      for (int q = 0; q < ntail; ++q) acc = pw_leaf_add(acc, v[16 * kl - a + q]);
      if (rec_key) rec_key[pos] = gkey + kPartialLeafCode + ntail;
      rec_val[pos] = acc;
    }
  }
}

// ---------------------------------------------------------------- fused last digit that EMITS partial-tree records (sharded sums)
// The dense path of k_flr_reduce with another ending: instead of one sum per group, the local share of every group leaves as the records
// of the partial-tree exchange (gb_partial_tree.hpp) -- the rows of the group's first and last global leaf as raw values, the leaves in
// between as the aligned blocks of Arrow's tree over the group's GLOBAL row numbering.  a = rows of the group on lower ranks: the open
// leaf starts with a % 16 virtual rows, so the leaf grid of the staging is the global one; the counter starts at kf = ceil(a / 16) with
// those bits VIRTUAL: a finished node whose left sibling is virtual is an "orphan" (a right child whose left sibling lives on a lower
// rank) and is emitted as it stands; orphans come out in ascending level = row order, then the pending nodes in descending level.
// Replaces the third sort pass + k_partial_fill of the sharded group-by (25 B/row) by this kernel's 9 B/row.
constexpr int kEmLevels = 16;  // real nodes only: a run of <= 2^19 rows (kFlrMaxRun, checked by the host) holds <= 2^15 leaves
__global__ void __launch_bounds__(kSortBlock, 3) k_flr_emit(const uint8_t* __restrict__ keys, const double* __restrict__ vals, const uint32_t* __restrict__ run_start,
                                                         int64_t nruns, int low_bits, const uint32_t* __restrict__ gid_of_slot, int64_t nslots, int64_t G,
                                                         const int64_t* __restrict__ prefix, const uint32_t* __restrict__ seg_start,
                                                         const uint32_t* __restrict__ occ_of_gid, const uint32_t* __restrict__ occ_slot,
                                                         const int64_t* __restrict__ rec_off_lg,
                                                         const int64_t* __restrict__ gid_map, int64_t* __restrict__ rec_key, double* __restrict__ rec_val) {
  constexpr int R = 1 << kFlrBits;
  __shared__ __attribute__((aligned(16))) double svals[16 * kFlrLeafStride];
  __shared__ __attribute__((aligned(8))) double leafsum[kFlrMaxLeaves + 1];
  __shared__ uint32_t cnt[kSortWaves][R];
  __shared__ unsigned long long match[kSortWaves][R];
  __shared__ double csum[kEmLevels][R];
  __shared__ double open_acc[R];
  __shared__ int lp[R + 1];
  __shared__ uint8_t leaf_d[kFlrLeafStride + 3];
  __shared__ uint32_t dinfo[R];
  // per group (digit) of the run and tile: the leaf coordinates [lo, hi) of its rows that are NOT raw records (lo | hi << 16; rows of the
  // group's first and last global leaf leave raw), and what a raw row needs: rows before the tile, head rows, first tail row, offsets
  __shared__ uint32_t g_mid[R], g_before[R], g_head[R], g_tail0[R];
  __shared__ int g_tailbase[R];
  __shared__ long long g_roff[R], g_gkey[R];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  for (int d = tid; d < kSortWaves * R; d += kSortBlock) {
    (&cnt[0][0])[d] = 0;
    (&match[0][0])[d] = 0;
  }
  __syncthreads();
  for (int64_t run = blockIdx.x; run < nruns; run += gridDim.x) {
    const int64_t s = run_start[run], e = run_start[run + 1];
    if (s == e) continue;
    // ---- per-group parameters (wave 0, lane = top digit)
    int pos = 0;
    unsigned long long cmask = 0, vmask = 0;
    long long nrows = 0, my_roff = 0, my_gkey = 0, rec_at = 0;
    uint32_t my_head = 0, my_tail0 = 0;
    int my_ntail = 0;
    bool head_partial = false, first_leaf_seen = false, live = false;
    int tile_c = 0, tile_lp = 0;
    if (wave == 0) {
      const uint32_t slot = ((uint32_t)lane << low_bits) | (uint32_t)run;
      // (a digit beyond the handle's slots -- the top digit of a slot space that is not a power of two -- has no rows and no entry)
      const uint32_t lg = (int64_t)slot < nslots ? gid_of_slot[slot] : 0xFFFFFFFFu;
      long long a = 0, C = 0;
      // (gid_of_slot is only defined on occupied slots: the entry counts when the group it names sits on this very slot)
      if ((int64_t)lg < G && occ_slot[occ_of_gid[lg]] == slot) {
        const uint32_t k = occ_of_gid[lg];
        C = (long long)seg_start[k + 1] - (long long)seg_start[k];
        a = prefix[lg];
        my_roff = rec_off_lg[lg];
        my_gkey = gid_map[lg] * 64;
        live = C > 0;
      }
      const long long b = a + C, kf = (a + 15) >> 4, kl = b >> 4;
      const bool allraw = kf > kl;
      const long long h = allraw ? C : 16 * kf - a, ntail = allraw ? 0 : b - 16 * kl;
      long long nnodes = 0;
      if (!allraw)
        for (long long sidx = kf; sidx < kl; ++nnodes) sidx += 1ll << aligned_block_level(sidx, kl);
      pos = (int)(a & 15);
      head_partial = pos != 0 || allraw;  // the first leaf this group finishes is not a node (its rows leave raw)
      cmask = vmask = allraw ? 0ull : (unsigned long long)kf;
      my_head = (uint32_t)h;
      my_tail0 = 0xFFFFFFFFu;  // (the rows that begin the last leaf stay inside: the open leaf's sum is their record)
      my_ntail = (int)ntail;
      rec_at = my_roff + h;  // the next node record (orphans leave as they arise, the pending nodes at the end of the run)
      g_head[lane] = my_head;
      g_tail0[lane] = my_tail0;
      g_tailbase[lane] = 0;
      (void)nnodes;
      g_roff[lane] = my_roff;
      g_gkey[lane] = my_gkey;
      open_acc[lane] = 0.0;
#pragma unroll
      for (int l = 0; l < kEmLevels; ++l) csum[l][lane] = 0.0;
    }
    uint32_t key[kFlrItems];
    double val[kFlrItems];
    auto load_tile = [&](int64_t t0) {
      const int rows = (int)(e - t0 < kFlrTile ? e - t0 : kFlrTile);
#pragma unroll
      for (int q = 0; q < kFlrItems; ++q) {
        const int r = wave * (64 * kFlrItems) + q * 64 + lane;
        const int rr = r < rows ? r : rows - 1;
        key[q] = keys[t0 + rr];
        val[q] = vals[t0 + rr];
      }
    };
    load_tile(s);
    for (int64_t t0 = s; t0 < e; t0 += kFlrTile) {
      const int rows = (int)(e - t0 < kFlrTile ? e - t0 : kFlrTile);
      uint32_t rank[kFlrItems];
#pragma unroll
      for (int q = 0; q < kFlrItems; ++q) {
        const int r = wave * (64 * kFlrItems) + q * 64 + lane;
        rank[q] = wave_match_rank(match[wave], cnt[wave], key[q] & (R - 1), r < rows, lane, lt_mask);
      }
      __syncthreads();
      if (tid < R) {
        uint32_t cw[kSortWaves];
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) cw[w] = cnt[w][tid];
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
          const uint32_t c = cw[w];
          cw[w] = tot;
          tot += c;
        }
        const int c = (int)tot;
        const uint32_t nl = c > 0 ? (uint32_t)(pos + c + 15) >> 4 : 0u;
        const uint32_t both = wave_inclusive_scan(tot | (nl << 16), SumOp());
        const uint32_t lincl = both >> 16, lex = lincl - nl;
        const uint32_t base = 16u * lex + (uint32_t)pos;
        tile_c = c;
        tile_lp = (int)lex;
        dinfo[tid] = lex | ((uint32_t)pos << 12) | ((uint32_t)c << 16);
        if (c > 0) leaf_d[lex] = (uint8_t)tid;
        if (tid == R - 1) lp[R] = (int)lincl;
        // rows of this tile with group-local index in [head, tail0) stay inside: in leaf coordinates (base + index - rows before the tile)
        const long long before = nrows;
        long long lo = (long long)base + ((long long)my_head - before), hi = (long long)base + ((long long)my_tail0 - before);
        lo = lo < 0 ? 0 : (lo > 65535 ? 65535 : lo);
        hi = hi < 0 ? 0 : (hi > 65535 ? 65535 : hi);
        g_mid[tid] = (uint32_t)lo | ((uint32_t)hi << 16);
        g_before[tid] = (uint32_t)before;
        nrows += c;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) cnt[w][tid] = cw[w] + base;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < kFlrItems; ++q) rank[q] += cnt[wave][key[q] & (R - 1)];
      uint32_t mid[kFlrItems];
#pragma unroll
      for (int q = 0; q < kFlrItems; ++q) mid[q] = g_mid[key[q] & (R - 1)];
#pragma unroll
      for (int q = 0; q < kFlrItems; ++q) {
        const int r = wave * (64 * kFlrItems) + q * 64 + lane;
        if (r < rows) {
          const uint32_t d = key[q] & (R - 1);
          const uint32_t p = rank[q];
          if ((p & 15u) == 0) leaf_d[p >> 4] = (uint8_t)d;  // this row opens a leaf
          svals[((p & 14u) >> 1) * (2 * kFlrLeafStride) + (p >> 4) * 2 + (p & 1u)] = val[q];
          if (p < (mid[q] & 0xFFFFu) || p >= (mid[q] >> 16)) {  // a row of the group's first or last global leaf: a raw record
            const uint32_t di = dinfo[d];
            const uint32_t i = g_before[d] + (p - (16u * (di & 0xFFFu) + ((di >> 12) & 15u)));
            const long long idx = i < g_head[d] ? (long long)i : (long long)i + g_tailbase[d];
            const long long at = g_roff[d] + idx;
            if (rec_key) rec_key[at] = g_gkey[d];
            rec_val[at] = val[q];
          }
        }
      }
      if (t0 + kFlrTile < e) load_tile(t0 + kFlrTile);
      __syncthreads();
      for (int d = tid; d < kSortWaves * R; d += kSortBlock) (&cnt[0][0])[d] = 0;
      {
        const int NL = lp[R];
        for (int Lf = tid; Lf < NL; Lf += kSortBlock) {
          const int d = leaf_d[Lf];
          double xs[16];
#pragma unroll
          for (int el = 0; el < 8; ++el) {
            struct alignas(16) Pair { double x, y; };
            const Pair pr = *reinterpret_cast<const Pair*>(svals + el * (2 * kFlrLeafStride) + 2 * Lf);
            xs[2 * el] = pr.x;
            xs[2 * el + 1] = pr.y;
          }
          const uint32_t di = dinfo[d];
          const double oa = open_acc[d];
          const int j = Lf - (int)(di & 0xFFFu), p0 = (int)((di >> 12) & 15u), c = (int)(di >> 16);
          const int q0 = j == 0 ? p0 : 0;
          int q1 = p0 + c - 16 * j;
          q1 = q1 < 16 ? q1 : 16;
          double a = (j == 0 && p0 > 0) ? oa : 0.0;
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if (q >= q0 && q < q1) a += xs[q];
          if (a != a) {
            a = (j == 0 && p0 > 0) ? oa : 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q)
              if (q >= q0 && q < q1) a = pw_leaf_add(a, xs[q]);
          }
          leafsum[Lf] = a;
        }
      }
      __syncthreads();
      if (wave == 0 && tile_c > 0) {
        const int p0 = pos, c = tile_c;
        const int nl = (p0 + c + 15) >> 4, nfull = (p0 + c) >> 4;
        const double* lf = leafsum + tile_lp;
        const double last = lf[nl - 1];
        for (int j = 0; j < nfull; ++j) {
          if (!first_leaf_seen && head_partial) {  // the group's first global leaf, incomplete on this rank: its rows left raw
            first_leaf_seen = true;
            continue;
          }
          first_leaf_seen = true;
          // push at the global leaf index: real pending sibling -> merge and carry; virtual sibling -> this node is an orphan (a record,
          // at once: orphans arise in row order), the carry is virtual; free level -> the node (or the virtual carry) waits there
          double v = lf[j];
          bool real = true;
          int cur = 0;
          unsigned long long m = 1ull;
          for (;;) {
            if (!(cmask & m)) {
              cmask |= m;
              if (real) csum[cur][lane] = v;
              else vmask |= m;
              break;
            }
            if (vmask & m) {
              if (real) {
                if (rec_key) rec_key[rec_at] = my_gkey + cur + 1;
                rec_val[rec_at] = v;
                ++rec_at;
                real = false;
              }
              vmask &= ~m;
            } else {
              v = pw_merge(csum[cur][lane], v);  // (a virtual carry never meets a real pending node: that node would begin before kf)
            }
            cmask &= ~m;
            ++cur;
            m <<= 1;
          }
        }
        const int rem = (p0 + c) & 15;
        pos = rem;
        if (rem) open_acc[lane] = last;
      }
    }
    if (wave == 0 && live) {
      const unsigned long long pend = cmask & ~vmask;
      for (int l = kEmLevels - 1; l >= 0; --l)
        if ((pend >> l) & 1ull) {
          if (rec_key) rec_key[rec_at] = my_gkey + l + 1;
          rec_val[rec_at] = csum[l][lane];
          ++rec_at;
        }
      if (my_ntail > 0) {  // the open leaf: the rows behind the last leaf boundary, summed in order by the leaf phase
        if (rec_key) rec_key[rec_at] = my_gkey + kPartialLeafCode + my_ntail;
        rec_val[rec_at] = open_acc[lane];
      }
    }
    __syncthreads();
  }
}
// record offsets of the plan (emission order j, local group order[j]) by local group id
__global__ void k_rec_off_by_group(const int64_t* __restrict__ rec_off, const int64_t* __restrict__ order, int64_t G, int64_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < G; j += stride) out[order ? order[j] : j] = rec_off[j];
}
__global__ void k_replay_keys(const int64_t* __restrict__ rec_key, int64_t m, int64_t gid_lo, int64_t n_own, uint32_t* __restrict__ slot,
                              uint32_t* __restrict__ lvl, unsigned int* __restrict__ bad, int pack_shift) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
    int64_t key = rec_key[i];
    int64_t g = (key >> 6) - gid_lo;
    if (g < 0 || g >= n_own) {
      atomicExch(bad, 1u);
      g = 0;
    }
    // (pack_shift >= 0: the level rides above the slot bits of the sort key -- the sort only looks at the low bits)
    slot[i] = pack_shift >= 0 ? (uint32_t)g | ((uint32_t)(key & 63) << pack_shift) : (uint32_t)g;
    lvl[i] = (uint32_t)(key & 63);
  }
}
// lvl_shift >= 0: `lvl` holds the sorted keys with the record's level packed above bit lvl_shift (one sort carries it along)
__global__ void __launch_bounds__(256) k_replay(const uint32_t* __restrict__ seg_start, int64_t n_own, const double* __restrict__ val,
                                                const uint32_t* __restrict__ lvl, double* __restrict__ out, unsigned int* __restrict__ bad,
                                                int lvl_shift) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_own; g += stride) {
    PairwiseCounter cn;
    cn.init();
    double acc = 0.0;
    int fill = 0;
    bool any = false;
    for (int64_t i = seg_start[g]; i < (int64_t)seg_start[g + 1]; ++i) {
      uint32_t l = lvl_shift >= 0 ? (lvl[i] & kSortKeyMask) >> lvl_shift : lvl[i];
      any = true;
      if (l == 0) {  // fragment value: extend the running 16-value leaf
        acc = pw_leaf_add(acc, val[i]);
        if (++fill == 16) {
          cn.push(acc, 0);
          acc = 0.0;
          fill = 0;
        }
      } else if (l > (uint32_t)kPartialLeafCode) {  // the first l - 32 rows of a leaf, already summed in order
        if (fill != 0) atomicExch(bad, 2u);
        acc = val[i];
        fill = (int)l - kPartialLeafCode;
      } else {
        if (fill != 0) atomicExch(bad, 2u);  // a node must start on a leaf boundary
        cn.push(val[i], (int)l - 1);
      }
    }
    if (fill) cn.push(acc, 0);
    if (!any) atomicExch(bad, 3u);
    out[g] = any ? cn.finish() : 0.0;
  }
}
__global__ void k_iota_u32(uint32_t* p, int64_t n) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = (uint32_t)i;
}

}  // namespace pdx

struct pdx_grouped {
  pdx_groupby* gb = nullptr;
  int64_t n = 0, G = 0;
  const double* vals_sorted = nullptr;
  uint32_t* seg_start = nullptr;  // G + 1, slot (occ) order
  uint32_t* occ_of_gid = nullptr; // G: local group id -> position in slot order
  int64_t* rec_off = nullptr;     // G + 1 after plan (emission order)
  const int64_t* prefix = nullptr;
  const int64_t* order = nullptr;
  int64_t total = -1;
  // fused record emission (k_flr_emit): the values sorted by the low slot bits only; vals_sorted is filled by the third pass on demand
  bool two_pass = false;
  NarrowTwo two{};
  unsigned int* dmax = nullptr;      // device: rows of the longest run (+ two unused words of k_run_max_len)
  unsigned int hmax[3] = {0, 0, 0};  // read back with the plan's total
  mutable hipStream_t stream = nullptr;
  std::vector<void*> owned;
  template <typename T>
  T* own(size_t count) {
    T* p = static_cast<T*>(pool_alloc((count ? count : 1) * sizeof(T)));
    if (p) owned.push_back(p);
    return p;
  }
  ~pdx_grouped() {
    StreamNote note(stream);
    pool_free_many(owned.data(), (int)owned.size());
  }
};

extern "C" {

int pdx_groupby_group_values(pdx_groupby* gb, const pdx_column* values, void* stream, pdx_grouped** out) {
  if (!gb || !out) return fail(PDX_INVALID, "pdx_groupby_group_values: null argument");
  PDX_TRY(check_column(values, "pdx_groupby_group_values"));
  if (gb->resample) return fail(PDX_INVALID, "pdx_groupby_group_values: needs a group-by handle");
  if (values->dtype != PDX_FLOAT64 || validity_or_null(values)) return fail(PDX_NOT_IMPLEMENTED, "pdx_groupby_group_values: float64 values without nulls only");
  if (values->length != gb->n) return fail(PDX_INVALID, "pdx_groupby_group_values: values length differs from the grouped key length");
  hipStream_t st = as_stream(stream);
  gb->use_on(st);  // ordered behind the handle's creation; frees of its blocks are ordered behind this stream
  std::unique_ptr<pdx_grouped> gowner(new pdx_grouped());
  gowner->stream = st;
  pdx_grouped* g = gowner.get();
  g->gb = gb;
  g->n = gb->n;
  g->G = gb->G;
  *out = nullptr;
  const int64_t n = gb->n, G = gb->G;
  g->seg_start = g->own<uint32_t>((size_t)G + 1);
  g->occ_of_gid = g->own<uint32_t>((size_t)G);
  if (!g->seg_start || !g->occ_of_gid) return PDX_OOM;
  if (n > 0 && gb->mode == 1) {
    // keys that arrived sorted: the rows are grouped as they stand (a private copy, so the handle does not borrow the caller's column)
    double* vc = g->own<double>((size_t)n);
    if (!vc) return PDX_OOM;
    PDX_HIP(hipMemcpyAsync(vc, static_cast<const double*>(values->values) + values->offset, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    PDX_HIP(hipMemcpyAsync(g->seg_start, gb->seg_start, ((size_t)G + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    g->vals_sorted = vc;
    hipLaunchKernelGGL(k_occ_of_gid, dim3(grid_for(G, 256)), dim3(256), 0, st, gb->gid_of_occ, G, g->occ_of_gid);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && !defer_sync()) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return hip_fail(e, "pdx_groupby_group_values");
  } else if (n > 0) {
    Scratch s;
    const uint32_t* ks = nullptr;
    const uint64_t* vs = nullptr;
    bool narrow_done = false;
    // (inside the sharded orchestration, which only wants the records: two passes, the last digit is left to the emitting kernel)
    static const bool emit_env = [] { const char* e = getenv("PDX_DIST_FUSED_EMIT"); return !(e && e[0] == '0'); }();
    const bool want_two = emit_env && fused_emit_wanted();
    int rc = sort_values_narrow_full(gb, static_cast<const uint64_t*>(values->values) + values->offset,
                                     [&](size_t bytes) { return (void*)g->own<uint8_t>(bytes); }, s, st, &vs, g->seg_start, &narrow_done,
                                     want_two ? &g->two : nullptr);
    if (rc != PDX_OK) return rc;
    if (narrow_done && g->two.k8) {
      g->two_pass = true;
      g->dmax = g->own<unsigned int>(3);
      if (!g->dmax) return PDX_OOM;
      PDX_HIP(hipMemsetAsync(g->dmax, 0, 3 * sizeof(unsigned int), st));
      hipLaunchKernelGGL(k_run_max_len, dim3(grid_for(g->two.nruns, 256)), dim3(256), 0, st, g->two.run_start, g->two.nruns, g->dmax, kFlrMaxRun);
    }
    if (!narrow_done) {
      rc = sort_values_by_slot(gb, static_cast<const uint64_t*>(values->values) + values->offset, nullptr, 0,
                               [&](size_t bytes) { return (void*)g->own<uint8_t>(bytes); }, s, st, &ks, &vs);
      if (rc != PDX_OK) return rc;
      hipLaunchKernelGGL(k_seg_starts, dim3(grid_for(G + 1, 256)), dim3(256), 0, st, ks, n, gb->occ_slot, G, g->seg_start);
    }
    g->vals_sorted = reinterpret_cast<const double*>(vs);
    hipLaunchKernelGGL(k_occ_of_gid, dim3(grid_for(G, 256)), dim3(256), 0, st, gb->gid_of_occ, G, g->occ_of_gid);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && !defer_sync()) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return hip_fail(e, "pdx_groupby_group_values");
  }
  *out = gowner.release();
  return PDX_OK;
}
int pdx_grouped_destroy(pdx_grouped* g) {
  delete g;
  return PDX_OK;
}
int pdx_grouped_counts(pdx_grouped* g, int64_t* out_counts, void* stream) {
  if (!g || !out_counts) return fail(PDX_INVALID, "pdx_grouped_counts: null argument");
  hipStream_t st = as_stream(stream);
  g->stream = st;  // frees of the handle's blocks are ordered behind this stream
  if (g->G) hipLaunchKernelGGL(k_grouped_counts, dim3(grid_for(g->G, 256)), dim3(256), 0, st, g->seg_start, g->gb->gid_of_occ, g->G, out_counts);
  PDX_LAUNCH_CHECK();
  if (!defer_sync()) PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}
int pdx_grouped_partial_plan(pdx_grouped* g, const int64_t* prefix, const int64_t* order, int64_t* out_total, void* stream) {
  if (!g || !prefix || !out_total) return fail(PDX_INVALID, "pdx_grouped_partial_plan: null argument");
  hipStream_t st = as_stream(stream);
  g->stream = st;  // frees of the handle's blocks are ordered behind this stream
  *out_total = 0;
  g->prefix = prefix;
  g->order = order;
  g->total = 0;
  if (g->G == 0) return PDX_OK;
  if (!g->rec_off) g->rec_off = g->own<int64_t>((size_t)g->G + 1);
  if (!g->rec_off) return PDX_OOM;
  Scratch s;
  int64_t* total = s.get<int64_t>(1);
  PDX_SCRATCH_CHECK(s);
  hipLaunchKernelGGL(k_partial_plan, dim3(grid_for(g->G, 256)), dim3(256), 0, st, g->seg_start, g->occ_of_gid, order, g->G, prefix, g->rec_off);
  PDX_TRY((device_exclusive_scan<int64_t, SumOp>(g->rec_off, g->rec_off, g->G, total, s, st)));
  if (g->two_pass) PDX_HIP(hipMemcpyAsync(g->hmax, g->dmax, sizeof(g->hmax), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipMemcpyAsync(&g->total, total, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
  *out_total = g->total;
  return PDX_OK;
}
int pdx_grouped_partial_fill(pdx_grouped* g, const int64_t* gid_map, int64_t* rec_key, double* rec_val, void* stream) {
  // (rec_key may be null: values only -- the owner of a group can regenerate every record's code from the rows (a, c) each rank holds)
  if (!g || !gid_map || !rec_val) return fail(PDX_INVALID, "pdx_grouped_partial_fill: null argument");
  if (g->total < 0 || !g->prefix) return fail(PDX_INVALID, "pdx_grouped_partial_fill: call pdx_grouped_partial_plan first");
  hipStream_t st = as_stream(stream);
  g->stream = st;  // frees of the handle's blocks are ordered behind this stream
  static_assert(kFlrBits == 6, "NarrowTwo stops in front of a last digit of <= 6 bits");
  if (g->G && g->two_pass && g->hmax[0] <= kFlrMaxRun) {
    // the last digit and the records in one kernel over the runs (workgroup per run); record offsets by LOCAL group id for it
    PDX_PROFILE("partial_fill_fused", st);
    Scratch s;
    int64_t* rec_off_lg = s.get<int64_t>((size_t)g->G);
    PDX_SCRATCH_CHECK(s);
    hipLaunchKernelGGL(k_rec_off_by_group, dim3(grid_for(g->G, 256)), dim3(256), 0, st, g->rec_off, g->order, g->G, rec_off_lg);
    const int grid = (int)std::min<int64_t>(g->two.nruns, (int64_t)kCUs * 8);
    hipLaunchKernelGGL(k_flr_emit, dim3(grid), dim3(kSortBlock), 0, st, g->two.k8, reinterpret_cast<const double*>(g->two.v1), g->two.run_start, g->two.nruns,
                       g->two.low_bits, g->gb->gid_of_slot, g->gb->nslots, g->G, g->prefix, g->seg_start, g->occ_of_gid, g->gb->occ_slot, rec_off_lg, gid_map, rec_key, rec_val);
  } else if (g->G) {
    if (g->two_pass && !g->vals_sorted) {  // a run too long for one workgroup (skewed keys): the third pass after all, then the per-group kernels
      Scratch s;
      PDX_TRY(finish_narrow_sort(g->two, g->n, s, st));
      g->vals_sorted = reinterpret_cast<const double*>(g->two.v0);
    }
    PDX_PROFILE("partial_fill", st);
    {
      const int wave_form = [] { const char* e = getenv("PDX_PARTIAL_FILL_WAVE"); return !(e && e[0] == '0'); }() ? 1 : 0;
      if (wave_form) {  // groups of <= 16 interior leaves (and the ones inside one leaf): 16 lanes each; <= 64: a wave each; longer: a thread each
        hipLaunchKernelGGL((k_partial_fill_wave<16>), dim3((unsigned)std::min<int64_t>(ceil_div(g->G, 16), (int64_t)kCUs * 32)), dim3(256), 0, st,
                           g->vals_sorted, g->seg_start, g->occ_of_gid, g->order, g->G, g->prefix, gid_map, g->rec_off, rec_key, rec_val);
        hipLaunchKernelGGL((k_partial_fill_wave<64>), dim3((unsigned)std::min<int64_t>(ceil_div(g->G, 4), (int64_t)kCUs * 32)), dim3(256), 0, st,
                           g->vals_sorted, g->seg_start, g->occ_of_gid, g->order, g->G, g->prefix, gid_map, g->rec_off, rec_key, rec_val);
      }
  hipLaunchKernelGGL(k_partial_fill, dim3(grid_for(g->G, 256)), dim3(256), 0, st, g->vals_sorted, g->seg_start, g->occ_of_gid, g->order, g->G, g->prefix,
                       gid_map, g->rec_off, rec_key, rec_val, wave_form);
    }
  }
  PDX_LAUNCH_CHECK();
  if (!defer_sync()) PDX_HIP(hipStreamSynchronize(st));
  return PDX_OK;
}
// where the records of every owner's groups begin in this rank's record stream: owner d holds the global ids [G * d / W, G * (d + 1) / W), the
// records were emitted in global-id order (order = the local groups sorted by global id), so cut d is the record offset of the first local
// group whose global id reaches the bound.  W + 1 values on the device (cuts[W] = all records).  After pdx_grouped_partial_plan.
__global__ void k_record_cuts_by_group(const int64_t* __restrict__ rec_off, const int64_t* __restrict__ order, const int64_t* __restrict__ gid_map, int64_t Gl,
                                       int64_t total, int64_t G, int W, int64_t* __restrict__ cuts) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d > W) return;
  const int64_t bound = G * d / W;
  int64_t lo = 0, hi = Gl;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (gid_map[order ? order[mid] : mid] < bound) lo = mid + 1;
    else hi = mid;
  }
  cuts[d] = lo < Gl ? rec_off[lo] : total;
}
int pdx_grouped_record_cuts(pdx_grouped* g, const int64_t* gid_map, int64_t num_global_groups, int world, int64_t* cuts, void* stream) {
  if (!g || !gid_map || !cuts || world <= 0) return fail(PDX_INVALID, "pdx_grouped_record_cuts: bad argument");
  if (g->total < 0) return fail(PDX_INVALID, "pdx_grouped_record_cuts: call pdx_grouped_partial_plan first");
  hipStream_t st = as_stream(stream);
  g->stream = st;
  hipLaunchKernelGGL(k_record_cuts_by_group, dim3((unsigned)ceil_div(world + 1, 64)), dim3(64), 0, st, g->rec_off, g->order, gid_map, g->G, g->total,
                     num_global_groups, world, cuts);
  PDX_LAUNCH_CHECK();
  return PDX_OK;
}
int pdx_replay_partials(const int64_t* rec_key, const double* rec_val, int64_t m, int64_t gid_lo, int64_t n_own, double* out_sum, void* stream) {
  if (m < 0 || n_own < 0 || (m && (!rec_key || !rec_val)) || (n_own && !out_sum)) return fail(PDX_INVALID, "pdx_replay_partials: bad argument");
  if (m > 0x7FFFFFFFll || n_own > 0x3FFFFFFFll) return fail(PDX_NOT_IMPLEMENTED, "pdx_replay_partials: too many records / groups for one call");
  hipStream_t st = as_stream(stream);
  if (n_own == 0) return PDX_OK;
  Scratch s;
  uint32_t* slot = s.get<uint32_t>((size_t)m);
  uint32_t* lvl = s.get<uint32_t>((size_t)m);
  uint32_t *k0 = s.get<uint32_t>((size_t)m), *k1 = s.get<uint32_t>((size_t)m), *k2 = s.get<uint32_t>((size_t)m), *k3 = s.get<uint32_t>((size_t)m);
  uint64_t *v0 = s.get<uint64_t>((size_t)m), *v1 = s.get<uint64_t>((size_t)m);
  uint32_t *l0 = s.get<uint32_t>((size_t)m), *l1 = s.get<uint32_t>((size_t)m);
  uint32_t* ids = s.get<uint32_t>((size_t)n_own);
  uint32_t* ss = s.get<uint32_t>((size_t)n_own + 1);
  unsigned int* bad = s.get<unsigned int>(1);
  PDX_SCRATCH_CHECK(s);
  PDX_HIP(hipMemsetAsync(bad, 0, sizeof(unsigned int), st));
  int bits = ilog2((uint64_t)n_own + 1);
  if (bits < 1) bits = 1;
  // the record's level (6 bits) rides above the slot bits of the 31-bit sort key when there is room: ONE sort instead of two
  const int pack_shift = bits <= 25 ? 25 : -1;
  if (m) hipLaunchKernelGGL(k_replay_keys, dim3(grid_for(m, 256, 4)), dim3(256), 0, st, rec_key, m, gid_lo, n_own, slot, lvl, bad, pack_shift);
  const uint32_t *ks = slot, *ks2 = nullptr, *ls = lvl;
  const uint64_t* vs = reinterpret_cast<const uint64_t*>(rec_val);
  if (m) {
    ProfileTagOverride replay_tag("replay_sort");  // (not the per-row scatter passes the bench prices against the roofline)
    PDX_TRY(radix_sort_pairs<uint64_t>(slot, reinterpret_cast<const uint64_t*>(rec_val), k0, v0, k1, v1, m, bits, &ks, &vs, true, s, st));
    if (pack_shift < 0) PDX_TRY(radix_sort_pairs<uint32_t>(slot, lvl, k2, l0, k3, l1, m, bits, &ks2, &ls, true, s, st));
  }
  if (pack_shift >= 0) {
    hipLaunchKernelGGL(k_seg_starts_masked, dim3(grid_for(n_own + 1, 256)), dim3(256), 0, st, ks, m, (1u << pack_shift) - 1u, n_own, ss);
  } else {
    hipLaunchKernelGGL(k_iota_u32, dim3(grid_for(n_own, 256)), dim3(256), 0, st, ids, n_own);
    hipLaunchKernelGGL(k_seg_starts, dim3(grid_for(n_own + 1, 256)), dim3(256), 0, st, ks, m, ids, n_own, ss);
  }
  {
    PDX_PROFILE("replay_partials", st);
    hipLaunchKernelGGL(k_replay, dim3(grid_for(n_own, 256)), dim3(256), 0, st, ss, n_own, reinterpret_cast<const double*>(vs), pack_shift >= 0 ? ks : ls, out_sum,
                       bad, pack_shift);
  }
  PDX_LAUNCH_CHECK();
  unsigned int hbad = 0;
  PDX_HIP(hipMemcpyAsync(&hbad, bad, sizeof(hbad), hipMemcpyDeviceToHost, st));
  PDX_HIP(hipStreamSynchronize(st));
  if (hbad == 1) return fail(PDX_INVALID, "pdx_replay_partials: record for a group outside [gid_lo, gid_lo + n_own)");
  if (hbad == 2) return fail(PDX_INVALID, "pdx_replay_partials: node record inside an unfinished leaf (records out of order)");
  if (hbad == 3) return fail(PDX_INVALID, "pdx_replay_partials: an owned group received no record");
  return PDX_OK;
}

}  // extern "C"
