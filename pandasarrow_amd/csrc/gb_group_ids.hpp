// gb_group_ids.hpp -- part of groupby.hip (textually included there, in its namespace context; split by stage, kernels unchanged):
// occupied slots -> dense group ids in first-occurrence order, per-row ids, group offsets of the sorted layout.
#pragma once

// first-row of every slot: from the hash table (table != nullptr) or the dense first[] array
struct OccPred {
  const Slot* table;
  const unsigned int* first;
  unsigned int region, cap;  // region != 0: i is a LOGICAL slot of the partitioned table
  __device__ bool operator()(int64_t i) const { return (table ? table[phys_slot(i, region, cap)].first : first[i]) != kNoRow; }
};
struct OccEmit {
  const Slot* table;
  const unsigned int* first;
  unsigned int region, cap;
  uint32_t* occ_slot;
  uint32_t* occ_first;
  __device__ void operator()(int64_t pos, int64_t i) const {
    occ_slot[pos] = (uint32_t)i;
    occ_first[pos] = table ? table[phys_slot(i, region, cap)].first : first[i];
  }
};

// sorted_slot[r] = slot of the r-th group in first-occurrence order
// null_slot: the slot of the null key; table == nullptr => dense mode (key = dense_min + slot)
__global__ void k_assign_gids(const Slot* __restrict__ table, long long dense_min, unsigned int dense_mask, uint32_t* __restrict__ gid_of_slot,
                              const uint32_t* __restrict__ sorted_first, const uint32_t* __restrict__ sorted_slot, int64_t G,
                              unsigned int null_slot, int64_t* __restrict__ uniques, uint8_t* __restrict__ unique_ok,
                              int64_t* __restrict__ first_rows, unsigned int region) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < G; r += stride) {
    unsigned int s = sorted_slot[r];
    gid_of_slot[s] = (unsigned int)r;
    long long k;
    if (table) {
      k = table[phys_slot(s, region, null_slot)].key;
      if (s == null_slot + 1) k = kEmptyKey;
    } else {
      k = dense_mask ? (long long)((unsigned long long)dense_min + (((unsigned long long)s - (unsigned long long)dense_min) & dense_mask))
                     : (long long)((unsigned long long)dense_min + (unsigned long long)s);
    }
    if (s == null_slot) k = 0;
    uniques[r] = k;
    unique_ok[r] = s != null_slot;
    first_rows[r] = (int64_t)sorted_first[r];
  }
}
// ---- first-occurrence rank without a sort: every group's first row sets one bit of an n-bit map; a group's id is the number of
// set bits in front of its own (the first rows are distinct).  n / 8 bytes of traffic and a handful of launches, where sorting the
// (first row, slot) pairs took four LSD passes = twenty dependent small kernels (~0.45 ms of a 16 ms step at 1e6 groups, and a
// full-size sort when almost every row is its own group).
constexpr int kRankWords = 4;  // 64-bit words per counted block (256 rows): a group reads <= 4 words + one prefix for its rank
__global__ void k_mark_first_rows(const uint32_t* __restrict__ occ_first, int64_t G, unsigned long long* __restrict__ bits) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < G; j += stride) {
    const uint32_t fr = occ_first[j];
    atomicOr(&bits[fr >> 6], 1ull << (fr & 63));
  }
}
__global__ void k_rank_block_counts(const unsigned long long* __restrict__ bits, int64_t nwords, int64_t nblocks, int64_t* __restrict__ counts) {
  // kRankWords lanes per block, one word each (coalesced), folded with shuffles
  static_assert(kRankWords == 4, "sixteen blocks per wave");
  const int sub = threadIdx.x & (kRankWords - 1);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x / kRankWords;
  const int64_t nb_round = (nblocks + 15) & ~(int64_t)15;  // whole waves stay in the loop (the shuffles need all lanes)
  for (int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kRankWords; b < nb_round; b += stride) {
    const int64_t w = b * kRankWords + sub;
    int c = (b < nblocks && w < nwords) ? __popcll(bits[w]) : 0;
    c += __shfl_xor(c, 2, 64);
    c += __shfl_xor(c, 1, 64);
    if (sub == 0 && b < nblocks) counts[b] = c;
  }
}
// k_assign_gids with the rank computed from the bit map (block_pre = exclusive prefix of k_rank_block_counts)
__global__ void k_assign_gids_ranked(const Slot* __restrict__ table, long long dense_min, unsigned int dense_mask, uint32_t* __restrict__ gid_of_slot,
                                     const uint32_t* __restrict__ occ_first, const uint32_t* __restrict__ occ_slot, int64_t G,
                                     const unsigned long long* __restrict__ bits, const int64_t* __restrict__ block_pre, unsigned int null_slot,
                                     int64_t* __restrict__ uniques, uint8_t* __restrict__ unique_ok, int64_t* __restrict__ first_rows,
                                     unsigned int region, uint32_t* __restrict__ gid_of_occ) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < G; j += stride) {
    const unsigned int s = occ_slot[j];
    const uint32_t fr = occ_first[j];
    const int64_t w = fr >> 6, b = w / kRankWords;
    int64_t r = block_pre[b];
    for (int64_t q = b * kRankWords; q < w; ++q) r += __popcll(bits[q]);
    r += __popcll(bits[w] & ((1ull << (fr & 63)) - 1ull));
    gid_of_slot[s] = (unsigned int)r;
    gid_of_occ[j] = (unsigned int)r;
    long long k;
    if (table) {
      k = table[phys_slot(s, region, null_slot)].key;
      if (s == null_slot + 1) k = kEmptyKey;
    } else {
      k = dense_mask ? (long long)((unsigned long long)dense_min + (((unsigned long long)s - (unsigned long long)dense_min) & dense_mask))
                     : (long long)((unsigned long long)dense_min + (unsigned long long)s);
    }
    if (s == null_slot) k = 0;
    uniques[r] = k;
    unique_ok[r] = s != null_slot;
    first_rows[r] = (int64_t)fr;
  }
}
__global__ void k_gid_of_occ(const uint32_t* __restrict__ gid_of_slot, const uint32_t* __restrict__ occ_slot, int64_t G, uint32_t* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < G; k += stride) out[k] = gid_of_slot[occ_slot[k]];
}
__global__ void k_row_gids(const uint32_t* __restrict__ gid_of_slot, const uint32_t* __restrict__ slot_of_row, int64_t n, uint32_t* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = gid_of_slot[slot_of_row[i]];
}

__global__ void k_map_ids(const uint32_t* __restrict__ gid_of_slot, const uint32_t* __restrict__ slot_of_row, const uint32_t* __restrict__ seg_start,
                          int64_t G, int64_t n, const int64_t* __restrict__ map, int64_t* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint32_t g;
    if (gid_of_slot) g = gid_of_slot[slot_of_row[i]];
    else {
      int64_t lo = 0, hi = G;
      while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (seg_start[mid] <= (uint32_t)i) lo = mid;
        else hi = mid;
      }
      g = (uint32_t)lo;
    }
    out[i] = map[g];
  }
}

// keys for the value sort when the value column has nulls: bit 31 = row is null
__global__ void k_flag_keys(const uint32_t* __restrict__ slot_of_row, const uint8_t* __restrict__ valid, int64_t off, int64_t n,
                            uint32_t* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = slot_of_row[i] | (bit_get(valid, off + i) ? 0u : 0x80000000u);
}

// seg_start[k] = first position in sorted keys whose (masked) key >= occ_slot[k]; seg_start[G] = n
__global__ void k_seg_starts(const uint32_t* __restrict__ sorted_keys, int64_t n, const uint32_t* __restrict__ occ_slot, int64_t G,
                             uint32_t* __restrict__ seg_start) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= G; k += stride) {
    if (k == G) {
      seg_start[k] = (uint32_t)n;
      continue;
    }
    uint32_t target = occ_slot[k];
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      if ((sorted_keys[mid] & kSortKeyMask) < target) lo = mid + 1;
      else hi = mid;
    }
    seg_start[k] = (uint32_t)lo;
  }
}

// seg_start[k] = slot_start[occ_slot[k]] (starts of every slot's rows, from k_level_starts); seg_start[G] = n
__global__ void k_seg_starts_from_slots(const uint32_t* __restrict__ slot_start, int64_t n, const uint32_t* __restrict__ occ_slot, int64_t G,
                                        uint32_t* __restrict__ seg_start) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= G; k += stride) seg_start[k] = k == G ? (uint32_t)n : slot_start[occ_slot[k]];
}

// k_seg_starts on keys that carry extra bits above `mask`
__global__ void k_seg_starts_masked(const uint32_t* __restrict__ sorted_keys, int64_t n, uint32_t mask, int64_t G, uint32_t* __restrict__ seg_start) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= G; k += stride) {
    int64_t lo = 0, hi = n;
    if (k == G) lo = n;
    else
      while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((sorted_keys[mid] & mask) < (uint32_t)k) lo = mid + 1;
        else hi = mid;
      }
    seg_start[k] = (uint32_t)lo;
  }
}
