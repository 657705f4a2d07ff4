// gb_handle.hpp -- part of groupby.hip: the opaque handle behind pdx_groupby (hash group-by or resample segments).
#pragma once

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
  uint32_t* slot_part = nullptr;    // n, logical slot per partitioned position
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
  // segments mode
  uint32_t* seg_start = nullptr;   // G + 1
  BinParams bin{};
  long long label_base = 0;
  mutable hipStream_t stream = nullptr;  // the stream of the last call that used the handle (pool frees are ordered behind it)
  std::vector<void*> owned;
  template <typename T>
  T* own(size_t count) {
    T* p = static_cast<T*>(pool_alloc((count ? count : 1) * sizeof(T)));
    if (p) owned.push_back(p);
    return p;
  }
  ~pdx_groupby() {
    StreamNote note(stream);
    pool_free_many(owned.data(), (int)owned.size());
  }
};
