// prims.hpp — device-wide primitives (stable radix sort of pairs, run-length encode,
// exclusive scan).  Round-1 implementation: thin wrappers over rocPRIM, compiled in
// their own translation unit; every call is asynchronous on `stream` and takes its
// temporary storage from the caller's arena through `alloc`.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace hope {

typedef void* (*prim_alloc_fn)(void* user, size_t bytes);

struct PrimAlloc {
  prim_alloc_fn fn;
  void* user;
  void* operator()(size_t bytes) const { return fn(user, bytes); }
};

// stable LSD radix sort of (key, value) pairs on key bits [0, end_bit)
hipError_t sort_pairs_u32(const PrimAlloc& a, const uint32_t* kin, uint32_t* kout, const uint32_t* vin,
                          uint32_t* vout, size_t n, int end_bit, hipStream_t s);
hipError_t sort_pairs_u64(const PrimAlloc& a, const uint64_t* kin, uint64_t* kout, const uint32_t* vin,
                          uint32_t* vout, size_t n, int end_bit, hipStream_t s);
// the same within each segment [seg_off[i], seg_off[i + 1]) (segments stay where they are)
hipError_t segmented_sort_pairs_u32(const PrimAlloc& a, const uint32_t* kin, uint32_t* kout, const uint32_t* vin,
                                    uint32_t* vout, size_t n, unsigned n_segments, const uint32_t* seg_off,
                                    int end_bit, hipStream_t s);
// run-length encode: unique[r], counts[r], *n_runs (device)
hipError_t rle_u32(const PrimAlloc& a, const uint32_t* keys, size_t n, uint32_t* unique, uint32_t* counts,
                   uint32_t* n_runs, hipStream_t s);
hipError_t rle_u64(const PrimAlloc& a, const uint64_t* keys, size_t n, uint64_t* unique, uint32_t* counts,
                   uint32_t* n_runs, hipStream_t s);
// out[i] = sum_{j<i} in[j]
hipError_t exclusive_scan_u32(const PrimAlloc& a, const uint32_t* in, uint32_t* out, size_t n, hipStream_t s);
// three independent scans in one pass (component-wise sums)
struct U3 {
  uint32_t a, b, c;
};
hipError_t exclusive_scan_u32x3(const PrimAlloc& a, const U3* in, U3* out, size_t n, hipStream_t s);

}  // namespace hope
