// prims.hip — see prims.hpp
#include "prims.hpp"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

namespace hope {

template <class K>
static hipError_t sort_pairs_t(const PrimAlloc& a, const K* kin, K* kout, const uint32_t* vin, uint32_t* vout,
                               size_t n, int end_bit, hipStream_t s)
{
  if (n == 0) return hipSuccess;
  size_t bytes = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0u, (unsigned)end_bit, s);
  if (e != hipSuccess) return e;
  void* tmp = a(bytes ? bytes : 16);
  if (!tmp) return hipErrorOutOfMemory;
  return rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, n, 0u, (unsigned)end_bit, s);
}

hipError_t sort_pairs_u32(const PrimAlloc& a, const uint32_t* kin, uint32_t* kout, const uint32_t* vin,
                          uint32_t* vout, size_t n, int end_bit, hipStream_t s)
{
  return sort_pairs_t<uint32_t>(a, kin, kout, vin, vout, n, end_bit, s);
}

hipError_t segmented_sort_pairs_u32(const PrimAlloc& a, const uint32_t* kin, uint32_t* kout, const uint32_t* vin,
                                    uint32_t* vout, size_t n, unsigned n_segments, const uint32_t* seg_off,
                                    int end_bit, hipStream_t s)
{
  if (n == 0) return hipSuccess;
  size_t bytes = 0;
  hipError_t e = rocprim::segmented_radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, (unsigned)n, n_segments,
                                                     seg_off, seg_off + 1, 0u, (unsigned)end_bit, s);
  if (e != hipSuccess) return e;
  void* tmp = a(bytes ? bytes : 16);
  if (!tmp) return hipErrorOutOfMemory;
  return rocprim::segmented_radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, (unsigned)n, n_segments, seg_off,
                                             seg_off + 1, 0u, (unsigned)end_bit, s);
}

hipError_t sort_pairs_u64(const PrimAlloc& a, const uint64_t* kin, uint64_t* kout, const uint32_t* vin,
                          uint32_t* vout, size_t n, int end_bit, hipStream_t s)
{
  return sort_pairs_t<uint64_t>(a, kin, kout, vin, vout, n, end_bit, s);
}

template <class K>
static hipError_t rle_t(const PrimAlloc& a, const K* keys, size_t n, K* unique, uint32_t* counts, uint32_t* n_runs,
                        hipStream_t s)
{
  if (n == 0) return hipMemsetAsync(n_runs, 0, sizeof(uint32_t), s);
  size_t bytes = 0;
  hipError_t e = rocprim::run_length_encode(nullptr, bytes, keys, (unsigned)n, unique, counts, n_runs, s);
  if (e != hipSuccess) return e;
  void* tmp = a(bytes ? bytes : 16);
  if (!tmp) return hipErrorOutOfMemory;
  return rocprim::run_length_encode(tmp, bytes, keys, (unsigned)n, unique, counts, n_runs, s);
}

hipError_t rle_u32(const PrimAlloc& a, const uint32_t* keys, size_t n, uint32_t* unique, uint32_t* counts,
                   uint32_t* n_runs, hipStream_t s)
{
  return rle_t<uint32_t>(a, keys, n, unique, counts, n_runs, s);
}

hipError_t rle_u64(const PrimAlloc& a, const uint64_t* keys, size_t n, uint64_t* unique, uint32_t* counts,
                   uint32_t* n_runs, hipStream_t s)
{
  return rle_t<uint64_t>(a, keys, n, unique, counts, n_runs, s);
}

hipError_t exclusive_scan_u32(const PrimAlloc& a, const uint32_t* in, uint32_t* out, size_t n, hipStream_t s)
{
  if (n == 0) return hipSuccess;
  size_t bytes = 0;
  hipError_t e = rocprim::exclusive_scan(nullptr, bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), s);
  if (e != hipSuccess) return e;
  void* tmp = a(bytes ? bytes : 16);
  if (!tmp) return hipErrorOutOfMemory;
  return rocprim::exclusive_scan(tmp, bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), s);
}

struct PlusU3 {
  __host__ __device__ U3 operator()(const U3& x, const U3& y) const { return U3{x.a + y.a, x.b + y.b, x.c + y.c}; }
};

hipError_t exclusive_scan_u32x3(const PrimAlloc& a, const U3* in, U3* out, size_t n, hipStream_t s)
{
  if (n == 0) return hipSuccess;
  size_t bytes = 0;
  hipError_t e = rocprim::exclusive_scan(nullptr, bytes, in, out, U3{0u, 0u, 0u}, n, PlusU3(), s);
  if (e != hipSuccess) return e;
  void* tmp = a(bytes ? bytes : 16);
  if (!tmp) return hipErrorOutOfMemory;
  return rocprim::exclusive_scan(tmp, bytes, in, out, U3{0u, 0u, 0u}, n, PlusU3(), s);
}

}  // namespace hope
