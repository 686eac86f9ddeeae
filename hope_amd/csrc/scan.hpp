// scan.hpp -- hand-written device-wide exclusive scan of 32-bit counters (reduce, scan of the block sums, apply):
// three launches per level, no spinning on other workgroups.  out[i] = sum_{j<i} in[j]; in == out is allowed.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace hope {

constexpr int kScanThreads = 256;
constexpr int kScanIPT = 16;
constexpr int kScanTile = kScanThreads * kScanIPT;  // 4096 counters per workgroup

// inclusive scan over the lanes of a wave
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v, int lane)
{
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = (unsigned)__shfl_up((int)v, o);
    if (lane >= o) v += t;
  }
  return v;
}

// exclusive scan over the threads of a workgroup of kScanThreads; returns the prefix of `v`, *total = the sum.
// s_w: kScanThreads / 64 + 1 words of LDS.  Ends with a barrier-protected read: callers may reuse s_w after
// their next barrier.
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned* s_w, unsigned* total)
{
  constexpr int kW = kScanThreads / 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned inc = wave_incl_scan(v, lane);
  __syncthreads();  // (s_w may still be read from a previous call)
  if (lane == 63) s_w[w] = inc;
  __syncthreads();
  unsigned pre = 0u, tot = 0u;
#pragma unroll
  for (int i = 0; i < kW; ++i) {
    const unsigned t = s_w[i];
    if (i < w) pre += t;
    tot += t;
  }
  *total = tot;
  return pre + inc - v;
}

static __global__ void __launch_bounds__(kScanThreads) k_scan_reduce(const uint32_t* in, size_t n, uint32_t* sums)
{
  __shared__ unsigned s_w[kScanThreads / 64 + 1];
  const size_t base = (size_t)blockIdx.x * kScanTile;
  unsigned v = 0u;
#pragma unroll
  for (int k = 0; k < kScanIPT; ++k) {
    const size_t i = base + (size_t)k * kScanThreads + threadIdx.x;
    if (i < n) v += in[i];
  }
  unsigned tot;
  block_excl_scan(v, s_w, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// one workgroup: exclusive scan of up to any number of sums, kScanTile at a time (in place)
static __global__ void __launch_bounds__(kScanThreads) k_scan_sums(uint32_t* sums, size_t n)
{
  __shared__ unsigned s_w[kScanThreads / 64 + 1];
  unsigned carry = 0u;
  for (size_t base = 0; base < n; base += kScanTile) {
    unsigned item[kScanIPT];
    unsigned v = 0u;
    const size_t first = base + (size_t)threadIdx.x * kScanIPT;  // consecutive counters per thread
#pragma unroll
    for (int k = 0; k < kScanIPT; ++k) {
      item[k] = first + k < n ? sums[first + k] : 0u;
      v += item[k];
    }
    unsigned tot;
    unsigned pre = carry + block_excl_scan(v, s_w, &tot);
#pragma unroll
    for (int k = 0; k < kScanIPT; ++k) {
      if (first + k < n) sums[first + k] = pre;
      pre += item[k];
    }
    carry += tot;
  }
}

static __global__ void __launch_bounds__(kScanThreads)
k_scan_apply(const uint32_t* in, size_t n, const uint32_t* sums, uint32_t* out)
{
  __shared__ unsigned s_w[kScanThreads / 64 + 1];
  const size_t base = (size_t)blockIdx.x * kScanTile;
  const size_t first = base + (size_t)threadIdx.x * kScanIPT;
  unsigned item[kScanIPT];
  unsigned v = 0u;
  if (first + kScanIPT <= n) {
    const uint4* p = reinterpret_cast<const uint4*>(in + first);  // (first is a multiple of 16: aligned)
#pragma unroll
    for (int k = 0; k < kScanIPT / 4; ++k) {
      const uint4 q = p[k];
      item[4 * k] = q.x;
      item[4 * k + 1] = q.y;
      item[4 * k + 2] = q.z;
      item[4 * k + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < kScanIPT; ++k) item[k] = first + k < n ? in[first + k] : 0u;
  }
#pragma unroll
  for (int k = 0; k < kScanIPT; ++k) v += item[k];
  unsigned tot;
  unsigned pre = sums[blockIdx.x] + block_excl_scan(v, s_w, &tot);
  if (first + kScanIPT <= n) {
    uint4* p = reinterpret_cast<uint4*>(out + first);
#pragma unroll
    for (int k = 0; k < kScanIPT / 4; ++k) {
      uint4 q;
      q.x = pre; pre += item[4 * k];
      q.y = pre; pre += item[4 * k + 1];
      q.z = pre; pre += item[4 * k + 2];
      q.w = pre; pre += item[4 * k + 3];
      p[k] = q;
    }
  } else {
#pragma unroll
    for (int k = 0; k < kScanIPT; ++k) {
      if (first + k < n) out[first + k] = pre;
      pre += item[k];
    }
  }
}

// sums: scratch of scan_sums_words(n) words.  in and out must be 16-byte aligned (arena allocations are).
inline size_t scan_sums_words(size_t n) { return (n + kScanTile - 1) / kScanTile + 1; }

inline void scan_exclusive_u32(const uint32_t* in, uint32_t* out, size_t n, uint32_t* sums, hipStream_t s)
{
  if (n == 0) return;
  const unsigned nb = (unsigned)((n + kScanTile - 1) / kScanTile);
  hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(kScanThreads), 0, s, in, n, sums);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kScanThreads), 0, s, sums, (size_t)nb);
  hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(kScanThreads), 0, s, in, n, (const uint32_t*)sums, out);
}

}  // namespace hope
