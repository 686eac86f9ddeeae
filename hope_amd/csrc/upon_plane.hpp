// upon_plane.hpp — SURVEY.md section 8f row F1: the points above the winning plane, inside its contour,
// clustered by Euclidean distance (Utilities::getClustersUponPlane, utilities.cpp:1363-1408).
// Integer results (membership, component labels) do not depend on any evaluation order; the only float
// decisions are the per-point filter and `dist2 < r2`, both written in the reference's operation order.
//
// Clustering = connected components of the symmetric graph "dist2 < r2".  Points are binned in cells a
// little wider than the tolerance, sorted by cell key, and every point tests the points of its 27 neighbour
// cells (9 binary searches for 3-cell x runs), uniting through a lock-free union-find whose roots are the
// smallest point index of the component -- which is also the order PCL discovers the clusters in.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/hope_detmath.h"
#include "kernels.hpp"

namespace hope {

struct UponFilter {
  double z_thr;          // (double)z_max + 0.01
  float cx, cy;          // getStraightRect2D centre
  double half_w, half_h; // (double)width * 0.5, (double)height * 0.5
  int nc;                // contour vertices
};

struct UponGrid {
  float mn[3];
  float inv_cell;        // 1 / cell size
  unsigned nx, ny, nz;
};

// Utilities::isInContour (utilities.cpp:1337-1360)
__device__ __forceinline__ bool is_in_contour(const float2* contour, int nc, float px, float py)
{
  float angle_sum = 0.f;
  float2 vi = nc > 0 ? contour[0] : make_float2(0.f, 0.f);
  for (int i = 0; i < nc; ++i) {
    const float2 vj = contour[i + 1 < nc ? i + 1 : 0];
    const float ax = vi.x - px, ay = vi.y - py, bx = vj.x - px, by = vj.y - py;
    const float cz = ax * by - ay * bx;
    const float nrm = sqrtf(cz * cz);
    const float dot = ax * bx + ay * by;
    angle_sum += fabsf(hope_det_atan2f(nrm, dot));
    vi = vj;
  }
  return fabs((double)angle_sum - 2 * 3.14159265358979323846) < 0.01;
}

template <class Src>
__global__ void __launch_bounds__(kBlock)
k_upon_flags(Src src, const FrameInfo* fi, int frame, const float2* contour, UponFilter f, uint32_t* flags)
{
  const unsigned n = src_count(src);
  const unsigned i = blockIdx.x * kBlock + threadIdx.x;
  if (i > n) return;
  uint32_t keep = 0u;
  float x, y, z;
  if (i < n && load_point(src, fi, frame, i, x, y, z)) {
    if (!((double)z < f.z_thr) && !((double)fabsf(x - f.cx) > f.half_w || (double)fabsf(y - f.cy) > f.half_h))
      keep = is_in_contour(contour, f.nc, x, y) ? 1u : 0u;
  }
  flags[i] = keep;
}

// compact the kept points (ascending source order = the order of `temp`) and reduce their bounding box
template <class Src>
__global__ void __launch_bounds__(kBlock)
k_upon_compact(Src src, const FrameInfo* fi, int frame, const uint32_t* flags, const uint32_t* scan, float4* pt, int* src_index,
               unsigned* bbox_ord)
{
  const unsigned n = src_count(src);
  const unsigned i = blockIdx.x * kBlock + threadIdx.x;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  if (i < n && flags[i]) {
    float x, y, z;
    load_point(src, fi, frame, i, x, y, z);
    const unsigned o = scan[i];
    pt[o] = make_float4(x, y, z, 0.f);
    src_index[o] = (int)i;
    mn[0] = mx[0] = x; mn[1] = mx[1] = y; mn[2] = mx[2] = z;
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      mn[d] = fminf(mn[d], __shfl_xor(mn[d], off));
      mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], off));
    }
  if ((threadIdx.x & (kWave - 1)) == 0 && mn[0] <= mx[0]) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      atomicMin(&bbox_ord[d], f2ord(mn[d]));
      atomicMax(&bbox_ord[3 + d], f2ord(mx[d]));
    }
  }
}

__device__ __forceinline__ void upon_cell(const UponGrid& g, const float4& p, unsigned& ix, unsigned& iy, unsigned& iz)
{
  ix = min((unsigned)fmaxf((p.x - g.mn[0]) * g.inv_cell, 0.f), g.nx - 1);
  iy = min((unsigned)fmaxf((p.y - g.mn[1]) * g.inv_cell, 0.f), g.ny - 1);
  iz = min((unsigned)fmaxf((p.z - g.mn[2]) * g.inv_cell, 0.f), g.nz - 1);
}

__global__ void __launch_bounds__(kBlock)
k_upon_keys(const float4* pt, unsigned n, UponGrid g, uint64_t* keys, uint32_t* vals, int* parent)
{
  const unsigned i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  unsigned ix, iy, iz;
  upon_cell(g, pt[i], ix, iy, iz);
  keys[i] = ((uint64_t)iz * g.ny + iy) * g.nx + ix;
  vals[i] = i;
  parent[i] = (int)i;
}

__global__ void __launch_bounds__(kBlock)
k_upon_gather(const float4* pt, const uint32_t* vals, unsigned n, float4* sorted_pt)
{
  const unsigned s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n) return;
  float4 p = pt[vals[s]];
  p.w = __uint_as_float(vals[s]);
  sorted_pt[s] = p;
}

__global__ void __launch_bounds__(kBlock)
k_upon_union(const float4* sorted_pt, const uint64_t* keys, unsigned n, UponGrid g, float r2, int* parent)
{
  const unsigned s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n) return;
  const float4 p = sorted_pt[s];
  const int i = (int)__float_as_uint(p.w);
  unsigned ix, iy, iz;
  upon_cell(g, p, ix, iy, iz);
  const unsigned x0 = ix ? ix - 1 : 0, x1 = min(ix + 1, g.nx - 1);
  for (int dz = -1; dz <= 1; ++dz) {
    const long long z = (long long)iz + dz;
    if (z < 0 || z >= (long long)g.nz) continue;
    for (int dy = -1; dy <= 1; ++dy) {
      const long long y = (long long)iy + dy;
      if (y < 0 || y >= (long long)g.ny) continue;
      const uint64_t row = ((uint64_t)z * g.ny + (uint64_t)y) * g.nx;
      const uint64_t k0 = row + x0, k1 = row + x1;
      unsigned lo = 0, hi = n;
      while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        if (keys[mid] < k0) lo = mid + 1; else hi = mid;
      }
      for (unsigned t = lo; t < n && keys[t] <= k1; ++t) {
        const float4 q = sorted_pt[t];
        const int j = (int)__float_as_uint(q.w);
        if (j <= i) continue;                                // every pair once
        const float dx = p.x - q.x, dy2 = p.y - q.y, dz2 = p.z - q.z;
        if (!((dx * dx + dy2 * dy2) + dz2 * dz2 < r2)) continue;   // FLANN L2_Simple, strict radius
        int a = i, b = j;
        while (true) {
          a = uf_find(parent, a);
          b = uf_find(parent, b);
          if (a == b) break;
          if (a > b) { const int tmp = a; a = b; b = tmp; }
          if (atomicCAS(&parent[b], b, a) == b) break;
        }
      }
    }
  }
}

__global__ void __launch_bounds__(kBlock) k_upon_labels(int* parent, unsigned n, uint32_t* label, uint32_t* iota)
{
  const unsigned i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  label[i] = (uint32_t)uf_find(parent, (int)i);
  iota[i] = i;
}

__global__ void __launch_bounds__(kBlock)
k_upon_output(const float4* pt, const int* src_index, const uint32_t* order, unsigned n, float* out_xyz, int* out_src)
{
  const unsigned s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n) return;
  const uint32_t i = order[s];
  const float4 p = pt[i];
  out_xyz[3 * (size_t)s] = p.x;
  out_xyz[3 * (size_t)s + 1] = p.y;
  out_xyz[3 * (size_t)s + 2] = p.z;
  out_src[s] = src_index[i];
}

}  // namespace hope
