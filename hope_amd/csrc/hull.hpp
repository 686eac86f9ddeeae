// hull.hpp — F2: the device side of the convex hull of a plane patch.
//
// Utilities::computeHull (utilities.cpp:1531-1536, called plane_segment.cpp:957) and PlaneSegment::computeHull
// (plane_segment.cpp:391-406) hand EVERY point of a plane to pcl::ConvexHull.  Almost none of them can be a
// vertex: a point strictly inside the octagon of the patch's eight extreme points (smallest and largest x, y,
// x + y and x - y) is strictly inside the hull (Akl & Toussaint).  k_hull_candidates keeps the others, in member
// order; the hull of what it keeps IS the hull of the patch, vertex for vertex, and only those few points cross
// to the host, where the monotone chain runs (hope_amd/host/hull2d.hpp).
#pragma once
#include "voxel_rank.hpp"  // block_scan_excl_1024

namespace hope {

constexpr int kHullBlock = kRankBlock;
constexpr int kHullDirs = 8;

// grid = 1, block = kHullBlock.  members[*first_p ... + size): the cluster's points as indices into fit_pt.
// out_pos: the positions (0 .. size-1, ascending) of the members kept; out_count[0]: how many.
__global__ void __launch_bounds__(kHullBlock)
k_hull_candidates(const uint32_t* members, const uint32_t* first_p, unsigned size, const float4* fit_pt,
                  uint32_t* out_pos, uint32_t* out_count)
{
  __shared__ float s_val[kHullDirs][kHullBlock / kWave];
  __shared__ unsigned s_at[kHullDirs][kHullBlock / kWave];
  __shared__ double s_oct[kHullDirs][2];
  __shared__ unsigned s_w[kHullBlock / kWave + 1];
  const unsigned tid = threadIdx.x;
  const int lane = (int)(tid & (kWave - 1)), wv = (int)(tid / kWave);
  const uint32_t* mem = members + *first_p;
  // ---- the extreme member of each of eight directions, anticlockwise from west; ties: first in the list ----
  float best[kHullDirs];
  unsigned at[kHullDirs];
#pragma unroll
  for (int d = 0; d < kHullDirs; ++d) {
    best[d] = -INFINITY;
    at[d] = 0xffffffffu;
  }
  for (unsigned i = tid; i < size; i += kHullBlock) {
    const float4 p = fit_pt[mem[i]];
    if (!(isfinite(p.x) && isfinite(p.y))) continue;
    const float key[kHullDirs] = {-p.x, -(p.x + p.y), -p.y, p.x - p.y, p.x, p.x + p.y, p.y, p.y - p.x};
#pragma unroll
    for (int d = 0; d < kHullDirs; ++d)
      if (key[d] > best[d]) {  // (a thread meets its members in ascending position)
        best[d] = key[d];
        at[d] = i;
      }
  }
#pragma unroll
  for (int d = 0; d < kHullDirs; ++d) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
      const float v = __shfl_down(best[d], o);
      const unsigned a = (unsigned)__shfl_down((int)at[d], o);
      if (v > best[d] || (v == best[d] && a < at[d])) {
        best[d] = v;
        at[d] = a;
      }
    }
    if (lane == 0) {
      s_val[d][wv] = best[d];
      s_at[d][wv] = at[d];
    }
  }
  __syncthreads();
  if (tid < (unsigned)kHullDirs) {
    float v = -INFINITY;
    unsigned a = 0xffffffffu;
    for (int w = 0; w < kHullBlock / kWave; ++w)
      if (s_val[tid][w] > v || (s_val[tid][w] == v && s_at[tid][w] < a)) {
        v = s_val[tid][w];
        a = s_at[tid][w];
      }
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a != 0xffffffffu) p = fit_pt[mem[a]];
    s_oct[tid][0] = (double)p.x;
    s_oct[tid][1] = (double)p.y;
  }
  __syncthreads();
  double ox[kHullDirs], oy[kHullDirs];
#pragma unroll
  for (int d = 0; d < kHullDirs; ++d) {
    ox[d] = s_oct[d][0];
    oy[d] = s_oct[d][1];
  }
  // ---- keep what is not strictly inside the octagon (with a margin far above the rounding of the test) ----
  unsigned base = 0u;
  for (unsigned i0 = 0u; i0 < size; i0 += kHullBlock) {  // (uniform)
    const unsigned i = i0 + tid;
    unsigned keep = 0u;
    if (i < size) {
      const float4 p = fit_pt[mem[i]];
      const double px = (double)p.x, py = (double)p.y;
      bool inside = isfinite(p.x) && isfinite(p.y);
#pragma unroll
      for (int d = 0; d < kHullDirs; ++d) {
        const int e = (d + 1) % kHullDirs;
        const double ex = ox[e] - ox[d], ey = oy[e] - oy[d];
        if (ex == 0.0 && ey == 0.0) continue;  // (two directions share their extreme point)
        const double a = ex * (py - oy[d]), b = ey * (px - ox[d]);
        if (!(a - b > 1e-9 * (fabs(a) + fabs(b)))) inside = false;
      }
      keep = inside ? 0u : 1u;
    }
    unsigned total;
    const unsigned ex = block_scan_excl_1024(keep, s_w, total);
    if (keep) out_pos[base + ex] = i;
    base += total;
  }
  if (tid == 0) out_count[0] = base;
}

}  // namespace hope
