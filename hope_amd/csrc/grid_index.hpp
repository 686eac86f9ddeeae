// grid_index.hpp -- A6 (pcl::NormalEstimation with a radius search, utilities.cpp:428-446) for ONE cloud over a
// 3-D occupancy index instead of the column index.
//
// VoxelGrid's output holds at most one point per cell, in ascending cell order.  So a bitmap of the occupied cells
// with a running count per 32-bit word IS an index: the point of cell c is vox[rank(c)], rank(c) = the word's
// prefix + the set bits below c.  (27 MB for the 1.08e8 cells of the 1e8-point scene: 0.1 ms to build, against
// 0.57 ms for keys + radix sort + table + gather of the column index, which A6 then does not need at all.)
// A radius query looks at the few (y, z) rows of cells its sphere can reach; the x-neighbours of a row are adjacent
// bits, so ONE 16-byte read of (bits, prefix) x 2 words yields the row's occupied cells AND, their ranks being
// consecutive, the contiguous run of vox[] that holds their points.  A clutter point -- 92 % of the down-sampled
// 1e8-point scene: 30 points per xy column, a neighbour or two within the radius -- costs ~21 independent table
// reads and a handful of point reads, where the column index paid a dependent chain of a binary search on z per
// column (9 columns x ~7 round trips).
//
// The cell of a point is derived from the point itself (the down-sampled centroid), with the same monotone
// function the column index uses (cell_unclamped + clamp), so the reach arguments of neighbours.hpp carry over.
// A centroid can round across its VoxelGrid cell's face; then two points may share a derived cell or the derived
// cells may not ascend with the index.  k_grid_mark checks exactly that (derived cells strictly ascending along
// the array) and raises Globals::grid_unsorted; the host then runs the call over the column index instead.  The same
// check lets an adopted down-sampled cloud (hope_segment_downsampled*) take this path when it is one.
#pragma once
#include "neighbours.hpp"
#include "scan.hpp"

namespace hope {

constexpr unsigned long long kGridMaxCells = 1ull << 30;  // 256 MiB of table at most
constexpr int kGridMaxRing = 8;                           // a row's window of 2 ring + 1 cells must fit 32 bits

__device__ __forceinline__ unsigned grid_cell(const float4 p, const FrameInfo& F, float inv_xy, float inv_z)
{
  const int cx = clampi(cell_unclamped(p.x, inv_xy, F.min_b[0]), 0, F.div_b[0] - 1);
  const int cy = clampi(cell_unclamped(p.y, inv_xy, F.min_b[1]), 0, F.div_b[1] - 1);
  const int cz = clampi(cell_unclamped(p.z, inv_z, F.min_b[2]), 0, F.div_b[2] - 1);
  return (unsigned)((cz * F.div_b[1] + cy) * F.div_b[0] + cx);
}

// tab[w] = (occupancy of cells 32 w .. 32 w + 31, occupied cells below 32 w); zeroed by the host.
// The cells ascend along the array (that is what is being checked), so the lanes of a wave that share a word are
// neighbours: their bits are OR-ed along the wave first and the last lane of each run issues the one atomic.
static __global__ void __launch_bounds__(kBlock)
k_grid_mark(const float4* vox, unsigned n, const FrameInfo* fi, float inv_xy, float inv_z, uint2* tab, Globals* g)
{
  const unsigned e = blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & (kWave - 1);
  const FrameInfo& F = fi[0];
  const bool ok = e < n;
  const unsigned c = ok ? grid_cell(vox[e], F, inv_xy, inv_z) : 0xffffffffu;
  unsigned prev = (unsigned)__shfl_up((int)c, 1);
  if (lane == 0) prev = (ok && e > 0u) ? grid_cell(vox[e - 1u], F, inv_xy, inv_z) : 0u;
  if (ok && e > 0u && prev >= c) g->grid_unsorted = 1u;
  const unsigned w = c >> 5;
  unsigned bits = ok ? 1u << (c & 31u) : 0u;
  // segmented inclusive OR over runs of equal w (runs are contiguous when the cells ascend; if they do not, a run
  // is merely cut short and more atomics are issued)
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const unsigned ob = (unsigned)__shfl_up((int)bits, o);
    const unsigned ow = (unsigned)__shfl_up((int)w, o);
    if (lane >= o && ow == w) bits |= ob;
  }
  const unsigned nw = (unsigned)__shfl_down((int)w, 1);
  if (ok && (lane == kWave - 1 || nw != w)) atomicOr(&tab[w].x, bits);
}

static __global__ void __launch_bounds__(kBlock) k_grid_popc(const uint2* tab, unsigned n_words, uint32_t* cnt)
{
  const unsigned w = blockIdx.x * kBlock + threadIdx.x;
  if (w < n_words) cnt[w] = (unsigned)__popc(tab[w].x);
}

static __global__ void __launch_bounds__(kBlock) k_grid_pref(uint2* tab, unsigned n_words, const uint32_t* pref)
{
  const unsigned w = blockIdx.x * kBlock + threadIdx.x;
  if (w < n_words) tab[w].y = pref[w];
}

// the points of the grid in VoxelGrid order; a fetched point carries its index in .w (the tie-break of the
// (distance, index) order)
struct GridCols {
  const float4* vox;
  __device__ __forceinline__ float4 pt(unsigned e) const
  {
    float4 p = vox[e];
    p.w = __uint_as_float(e);
    return p;
  }
  static constexpr bool kLds = false;
  static constexpr bool kNotes = true;
  typedef unsigned cand_t;
  __device__ __forceinline__ unsigned note_base() const { return 0u; }
};

// the run [lo, hi) of vox[] that holds the points of cells base .. base + nx - 1 (one row of the stencil)
__device__ __forceinline__ void grid_row(const uint2 t0, const uint2 t1, unsigned sh, unsigned nx, unsigned& lo,
                                         unsigned& hi)
{
  const unsigned long long bits = (((unsigned long long)t1.x << 32) | (unsigned long long)t0.x) >> sh;
  const unsigned win = (unsigned)bits & ((1u << nx) - 1u);
  lo = hi = 0u;
  if (!win) return;
  const unsigned f = sh + (unsigned)(__ffs((int)win) - 1);  // first occupied cell, as a bit of (t0, t1)
  lo = f < 32u ? t0.y + (unsigned)__popc(t0.x & ((1u << f) - 1u)) : t1.y + (unsigned)__popc(t1.x & ((1u << (f - 32u)) - 1u));
  hi = lo + (unsigned)__popc(win);
}

#ifndef HOPE_GRID_ROWS
#define HOPE_GRID_ROWS 4
#endif
// fn(e, p) for every point in the cells [x0, x1] x [y0, y1] x [z0, z1]; several rows' table entries in flight
template <class Fn>
__device__ __forceinline__ void grid_scan(const GridCols& cols, const uint2* tab, int dx, int dy, int x0, int x1, int y0,
                                          int y1, int z0, int z1, Fn&& fn)
{
  constexpr int kRows = HOPE_GRID_ROWS;  // rows (z levels) whose table entries are in flight together
  constexpr int kPre = 2;                // points per row fetched ahead
  const unsigned nx = (unsigned)(x1 - x0 + 1);
  for (int cy = y0; cy <= y1; ++cy) {
    for (int cz = z0; cz <= z1; cz += kRows) {
      unsigned base[kRows];
      uint2 t0[kRows], t1[kRows];
#pragma unroll
      for (int k = 0; k < kRows; ++k) {
        base[k] = (unsigned)((min(cz + k, z1) * dy + cy) * dx + x0);
        t0[k] = tab[base[k] >> 5];
        t1[k] = tab[(base[k] >> 5) + 1u];
      }
      // the first points of all rows are fetched together (a row holds one or two points more often than not:
      // fetched row by row, every row is a round trip of its own; 1.55 -> 1.40 ms on the 10^8-point scene, three or
      // four points ahead: 1.41 / 1.51)
      unsigned lo[kRows], hi[kRows];
      float4 pp[kRows][kPre];
#pragma unroll
      for (int k = 0; k < kRows; ++k) {
        grid_row(t0[k], t1[k], base[k] & 31u, nx, lo[k], hi[k]);
        if (cz + k > z1) hi[k] = lo[k];
        const bool any = hi[k] > lo[k];
#pragma unroll
        for (int j = 0; j < kPre; ++j) pp[k][j] = cols.pt(any ? min(lo[k] + (unsigned)j, hi[k] - 1u) : 0u);
      }
#pragma unroll
      for (int k = 0; k < kRows; ++k) {
#pragma unroll
        for (int j = 0; j < kPre; ++j)
          if (lo[k] + (unsigned)j < hi[k]) fn(lo[k] + (unsigned)j, pp[k][j]);
        if (lo[k] + (unsigned)kPre < hi[k]) for_run(cols, lo[k] + (unsigned)kPre, hi[k], fn);
      }
    }
  }
}

template <int CAPL, int EXP = 0>  // EXP: timing experiments of tools/micro/part_bench.hip only
__global__ void __launch_bounds__(kBlock, 4)
k_normals_grid(const float4* vox, const uint2* tab, const FrameInfo* fi, float inv_xy, float inv_z, float radius,
               float r2, int ring, unsigned own_lo, unsigned own_hi, unsigned n, float4* normals, float th_norm,
               uint32_t* fit_flag)
{
  static_assert(CAPL <= kNbCap, "the largest sorting network has kNbCap inputs");
  __shared__ unsigned s_note[CAPL * kBlock];
  const unsigned e = blockIdx.x * kBlock + threadIdx.x;
  if (fit_flag && e == 0u) fit_flag[n] = 0u;
  if (e >= n || e < own_lo || e >= own_hi) return;  // (own range: the staged multi-GPU call; no barrier below)
  const FrameInfo& F = fi[0];
  const int dx = F.div_b[0], dy = F.div_b[1], dz = F.div_b[2];
  const GridCols cols{vox};
  const float4 q = cols.pt(e);
  const int cqx = clampi(cell_unclamped(q.x, inv_xy, F.min_b[0]), 0, dx - 1);
  const int cqy = clampi(cell_unclamped(q.y, inv_xy, F.min_b[1]), 0, dy - 1);
  int kl, kr, kd, ku;
  stencil_reach(q, ring, radius, inv_xy, F.slack, kl, kr, kd, ku);
  const int x0 = max(cqx - kl, 0), x1 = min(cqx + kr, dx - 1);
  const int y0 = max(cqy - kd, 0), y1 = min(cqy + ku, dy - 1);
  // z: a neighbour passes ((dx^2 + dy^2) + dz^2) < r2 in float, so |p.z - q.z| <= radius (1 + a few ulp); the cell
  // function is monotone, so its cell lies between the cells of q.z -+ rho, rho a little more than that
  const float rho = radius * 1.0001f + fabsf(q.z) * 1e-6f;
  const int z0 = clampi(cell_unclamped(q.z - rho, inv_z, F.min_b[2]), 0, dz - 1);
  const int z1 = clampi(cell_unclamped(q.z + rho, inv_z, F.min_b[2]), 0, dz - 1);
  unsigned* cand_e = s_note + threadIdx.x;
  int total = 0;
  grid_scan(cols, tab, dx, dy, x0, x1, y0, y1, z0, z1, [&](unsigned pe, const float4& p) {
    if (dist2(q.x, q.y, q.z, p.x, p.y, p.z) < r2) {
      if (total < CAPL) cand_e[total * kBlock] = pe;
      ++total;
    }
  });
  float4 nrm = make_float4(NAN, NAN, NAN, NAN);
  if (EXP == 1) {
    normals[e] = make_float4((float)total, 0.f, 0.f, 0.f);
    return;
  }
  if (total >= 3) {
    float accu[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (total <= CAPL) {
      // wave-uniform choice of the ordering network: the smallest one that holds the longest list of the wave
      if (CAPL <= 8 || !__any(total > 8))
        accumulate_sorted<GridCols, 8>(cols, q, cand_e, total, accu);
      else if (CAPL <= 12 || !__any(total > 12))
        accumulate_sorted<GridCols, 12>(cols, q, cand_e, total, accu);
      else if (CAPL <= 16 || !__any(total > 16))
        accumulate_sorted<GridCols, 16>(cols, q, cand_e, total, accu);
      else
        accumulate_sorted<GridCols, 32>(cols, q, cand_e, total, accu);
    } else {
      // more neighbours than the largest network orders: one scan per pick, in (distance, index) order
      float last_d = -1.f;
      unsigned last_i = 0;
      for (int t = 0; t < total; ++t) {
        float best_d = INFINITY;
        unsigned best_i = 0xffffffffu;
        float bx = 0.f, by = 0.f, bz = 0.f;
        grid_scan(cols, tab, dx, dy, x0, x1, y0, y1, z0, z1, [&](unsigned pe, const float4& p) {
          const float d = dist2(q.x, q.y, q.z, p.x, p.y, p.z);
          if (d < r2 && (t == 0 || di_less(last_d, last_i, d, pe)) && di_less(d, pe, best_d, best_i)) {
            best_d = d;
            best_i = pe;
            bx = p.x;
            by = p.y;
            bz = p.z;
          }
        });
        accu_add(accu, bx, by, bz);
        last_d = best_d;
        last_i = best_i;
      }
    }
    if (EXP == 2) {
      normals[e] = make_float4(accu[0] + accu[1] + accu[2], accu[3] + accu[4] + accu[5], accu[6] + accu[7], accu[8]);
      return;
    }
    nrm = normal_from_sums(accu, total, q);
  }
  normals[e] = nrm;
  if (fit_flag) fit_flag[e] = fabsf(nrm.z) > th_norm ? 1u : 0u;  // NaN fails (k_fit_flags)
}

}  // namespace hope
