// neighbours.hpp — A6 (normals) and A8 (k nearest neighbours) on the column index, tiled
// through LDS.
//
// The down-sampled cloud is sorted by (frame column, z cell); col_first[] gives the range of
// every xy column.  A workgroup owns a 16x16-column tile of one frame: the points of the tile
// plus a 2-column halo are contiguous per column row in the sorted array, so they are copied
// into LDS with ~20 coalesced segment copies, together with a local column table.  Every
// neighbour query of the tile's points is then served from LDS instead of dependent L2 gathers.
// The same search code also runs on a "global" accessor (straight from HBM) for the cases the
// tile cannot serve: a tile with more points than the LDS copy holds, and k-NN rings that leave
// the halo.  Both accessors see the same points in the same arithmetic, so results are identical.
//
// Reference semantics: pcl::NormalEstimation with a radius search (utilities.cpp:428-446) and
// ZGrowing::findPointNeighbours' nearestKSearch (z_growing.cpp:196-226); see kernels.hpp for the
// eigen solver and the distance function.
#pragma once
#include "kernels.hpp"
#include "sortnet.hpp"

namespace hope {

constexpr int kTile = 16;                  // interior columns per tile side
constexpr int kHaloNormals = 2;            // halo columns on each side: the normal radius reaches 2 columns
constexpr int kHaloKnn = 4;                // k-NN: rings 0..4 are served from LDS
constexpr int kTileCap = 1536;             // points (tile + halo) held in LDS: 24 KiB
constexpr int kRunDirect = 24;             // a stencil row with at most this many points is scanned whole

struct GlobalCols {  // the whole frame, straight from HBM
  const float4* spt;
  const uint32_t* col_first;
  const int* fit_rank;
  unsigned col_off;
  int dx;
  __device__ __forceinline__ void range(int cx, int cy, unsigned& lo, unsigned& hi) const
  {
    const unsigned c = col_off + (unsigned)(cy * dx + cx);
    lo = col_first[c];
    hi = col_first[c + 1];
  }
  // the contiguous run of points in columns cx0..cx1 of row cy
  __device__ __forceinline__ void row_range(int cx0, int cx1, int cy, unsigned& lo, unsigned& hi) const
  {
    const unsigned c = col_off + (unsigned)(cy * dx);
    lo = col_first[c + cx0];
    hi = col_first[c + cx1 + 1];
  }
  __device__ __forceinline__ float4 pt(unsigned e) const { return spt[e]; }
  __device__ __forceinline__ int fit(unsigned, const float4& p) const { return fit_rank[__float_as_uint(p.w)]; }
  static constexpr bool kLds = false;
};

template <int HALO>
struct TileCols {  // LDS copy of a tile and its halo
  static constexpr int kTabStride = kTile + 2 * HALO + 1;  // per row: the column starts + the row end
  const float4* s_pt;
  const unsigned* s_tab;
  int xh0, yh0;
  __device__ __forceinline__ void range(int cx, int cy, unsigned& lo, unsigned& hi) const
  {
    const unsigned* t = s_tab + (cy - yh0) * kTabStride + (cx - xh0);
    lo = t[0];
    hi = t[1];
  }
  __device__ __forceinline__ void row_range(int cx0, int cx1, int cy, unsigned& lo, unsigned& hi) const
  {
    const unsigned* t = s_tab + (cy - yh0) * kTabStride - xh0;
    lo = t[cx0];
    hi = t[cx1 + 1];
  }
  __device__ __forceinline__ float4 pt(unsigned e) const { return s_pt[e]; }
  __device__ __forceinline__ int fit(unsigned, const float4& p) const { return __float_as_int(p.w); }
  static constexpr bool kLds = true;
};

template <bool WITH_FIT, int HALO>
struct TileShared {
  static constexpr int kTW = kTile + 2 * HALO;
  static constexpr int kTabStride = kTW + 1;
  float4 pt[kTileCap];  // WITH_FIT: .w holds the fit rank (int bits) instead of the PCL-order index
  unsigned tab[kTW * kTabStride];
  unsigned rowg[kTW + 1];   // global start of each halo row segment
  unsigned rowl[kTW + 1];   // local start (prefix of segment sizes)
  unsigned intg[kTile + 1]; // global start of each interior row segment
  unsigned intp[kTile + 1]; // prefix of interior row sizes
};

struct TileGeom {
  int x0, x1, y0, y1;      // interior columns [x0,x1) x [y0,y1)
  int xh0, xh1, yh0, yh1;  // with halo, clamped to the grid
  unsigned total;          // points in tile + halo
  unsigned n_int;          // points in the interior
  bool in_lds;             // the LDS copy is valid
};

// Loads the size x size column tile at (x0, y0) of frame F with its halo.  Returns false if the
// tile holds no point; g.in_lds tells whether the LDS copy was made (it fits) or not.
template <bool WITH_FIT, int HALO>
__device__ __forceinline__ bool load_tile(const FrameInfo& F, int x0, int y0, int size, const float4* spt,
                                          const uint32_t* col_first, const int* fit_rank,
                                          TileShared<WITH_FIT, HALO>& S, TileGeom& g)
{
  constexpr int kHalo = HALO;
  constexpr int kTabStride = TileShared<WITH_FIT, HALO>::kTabStride;
  const int tid = threadIdx.x;
  const int dx = F.div_b[0], dy = F.div_b[1];
  __syncthreads();  // the previous tile of this block is done with S
  g.x0 = x0;
  g.y0 = y0;
  g.x1 = min(g.x0 + size, dx);
  g.y1 = min(g.y0 + size, dy);
  g.xh0 = max(g.x0 - kHalo, 0);
  g.yh0 = max(g.y0 - kHalo, 0);
  g.xh1 = min(g.x1 + kHalo, dx);
  g.yh1 = min(g.y1 + kHalo, dy);
  const int rows = g.yh1 - g.yh0, width = g.xh1 - g.xh0, irows = g.y1 - g.y0;
  const unsigned col_off = (unsigned)F.col_off;
  if (tid < rows) {
    const unsigned base = col_off + (unsigned)((g.yh0 + tid) * dx);
    const unsigned a = col_first[base + g.xh0], b = col_first[base + g.xh1];
    S.rowg[tid] = a;
    S.rowl[tid] = b - a;
  } else if (tid >= 64 && tid < 64 + irows) {
    const int r = tid - 64;
    const unsigned base = col_off + (unsigned)((g.y0 + r) * dx);
    const unsigned a = col_first[base + g.x0], b = col_first[base + g.x1];
    S.intg[r] = a;
    S.intp[r] = b - a;
  }
  __syncthreads();
  if (tid == 0) {
    unsigned acc = 0;
    for (int r = 0; r < rows; ++r) {
      const unsigned c = S.rowl[r];
      S.rowl[r] = acc;
      acc += c;
    }
    S.rowl[rows] = acc;
  } else if (tid == 64) {
    unsigned acc = 0;
    for (int r = 0; r < irows; ++r) {
      const unsigned c = S.intp[r];
      S.intp[r] = acc;
      acc += c;
    }
    S.intp[irows] = acc;
  }
  __syncthreads();
  g.total = S.rowl[rows];
  g.n_int = S.intp[irows];
  g.in_lds = g.total <= (unsigned)kTileCap;
  if (g.n_int == 0) return false;
  if (!g.in_lds) return true;
  for (int i = tid; i < rows * (width + 1); i += kBlock) {
    const int r = i / (width + 1), c = i - r * (width + 1);
    S.tab[r * kTabStride + c] =
        S.rowl[r] + (col_first[col_off + (unsigned)((g.yh0 + r) * dx + g.xh0 + c)] - S.rowg[r]);
  }
  for (unsigned i = tid; i < g.total; i += kBlock) {
    int r = 0;
    while (S.rowl[r + 1] <= i) ++r;
    float4 p = spt[S.rowg[r] + (i - S.rowl[r])];
    if (WITH_FIT) p.w = __int_as_float(fit_rank[__float_as_uint(p.w)]);
    S.pt[i] = p;
  }
  __syncthreads();
  return true;
}

// i-th interior point of the tile: position in the sorted array and in the LDS copy
template <bool WITH_FIT, int HALO>
__device__ __forceinline__ void interior_point(const TileShared<WITH_FIT, HALO>& S, const TileGeom& g, unsigned i,
                                               unsigned& e_glob, unsigned& e_loc)
{
  int r = 0;
  while (S.intp[r + 1] <= i) ++r;
  e_glob = S.intg[r] + (i - S.intp[r]);
  const int rh = r + (g.y0 - g.yh0);
  e_loc = S.rowl[rh] + (e_glob - S.rowg[rh]);
}

// Runs body(g) for the 16x16 tile `tile` of frame F, or, when the tile and its halo hold more
// points than the LDS copy (several z layers per column: noisy far range, walls), for its 8x8
// and then 4x4 sub-tiles; a 4x4 sub-tile that still does not fit is handed to body with
// g.in_lds == false (the global accessor).  Every branch is block-uniform.
template <bool WITH_FIT, int HALO, class Body>
__device__ __forceinline__ void tile_cascade(const FrameInfo& F, int tile, const float4* spt, const uint32_t* col_first,
                                             const int* fit_rank, TileShared<WITH_FIT, HALO>& S, Body&& body)
{
  const int dx = F.div_b[0], dy = F.div_b[1];
  const int ntx = (dx + kTile - 1) / kTile;
  const int ty = tile / ntx, tx = tile - ty * ntx;
  const int X0 = tx * kTile, Y0 = ty * kTile;
  TileGeom g;
  if (!load_tile<WITH_FIT, HALO>(F, X0, Y0, kTile, spt, col_first, fit_rank, S, g)) return;
  if (g.in_lds) {
    body(g);
    return;
  }
  for (int a = 0; a < 4; ++a) {
    const int x8 = X0 + (a & 1) * 8, y8 = Y0 + (a >> 1) * 8;
    if (x8 >= dx || y8 >= dy) continue;
    if (!load_tile<WITH_FIT, HALO>(F, x8, y8, 8, spt, col_first, fit_rank, S, g)) continue;
    if (g.in_lds) {
      body(g);
      continue;
    }
    for (int b = 0; b < 4; ++b) {
      const int x4 = x8 + (b & 1) * 4, y4 = y8 + (b >> 1) * 4;
      if (x4 >= dx || y4 >= dy) continue;
      if (!load_tile<WITH_FIT, HALO>(F, x4, y4, 4, spt, col_first, fit_rank, S, g)) continue;
      body(g);
    }
  }
}

// ---- scanning a stencil row inside a z window ---------------------------------------------------
// The entries of one column ascend in z cell, hence (cells being leaf_z tall) in z up to one
// cell: with a margin of 1.5 cells a binary search on z and a break on z are exact.  Short rows
// (the rule on horizontal surfaces) are scanned whole; only tall columns (walls, box sides) pay
// for the search.
template <class Cols>
__device__ __forceinline__ unsigned lower_bound_z(const Cols& cols, unsigned lo, unsigned hi, float z)
{
  while (lo < hi) {
    const unsigned mid = (lo + hi) >> 1;
    if (cols.pt(mid).z < z) lo = mid + 1; else hi = mid;
  }
  return lo;
}

template <class Cols, class Fn>
__device__ __forceinline__ void for_window(const Cols& cols, int x0, int x1, int cy, float zlo, float zhi, Fn&& fn)
{
  unsigned lo, hi;
  cols.row_range(x0, x1, cy, lo, hi);
  if (hi - lo <= (unsigned)kRunDirect) {
    for (unsigned e = lo; e < hi; ++e) fn(e, cols.pt(e));
    return;
  }
  for (int cx = x0; cx <= x1; ++cx) {
    cols.range(cx, cy, lo, hi);
    if (hi - lo > 8u) lo = lower_bound_z(cols, lo, hi, zlo);
    for (unsigned e = lo; e < hi; ++e) {
      const float4 p = cols.pt(e);
      if (p.z > zhi) break;
      fn(e, p);
    }
  }
}

// ---- A6: one normal ------------------------------------------------------------------------
// Neighbours with dist2 < r2 inside the (2*ring+1)^2 column stencil, accumulated in
// (dist2, index) order (computeMeanAndCovarianceMatrix, PCL 1.8.1 centroid.hpp, float, dense
// branch), then pcl::eigen33's smallest eigenvector.
//
// LDS accessor, at most 32 neighbours (a radius of ~one cell finds 5..10 on a horizontal
// surface, 15..25 on a wall): their positions are noted in the thread's LDS slice during the
// scan, fetched into registers with independent reads, ordered by a 16- or 32-input sorting
// network (no data-dependent branches) and accumulated in that order.  More neighbours, or the
// global accessor: re-scan the stencil for every pick.
constexpr int kNbCap = 32;

__device__ __forceinline__ void accu_add(float* accu, float x, float y, float z)
{
  accu[0] += x * x;
  accu[1] += x * y;
  accu[2] += x * z;
  accu[3] += y * y;
  accu[4] += y * z;
  accu[5] += z * z;
  accu[6] += x;
  accu[7] += y;
  accu[8] += z;
}

template <class Cols>
__device__ __forceinline__ void accumulate_by_rescan(const Cols& cols, const float4 q, int x0, int x1, int y0, int y1,
                                                     float zlo, float zhi, float r2, int total, float* accu)
{
  float last_d = -1.f;
  unsigned last_i = 0;
  for (int t = 0; t < total; ++t) {
    float best_d = INFINITY;
    unsigned best_i = 0xffffffffu;
    float bx = 0.f, by = 0.f, bz = 0.f;
    for (int cy = y0; cy <= y1; ++cy) {
      for_window(cols, x0, x1, cy, zlo, zhi, [&](unsigned, const float4& p) {
        const float d = dist2(q.x, q.y, q.z, p.x, p.y, p.z);
        const unsigned pi = __float_as_uint(p.w);
        if (d < r2 && (t == 0 || di_less(last_d, last_i, d, pi)) && di_less(d, pi, best_d, best_i)) {
          best_d = d;
          best_i = pi;
          bx = p.x;
          by = p.y;
          bz = p.z;
        }
      });
    }
    accu_add(accu, bx, by, bz);
    last_d = best_d;
    last_i = best_i;
  }
}

// compare-exchange on (dist2, index); the index is fetched only for equal distances
#define HOPE_CE(a, b)                                                                 \
  {                                                                                   \
    bool sw = cd[b] < cd[a];                                                          \
    if (cd[b] == cd[a] && cd[a] < INFINITY)                                           \
      sw = __float_as_uint(cols.pt(ce[b]).w) < __float_as_uint(cols.pt(ce[a]).w);     \
    const float td = sw ? cd[b] : cd[a];                                              \
    const unsigned te = sw ? ce[b] : ce[a];                                           \
    cd[b] = sw ? cd[a] : cd[b];                                                       \
    ce[b] = sw ? ce[a] : ce[b];                                                       \
    cd[a] = td;                                                                       \
    ce[a] = te;                                                                       \
  }

template <class Cols, int N>
__device__ __forceinline__ void accumulate_sorted(const Cols& cols, const float4 q, const unsigned short* cand_e,
                                                  int total, float* accu)
{
  static_assert(N == 16 || N == 32, "sorting networks exist for 16 and 32 inputs");
  float cd[N];
  unsigned ce[N];
#pragma unroll
  for (int s = 0; s < N; ++s) ce[s] = s < total ? (unsigned)cand_e[s * kBlock] : 0u;
#pragma unroll
  for (int s = 0; s < N; ++s) {
    const float4 p = cols.pt(ce[s]);
    cd[s] = s < total ? dist2(q.x, q.y, q.z, p.x, p.y, p.z) : INFINITY;
  }
  if (N == 16) {
    HOPE_SORT_NET16
  } else {
    HOPE_SORT_NET32
  }
#pragma unroll
  for (int s = 0; s < N; ++s) {
    if (s < total) {
      const float4 p = cols.pt(ce[s]);
      accu_add(accu, p.x, p.y, p.z);
    }
  }
}
#undef HOPE_CE

template <class Cols, int CAPL>
__device__ __forceinline__ float4 estimate_normal(const Cols& cols, const float4 q, int cqx, int cqy, int dx, int dy,
                                                  int ring, float radius, float r2, float leaf_z,
                                                  unsigned short* cand_e, int ablate = 0)
{
  static_assert(CAPL <= kNbCap, "the largest sorting network has kNbCap inputs");
  const int x0 = max(cqx - ring, 0), x1 = min(cqx + ring, dx - 1);
  const int y0 = max(cqy - ring, 0), y1 = min(cqy + ring, dy - 1);
  const float zlo = q.z - radius - 1.5f * leaf_z, zhi = q.z + radius + 1.5f * leaf_z;
  int total = 0;
  for (int cy = y0; cy <= y1; ++cy) {
    for_window(cols, x0, x1, cy, zlo, zhi, [&](unsigned e, const float4& p) {
      if (dist2(q.x, q.y, q.z, p.x, p.y, p.z) < r2) {
        if (Cols::kLds && total < CAPL) cand_e[total * kBlock] = (unsigned short)e;
        ++total;
      }
    });
  }
  if (total < 3) return make_float4(NAN, NAN, NAN, NAN);
  if (ablate == 2) return make_float4((float)total, 0.f, 0.f, 0.f);
  float accu[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (Cols::kLds && total <= CAPL) {
    // wave-uniform choice of the network: 16 inputs unless some lane of the wave has more
    if (CAPL <= 16 || !__any(total > 16))
      accumulate_sorted<Cols, 16>(cols, q, cand_e, total, accu);
    else
      accumulate_sorted<Cols, 32>(cols, q, cand_e, total, accu);
  } else {
    accumulate_by_rescan(cols, q, x0, x1, y0, y1, zlo, zhi, r2, total, accu);
  }
  const float cnt = (float)total;
#pragma unroll
  for (int k = 0; k < 9; ++k) accu[k] /= cnt;
  if (ablate == 3) return make_float4(accu[0] + accu[1] + accu[2], accu[3] + accu[4] + accu[5], accu[6] + accu[7], accu[8]);
  float cov[9];
  cov[0] = accu[0] - accu[6] * accu[6];
  cov[1] = accu[1] - accu[6] * accu[7];
  cov[2] = accu[2] - accu[6] * accu[8];
  cov[4] = accu[3] - accu[7] * accu[7];
  cov[5] = accu[4] - accu[7] * accu[8];
  cov[8] = accu[5] - accu[8] * accu[8];
  cov[3] = cov[1];
  cov[6] = cov[2];
  cov[7] = cov[5];
  float eval, evec[3];
  eigen33_smallest(cov, eval, evec);
  float nx = evec[0], ny = evec[1], nz = evec[2];
  const float eig_sum = cov[0] + cov[4] + cov[8];
  float curvature = 0.f;
  if (eig_sum != 0.f) curvature = fabsf(eval / eig_sum);
  // flipNormalTowardsViewpoint(point, 0, 0, 0, ...)   (PCL 1.8.1 normal_3d.h)
  const float vpx = 0.f - q.x, vpy = 0.f - q.y, vpz = 0.f - q.z;
  const float cos_theta = (vpx * nx + vpy * ny + vpz * nz);
  if (cos_theta < 0.f) {
    nx *= -1.f;
    ny *= -1.f;
    nz *= -1.f;
  }
  return make_float4(nx, ny, nz, curvature);
}

// grid = (max tiles per frame, n_frames)
template <int CAPL>
__global__ void __launch_bounds__(kBlock)
k_normals_tile(const float4* spt, const uint32_t* col_first, const FrameInfo* fi, float inv_xy, float inv_z,
               float radius, float r2, int ring, float4* normals, int ablate)
{
  __shared__ TileShared<false, kHaloNormals> S;
  __shared__ unsigned short s_cand_e[CAPL * kBlock];
  const FrameInfo& F = fi[blockIdx.y];
  const int dx = F.div_b[0], dy = F.div_b[1];
  if (F.n_vox <= 0 || (int)blockIdx.x >= ((dx + kTile - 1) / kTile) * ((dy + kTile - 1) / kTile)) return;
  const float leaf_z = 1.0f / inv_z;
  unsigned short* cand_e = s_cand_e + threadIdx.x;
  const GlobalCols gc{spt, col_first, nullptr, (unsigned)F.col_off, dx};
  tile_cascade<false, kHaloNormals>(F, (int)blockIdx.x, spt, col_first, nullptr, S, [&](const TileGeom& g) {
    if (ablate == 1) return;
    const bool lds = g.in_lds && ring <= kHaloNormals;
    const TileCols<kHaloNormals> tc{S.pt, S.tab, g.xh0, g.yh0};
    for (unsigned i = threadIdx.x; i < g.n_int; i += kBlock) {
      unsigned eg, el;
      interior_point(S, g, i, eg, el);
      const float4 q = spt[eg];
      const int cqx = clampi(cell_unclamped(q.x, inv_xy, F.min_b[0]), 0, dx - 1);
      const int cqy = clampi(cell_unclamped(q.y, inv_xy, F.min_b[1]), 0, dy - 1);
      float4 out;
      if (lds)
        out = estimate_normal<TileCols<kHaloNormals>, CAPL>(tc, q, cqx, cqy, dx, dy, ring, radius, r2, leaf_z, cand_e,
                                                            ablate);
      else
        out = estimate_normal<GlobalCols, CAPL>(gc, q, cqx, cqy, dx, dy, ring, radius, r2, leaf_z, cand_e, ablate);
      normals[__float_as_uint(q.w)] = out;
    }
  });
}

// ---- A8: exact k nearest neighbours among the fit points ---------------------------------------
// The sorted list holds 64-bit keys (dist2 bits << 32 | fit rank): distances are >= 0, so the
// unsigned order of the keys is the (distance, rank) order, and ties go to the smaller rank
// (= smaller index: the compaction keeps the order).  A candidate that cannot enter the full list
// is rejected with one compare; one that can replaces the last entry and bubbles up.
typedef unsigned long long knn_key;
constexpr knn_key kKnnEmpty = ((knn_key)0x7f800000u << 32) | 0xffffffffu;  // (+inf, no rank)

__device__ __forceinline__ float key_dist(knn_key k) { return __uint_as_float((unsigned)(k >> 32)); }

template <int KMAX>
__device__ __forceinline__ knn_key knn_worst(int k, const knn_key (&lst)[KMAX])
{
  if (k == KMAX) return lst[KMAX - 1];
  knn_key w = kKnnEmpty;
#pragma unroll
  for (int j = 0; j < KMAX; ++j)
    if (j == k - 1) w = lst[j];
  return w;
}

template <class Cols, int KMAX, bool DEDUP = false>
__device__ __forceinline__ void knn_offer(const Cols& cols, const float4 q, unsigned e, const float4& p, int k,
                                          knn_key (&lst)[KMAX])
{
  const int pr = cols.fit(e, p);
  if (pr < 0) return;
  const float d = dist2(q.x, q.y, q.z, p.x, p.y, p.z);
  const knn_key key = ((knn_key)__float_as_uint(d) << 32) | (unsigned)pr;
  if (!(key < knn_worst<KMAX>(k, lst))) return;
  if (DEDUP) {  // a second sweep over points that may already be in the list
    bool dup = false;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) dup |= (lst[j] == key);
    if (dup) return;
  }
  // replace the k-th entry, then bubble up (compile-time register indices)
#pragma unroll
  for (int j = 0; j < KMAX; ++j)
    if (j == k - 1) lst[j] = key;
#pragma unroll
  for (int j = KMAX - 1; j >= 1; --j) {
    if (j < k) {
      const knn_key a = lst[j - 1], b = lst[j];
      const bool sw = b < a;
      lst[j - 1] = sw ? b : a;
      lst[j] = sw ? a : b;
    }
  }
}

template <int KMAX>
__device__ __forceinline__ float knn_kth(int k, const knn_key (&lst)[KMAX])
{
  return key_dist(knn_worst<KMAX>(k, lst));
}

// every column outside ring R is more than (R - slack) cells away in x or y
__device__ __forceinline__ float ring_reach(int R, float slack, float cell_xy)
{
  return ((float)R - slack) * cell_xy * 0.999999f;
}

// Ring expansion R = r_begin .. r_end until the k-th distance is provably smaller than the
// distance to any unvisited column; returns true when it is.  The two edge rows of a ring are
// contiguous runs, the rows between contribute their two end columns.  Once the list is full only
// |dz| <= sqrt(k-th) can still enter, which bounds the z window of every further scan.
template <class Cols, int KMAX, bool DEDUP>
__device__ __forceinline__ bool knn_rings(const Cols& cols, const float4 q, int cqx, int cqy, int dx, int dy, int r_begin,
                                          int r_end, int k, float slack, float cell_xy, float leaf_z,
                                          knn_key (&lst)[KMAX])
{
  for (int R = r_begin; R <= r_end; ++R) {
    const int y0 = cqy - R, y1 = cqy + R, x0 = cqx - R, x1 = cqx + R;
    const int xa = max(x0, 0), xb = min(x1, dx - 1);
    for (int cy = max(y0, 0); cy <= min(y1, dy - 1); ++cy) {
      const float worst = knn_kth<KMAX>(k, lst);
      const float rz = worst < INFINITY ? __builtin_sqrtf(worst) * 1.000001f + 1.5f * leaf_z : INFINITY;
      const float zlo = q.z - rz, zhi = q.z + rz;
      auto offer = [&](unsigned e, const float4& p) { knn_offer<Cols, KMAX, DEDUP>(cols, q, e, p, k, lst); };
      if (cy == y0 || cy == y1) {
        for_window(cols, xa, xb, cy, zlo, zhi, offer);
      } else {
        if (x0 >= 0) for_window(cols, x0, x0, cy, zlo, zhi, offer);
        if (x1 < dx && x1 != x0) for_window(cols, x1, x1, cy, zlo, zhi, offer);
      }
    }
    const float reach = ring_reach(R, slack, cell_xy);
    if (reach > 0.f && knn_kth<KMAX>(k, lst) < reach * reach * 0.999999f) return true;
  }
  return false;
}

template <int KMAX>
__global__ void __launch_bounds__(kBlock)
k_knn_tile(const float4* spt, const uint32_t* col_first, const int* fit_rank, const FrameInfo* fi, float inv_xy,
           float inv_z, float cell_xy, int K, int* knn_out, int ablate)
{
  typedef TileCols<kHaloKnn> Tc;
  __shared__ TileShared<true, kHaloKnn> S;
  const FrameInfo& F = fi[blockIdx.y];
  const int dx = F.div_b[0], dy = F.div_b[1];
  if (F.n_fit <= 0 || (int)blockIdx.x >= ((dx + kTile - 1) / kTile) * ((dy + kTile - 1) / kTile)) return;
  const float leaf_z = 1.0f / inv_z;
  const int k = min(K, F.n_fit);
  const GlobalCols gc{spt, col_first, fit_rank, (unsigned)F.col_off, dx};
  // Fast pass: the 5x5 square is ring 2; a list that settles there holds only neighbours closer
  // than reach2, so a z window of that height loses nothing.
  const float reach2 = ring_reach(2, F.slack, cell_xy);
  tile_cascade<true, kHaloKnn>(F, (int)blockIdx.x, spt, col_first, fit_rank, S, [&](const TileGeom& g) {
    const Tc tc{S.pt, S.tab, g.xh0, g.yh0};
    if (ablate == 1) return;
    for (unsigned i = threadIdx.x; i < g.n_int; i += kBlock) {
      unsigned eg, el;
      interior_point(S, g, i, eg, el);
      const float4 q = spt[eg];
      const int me = fit_rank[__float_as_uint(q.w)];
      if (me < 0) continue;
      const int cqx = clampi(cell_unclamped(q.x, inv_xy, F.min_b[0]), 0, dx - 1);
      const int cqy = clampi(cell_unclamped(q.y, inv_xy, F.min_b[1]), 0, dy - 1);
      knn_key lst[KMAX];
#pragma unroll
      for (int j = 0; j < KMAX; ++j) lst[j] = kKnnEmpty;
      const int max_ring = max(max(cqx, dx - 1 - cqx), max(cqy, dy - 1 - cqy));
      bool done = false;
      int next_ring = 0;
      if (g.in_lds) {
        if (reach2 > 0.f) {
          const int x0 = max(cqx - 2, 0), x1 = min(cqx + 2, dx - 1);
          const float zlo = q.z - reach2 - 1.5f * leaf_z, zhi = q.z + reach2 + 1.5f * leaf_z;
          for (int cy = max(cqy - 2, 0); cy <= min(cqy + 2, dy - 1); ++cy)
            for_window(tc, x0, x1, cy, zlo, zhi,
                       [&](unsigned e, const float4& p) { knn_offer<Tc, KMAX>(tc, q, e, p, k, lst); });
          done = knn_kth<KMAX>(k, lst) < reach2 * reach2 * 0.999999f;
        }
        if (ablate == 2) done = true;
        if (!done) {
          // sparse neighbourhood: the exact ring expansion over the LDS copy (every z; points the
          // fast pass has already listed are skipped), then over the whole frame
          done = knn_rings<Tc, KMAX, true>(tc, q, cqx, cqy, dx, dy, 0, min(kHaloKnn, max_ring), k, F.slack, cell_xy,
                                           leaf_z, lst);
          next_ring = kHaloKnn + 1;
        }
      }
      if (ablate == 3) done = true;
      if (!done && next_ring <= max_ring)
        knn_rings<GlobalCols, KMAX, false>(gc, q, cqx, cqy, dx, dy, next_ring, max_ring, k, F.slack, cell_xy, leaf_z,
                                           lst);
#pragma unroll
      for (int j = 0; j < KMAX; ++j)
        if (j < K) knn_out[(size_t)me * K + j] = (j < k) ? (int)(unsigned)lst[j] : -1;
    }
  });
}

}  // namespace hope
