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

namespace hope {

constexpr int kTile = 16;                  // interior columns per tile side
constexpr int kHalo = 2;                   // halo columns on each side
constexpr int kTW = kTile + 2 * kHalo;     // 20
constexpr int kTabStride = kTW + 1;        // per row: kTW column starts + the row end
constexpr int kTileCap = 1024;             // points (tile + halo) held in LDS: 16 KiB

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

struct TileCols {  // LDS copy of a tile and its halo
  const float4* s_pt;
  const unsigned* s_tab;
  const int* s_fit;
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
  __device__ __forceinline__ int fit(unsigned e, const float4&) const { return s_fit[e]; }
  static constexpr bool kLds = true;
};

template <bool WITH_FIT>
struct TileShared {
  float4 pt[kTileCap];
  int fit[WITH_FIT ? kTileCap : 1];
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

// Loads the tile `tile` of frame F.  Returns false if the block has nothing to do.
template <bool WITH_FIT>
__device__ __forceinline__ bool load_tile(const FrameInfo& F, int tile, const float4* spt, const uint32_t* col_first,
                                          const int* fit_rank, TileShared<WITH_FIT>& S, TileGeom& g)
{
  const int tid = threadIdx.x;
  const int dx = F.div_b[0], dy = F.div_b[1];
  const int ntx = (dx + kTile - 1) / kTile;
  const int ty = tile / ntx, tx = tile - ty * ntx;
  g.x0 = tx * kTile;
  g.y0 = ty * kTile;
  g.x1 = min(g.x0 + kTile, dx);
  g.y1 = min(g.y0 + kTile, dy);
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
    const float4 p = spt[S.rowg[r] + (i - S.rowl[r])];
    S.pt[i] = p;
    if (WITH_FIT) S.fit[i] = fit_rank[__float_as_uint(p.w)];
  }
  __syncthreads();
  return true;
}

// i-th interior point of the tile: position in the sorted array and in the LDS copy
template <bool WITH_FIT>
__device__ __forceinline__ void interior_point(const TileShared<WITH_FIT>& S, const TileGeom& g, unsigned i,
                                               unsigned& e_glob, unsigned& e_loc)
{
  int r = 0;
  while (S.intp[r + 1] <= i) ++r;
  e_glob = S.intg[r] + (i - S.intp[r]);
  const int rh = r + (g.y0 - g.yh0);
  e_loc = S.rowl[rh] + (e_glob - S.rowg[rh]);
}

// ---- A6: one normal ------------------------------------------------------------------------
// Neighbours with dist2 < r2 inside the (2*ring+1)^2 column stencil, accumulated in
// (dist2, index) order (computeMeanAndCovarianceMatrix, PCL 1.8.1 centroid.hpp, float, dense
// branch), then pcl::eigen33's smallest eigenvector.  The 2*ring+1 columns of a stencil row are
// one contiguous run of points.
//
// LDS accessor, at most kNbCap neighbours (the rule: a radius of ~one cell finds 5..10): their
// positions are noted in the thread's LDS slice during the scan, then fetched into registers with
// independent reads, ordered by a 16-input sorting network (Batcher odd-even merge, 63
// compare-exchanges, no data-dependent branches) and accumulated in that order.  More
// neighbours, or the global accessor: re-scan the stencil for every pick.
constexpr int kNbCap = 16;

template <class Cols>
__device__ __forceinline__ void accumulate_by_rescan(const Cols& cols, const float4 q, int x0, int x1, int y0, int y1,
                                                     float r2, int total, float* accu)
{
  float last_d = -1.f;
  unsigned last_i = 0;
  for (int t = 0; t < total; ++t) {
    float best_d = INFINITY;
    unsigned best_i = 0xffffffffu;
    float bx = 0.f, by = 0.f, bz = 0.f;
    for (int cy = y0; cy <= y1; ++cy) {
      unsigned lo, hi;
      cols.row_range(x0, x1, cy, lo, hi);
      for (unsigned e = lo; e < hi; ++e) {
        const float4 p = cols.pt(e);
        const float d = dist2(q.x, q.y, q.z, p.x, p.y, p.z);
        const unsigned pi = __float_as_uint(p.w);
        if (d < r2 && (t == 0 || di_less(last_d, last_i, d, pi)) && di_less(d, pi, best_d, best_i)) {
          best_d = d;
          best_i = pi;
          bx = p.x;
          by = p.y;
          bz = p.z;
        }
      }
    }
    accu[0] += bx * bx;
    accu[1] += bx * by;
    accu[2] += bx * bz;
    accu[3] += by * by;
    accu[4] += by * bz;
    accu[5] += bz * bz;
    accu[6] += bx;
    accu[7] += by;
    accu[8] += bz;
    last_d = best_d;
    last_i = best_i;
  }
}

template <class Cols, int CAPL>
__device__ __forceinline__ float4 estimate_normal(const Cols& cols, const float4 q, int cqx, int cqy, int dx, int dy,
                                                  int ring, float r2, unsigned short* cand_e, int ablate = 0)
{
  static_assert(CAPL <= kNbCap, "the sorting network has kNbCap inputs");
  const int x0 = max(cqx - ring, 0), x1 = min(cqx + ring, dx - 1);
  const int y0 = max(cqy - ring, 0), y1 = min(cqy + ring, dy - 1);
  int total = 0;
  for (int cy = y0; cy <= y1; ++cy) {
    unsigned lo, hi;
    cols.row_range(x0, x1, cy, lo, hi);
    for (unsigned e = lo; e < hi; ++e) {
      const float4 p = cols.pt(e);
      if (dist2(q.x, q.y, q.z, p.x, p.y, p.z) < r2) {
        if (Cols::kLds && total < CAPL) cand_e[total * kBlock] = (unsigned short)e;
        ++total;
      }
    }
  }
  if (total < 3) return make_float4(NAN, NAN, NAN, NAN);
  if (ablate == 2) return make_float4((float)total, 0.f, 0.f, 0.f);
  float accu[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (Cols::kLds && total <= CAPL) {
    float cd[kNbCap];
    unsigned ce[kNbCap];
#pragma unroll
    for (int s = 0; s < kNbCap; ++s) ce[s] = (s < CAPL && s < total) ? (unsigned)cand_e[s * kBlock] : 0u;
#pragma unroll
    for (int s = 0; s < kNbCap; ++s) {
      const float4 p = cols.pt(ce[s]);
      cd[s] = s < total ? dist2(q.x, q.y, q.z, p.x, p.y, p.z) : INFINITY;
    }
    // compare-exchange on (dist2, index); the index is fetched only for equal distances
#define HOPE_CE(a, b)                                                                               \
  {                                                                                                 \
    bool sw = cd[b] < cd[a];                                                                        \
    if (cd[b] == cd[a] && cd[a] < INFINITY)                                                         \
      sw = __float_as_uint(cols.pt(ce[b]).w) < __float_as_uint(cols.pt(ce[a]).w);                   \
    const float td = sw ? cd[b] : cd[a];                                                            \
    const unsigned te = sw ? ce[b] : ce[a];                                                         \
    cd[b] = sw ? cd[a] : cd[b];                                                                     \
    ce[b] = sw ? ce[a] : ce[b];                                                                     \
    cd[a] = td;                                                                                     \
    ce[a] = te;                                                                                     \
  }
  HOPE_CE(0, 1) HOPE_CE(2, 3) HOPE_CE(0, 2) HOPE_CE(1, 3) HOPE_CE(1, 2) HOPE_CE(4, 5) HOPE_CE(6, 7)
  HOPE_CE(4, 6) HOPE_CE(5, 7) HOPE_CE(5, 6) HOPE_CE(0, 4) HOPE_CE(2, 6) HOPE_CE(2, 4) HOPE_CE(1, 5)
  HOPE_CE(3, 7) HOPE_CE(3, 5) HOPE_CE(1, 2) HOPE_CE(3, 4) HOPE_CE(5, 6) HOPE_CE(8, 9) HOPE_CE(10, 11)
  HOPE_CE(8, 10) HOPE_CE(9, 11) HOPE_CE(9, 10) HOPE_CE(12, 13) HOPE_CE(14, 15) HOPE_CE(12, 14)
  HOPE_CE(13, 15) HOPE_CE(13, 14) HOPE_CE(8, 12) HOPE_CE(10, 14) HOPE_CE(10, 12) HOPE_CE(9, 13)
  HOPE_CE(11, 15) HOPE_CE(11, 13) HOPE_CE(9, 10) HOPE_CE(11, 12) HOPE_CE(13, 14) HOPE_CE(0, 8) HOPE_CE(4, 12)
  HOPE_CE(4, 8) HOPE_CE(2, 10) HOPE_CE(6, 14) HOPE_CE(6, 10) HOPE_CE(2, 4) HOPE_CE(6, 8) HOPE_CE(10, 12)
  HOPE_CE(1, 9) HOPE_CE(5, 13) HOPE_CE(5, 9) HOPE_CE(3, 11) HOPE_CE(7, 15) HOPE_CE(7, 11) HOPE_CE(3, 5)
  HOPE_CE(7, 9) HOPE_CE(11, 13) HOPE_CE(1, 2) HOPE_CE(3, 4) HOPE_CE(5, 6) HOPE_CE(7, 8) HOPE_CE(9, 10)
  HOPE_CE(11, 12) HOPE_CE(13, 14)
#undef HOPE_CE
#pragma unroll
    for (int s = 0; s < kNbCap; ++s) {
      if (s < total) {
        const float4 p = cols.pt(ce[s]);
        accu[0] += p.x * p.x;
        accu[1] += p.x * p.y;
        accu[2] += p.x * p.z;
        accu[3] += p.y * p.y;
        accu[4] += p.y * p.z;
        accu[5] += p.z * p.z;
        accu[6] += p.x;
        accu[7] += p.y;
        accu[8] += p.z;
      }
    }
  } else {
    accumulate_by_rescan(cols, q, x0, x1, y0, y1, r2, total, accu);
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
k_normals_tile(const float4* spt, const uint32_t* col_first, const FrameInfo* fi, float inv_xy, float r2, int ring,
               float4* normals, int ablate)
{
  __shared__ TileShared<false> S;
  __shared__ unsigned short s_cand_e[CAPL * kBlock];
  const FrameInfo& F = fi[blockIdx.y];
  const int dx = F.div_b[0], dy = F.div_b[1];
  if (F.n_vox <= 0 || (int)blockIdx.x >= ((dx + kTile - 1) / kTile) * ((dy + kTile - 1) / kTile)) return;
  TileGeom g;
  if (!load_tile<false>(F, (int)blockIdx.x, spt, col_first, nullptr, S, g)) return;
  if (ablate == 1) return;
  const bool lds = g.in_lds && ring <= kHalo;
  unsigned short* cand_e = s_cand_e + threadIdx.x;
  const TileCols tc{S.pt, S.tab, S.fit, g.xh0, g.yh0};
  const GlobalCols gc{spt, col_first, nullptr, (unsigned)F.col_off, dx};
  for (unsigned i = threadIdx.x; i < g.n_int; i += kBlock) {
    unsigned eg, el;
    interior_point(S, g, i, eg, el);
    const float4 q = spt[eg];
    const int cqx = clampi(cell_unclamped(q.x, inv_xy, F.min_b[0]), 0, dx - 1);
    const int cqy = clampi(cell_unclamped(q.y, inv_xy, F.min_b[1]), 0, dy - 1);
    float4 out;
    if (lds)
      out = estimate_normal<TileCols, CAPL>(tc, q, cqx, cqy, dx, dy, ring, r2, cand_e, ablate);
    else
      out = estimate_normal<GlobalCols, CAPL>(gc, q, cqx, cqy, dx, dy, ring, r2, cand_e, ablate);
    normals[__float_as_uint(q.w)] = out;
  }
}

// ---- A8: exact k nearest neighbours among the fit points ---------------------------------------
// bd/bi: the sorted list (distance, fit rank); ties go to the smaller rank (= smaller index: the
// compaction keeps the order).  A candidate that cannot enter the full list is rejected before
// the insertion network.
template <class Cols, int KMAX>
__device__ __forceinline__ void knn_scan(const Cols& cols, const float4 q, unsigned lo, unsigned hi, int k,
                                         float (&bd)[KMAX], int (&bi)[KMAX])
{
  for (unsigned e = lo; e < hi; ++e) {
    const float4 p = cols.pt(e);
    const int pr = cols.fit(e, p);
    if (pr < 0) continue;
    const float d = dist2(q.x, q.y, q.z, p.x, p.y, p.z);
    float wd = INFINITY;
    int wi = -1;
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
      if (j == k - 1) {
        wd = bd[j];
        wi = bi[j];
      }
    if (!di_less(d, (unsigned)pr, wd, (unsigned)wi)) continue;
    // sorted insertion with compile-time register indices
#pragma unroll
    for (int j = KMAX - 1; j >= 1; --j) {
      if (j < k) {
        const bool before_prev = di_less(d, (unsigned)pr, bd[j - 1], (unsigned)bi[j - 1]);
        const bool here = di_less(d, (unsigned)pr, bd[j], (unsigned)bi[j]);
        const float nd = before_prev ? bd[j - 1] : (here ? d : bd[j]);
        const int ni = before_prev ? bi[j - 1] : (here ? pr : bi[j]);
        bd[j] = nd;
        bi[j] = ni;
      }
    }
    if (di_less(d, (unsigned)pr, bd[0], (unsigned)bi[0])) {
      bd[0] = d;
      bi[0] = pr;
    }
  }
}

// true iff the k-th distance is provably smaller than the distance to any column outside ring R
template <int KMAX>
__device__ __forceinline__ bool knn_settled(int R, int k, float slack, float cell_xy, const float (&bd)[KMAX])
{
  // every unvisited column is more than (R - slack) cells away in x or y
  const float reach = ((float)R - slack) * cell_xy * 0.999999f;
  float kth = INFINITY;
#pragma unroll
  for (int j = 0; j < KMAX; ++j)
    if (j == k - 1) kth = bd[j];
  return reach > 0.f && kth < reach * reach * 0.999999f;
}

// ring expansion R = r_begin .. r_end: the two edge rows of a ring are contiguous runs, the
// rows between contribute their two end columns
template <class Cols, int KMAX>
__device__ __forceinline__ bool knn_rings(const Cols& cols, const float4 q, int cqx, int cqy, int dx, int dy, int r_begin,
                                          int r_end, int k, float slack, float cell_xy, float (&bd)[KMAX],
                                          int (&bi)[KMAX])
{
  for (int R = r_begin; R <= r_end; ++R) {
    const int y0 = cqy - R, y1 = cqy + R, x0 = cqx - R, x1 = cqx + R;
    const int xa = max(x0, 0), xb = min(x1, dx - 1);
    for (int cy = max(y0, 0); cy <= min(y1, dy - 1); ++cy) {
      unsigned lo, hi;
      if (cy == y0 || cy == y1) {
        cols.row_range(xa, xb, cy, lo, hi);
        knn_scan<Cols, KMAX>(cols, q, lo, hi, k, bd, bi);
      } else {
        if (x0 >= 0) {
          cols.row_range(x0, x0, cy, lo, hi);
          knn_scan<Cols, KMAX>(cols, q, lo, hi, k, bd, bi);
        }
        if (x1 < dx && x1 != x0) {
          cols.row_range(x1, x1, cy, lo, hi);
          knn_scan<Cols, KMAX>(cols, q, lo, hi, k, bd, bi);
        }
      }
    }
    if (knn_settled<KMAX>(R, k, slack, cell_xy, bd)) return true;
  }
  return false;
}

template <int KMAX>
__global__ void __launch_bounds__(kBlock)
k_knn_tile(const float4* spt, const uint32_t* col_first, const int* fit_rank, const FrameInfo* fi, float inv_xy,
           float cell_xy, int K, int* knn_out)
{
  __shared__ TileShared<true> S;
  const FrameInfo& F = fi[blockIdx.y];
  const int dx = F.div_b[0], dy = F.div_b[1];
  if (F.n_fit <= 0 || (int)blockIdx.x >= ((dx + kTile - 1) / kTile) * ((dy + kTile - 1) / kTile)) return;
  TileGeom g;
  if (!load_tile<true>(F, (int)blockIdx.x, spt, col_first, fit_rank, S, g)) return;
  const int k = min(K, F.n_fit);
  const TileCols tc{S.pt, S.tab, S.fit, g.xh0, g.yh0};
  const GlobalCols gc{spt, col_first, fit_rank, (unsigned)F.col_off, dx};
  for (unsigned i = threadIdx.x; i < g.n_int; i += kBlock) {
    unsigned eg, el;
    interior_point(S, g, i, eg, el);
    const float4 q = spt[eg];
    const int me = fit_rank[__float_as_uint(q.w)];
    if (me < 0) continue;
    const int cqx = clampi(cell_unclamped(q.x, inv_xy, F.min_b[0]), 0, dx - 1);
    const int cqy = clampi(cell_unclamped(q.y, inv_xy, F.min_b[1]), 0, dy - 1);
    float bd[KMAX];
    int bi[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      bd[j] = INFINITY;
      bi[j] = -1;
    }
    const int max_ring = max(max(cqx, dx - 1 - cqx), max(cqy, dy - 1 - cqy));
    bool done = false;
    int next_ring = 0;
    if (g.in_lds) {
      // the whole halo square in one sweep: one contiguous run per row
      const int x0 = max(cqx - kHalo, 0), x1 = min(cqx + kHalo, dx - 1);
      for (int cy = max(cqy - kHalo, 0); cy <= min(cqy + kHalo, dy - 1); ++cy) {
        unsigned lo, hi;
        tc.row_range(x0, x1, cy, lo, hi);
        knn_scan<TileCols, KMAX>(tc, q, lo, hi, k, bd, bi);
      }
      done = knn_settled<KMAX>(kHalo, k, F.slack, cell_xy, bd);
      next_ring = kHalo + 1;
    }
    if (!done && next_ring <= max_ring)
      knn_rings<GlobalCols, KMAX>(gc, q, cqx, cqy, dx, dy, next_ring, max_ring, k, F.slack, cell_xy, bd, bi);
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
      if (j < K) knn_out[(size_t)me * K + j] = (j < k) ? bi[j] : -1;
  }
}

}  // namespace hope
