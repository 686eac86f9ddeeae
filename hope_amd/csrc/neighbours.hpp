// neighbours.hpp — A6 (normals) and A8 (k nearest neighbours) on the column index.
//
// The down-sampled cloud is sorted by (frame column, z cell); col_first[] gives the range of
// every xy column.  The search code (stencil scans in a z window, ordering network, sorted k-NN list, ring
// expansion) is written once over an accessor of that index, and two sets of kernels run it:
//   * k_normals_direct / k_knn_direct (the default): one thread per point in column order, the index read
//     straight from HBM through the caches (GlobalColsT) -- the lanes of a wave are neighbours, so their scans
//     share cache lines; k_knn_todo / k_knn_todo_threads finish the queries whose rings leave the fast range;
//   * k_normals_tile / k_knn_tile (HOPE_DEBUG_NEIGHBOURS=tiles; the first implementation, kept as a cross-check):
//     a workgroup owns a tile of columns, copies tile + halo into LDS with a local column table (TileCols) and
//     serves every query of the tile from there.
// Both accessors see the same points in the same arithmetic, so results are identical.
//
// Reference semantics: pcl::NormalEstimation with a radius search (utilities.cpp:428-446) and
// ZGrowing::findPointNeighbours' nearestKSearch (z_growing.cpp:196-226); see kernels.hpp for the
// eigen solver and the distance function.
#pragma once
#include "kernels.hpp"
#include "sortnet.hpp"

namespace hope {

constexpr int kTile = 16;                  // interior columns per tile side
constexpr int kSuper = 32;                 // a workgroup owns a 32x32-column super tile: taken whole when its points fit
                                           // in LDS (a depth frame's surfaces fill a fraction of their xy bounding box: a
                                           // 16x16 tile holds ~70 points, too few for 256 threads), else as its four tiles
constexpr int kSlabSide = 8;               // largest region loaded as a z slab
constexpr int kHaloNormals = 2;            // halo columns on each side: the normal radius reaches 2 columns
constexpr int kHaloKnn = 4;                // k-NN: rings 0..4 are served from LDS
constexpr int kTileCap = 1536;             // points (tile + halo) held in LDS: 24 KiB
constexpr int kRunDirect = 24;             // a stencil row with at most this many points is scanned whole

template <bool W_IS_RANK>
struct GlobalColsT {  // a column index straight from HBM: W_IS_RANK false = every down-sampled point
                      // (.w = PCL-order index), true = the coarse index of the fit points (.w = fit rank)
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
  __device__ __forceinline__ int fit(unsigned, const float4& p) const
  {
    return W_IS_RANK ? __float_as_int(p.w) : fit_rank[__float_as_uint(p.w)];
  }
  static constexpr bool kLds = false;
  static constexpr bool kNotes = false;  // A6 re-scans the stencil for every pick instead of noting the neighbours
  typedef unsigned short cand_t;
  __device__ __forceinline__ unsigned note_base() const { return 0u; }  // (notes hold position - note_base())
};
typedef GlobalColsT<false> GlobalCols;
typedef GlobalColsT<true> CoarseCols;
struct GlobalColsNotes : GlobalCols {  // ... or notes them as 32-bit positions in the sorted array (k_normals_direct)
  static constexpr bool kNotes = true;
  typedef unsigned cand_t;
};
// The column table straight from HBM, the points from an LDS copy of ONE contiguous stretch [w0, w0 + n) of the
// sorted array (k_normals_direct: the stretch that holds every stencil row of the workgroup's 256 consecutive points).
struct WindowCols {
  const float4* s_pt;
  unsigned w0;
  const uint32_t* col_first;
  unsigned col_off;
  int dx;
  __device__ __forceinline__ void range(int cx, int cy, unsigned& lo, unsigned& hi) const
  {
    const unsigned c = col_off + (unsigned)(cy * dx + cx);
    lo = col_first[c];
    hi = col_first[c + 1];
  }
  __device__ __forceinline__ void row_range(int cx0, int cx1, int cy, unsigned& lo, unsigned& hi) const
  {
    const unsigned c = col_off + (unsigned)(cy * dx);
    lo = col_first[c + cx0];
    hi = col_first[c + cx1 + 1];
  }
  __device__ __forceinline__ float4 pt(unsigned e) const { return s_pt[e - w0]; }
  __device__ __forceinline__ int fit(unsigned, const float4& p) const { return __float_as_int(p.w); }
  static constexpr bool kLds = false;
  static constexpr bool kNotes = true;
  typedef unsigned short cand_t;
  __device__ __forceinline__ unsigned note_base() const { return w0; }
};

template <int HALO>
struct TileCols {  // LDS copy of a tile and its halo
  static constexpr int kTabStride = kSuper + 2 * HALO + 1;  // per row: the column starts + the row end
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
  static constexpr bool kNotes = true;
  typedef unsigned short cand_t;
  __device__ __forceinline__ unsigned note_base() const { return 0u; }
};

template <bool WITH_FIT, int HALO>
struct TileShared {
  static constexpr int kTW = kSuper + 2 * HALO;
  static constexpr int kTabStride = kTW + 1;
  static constexpr int kSlabW = kSlabSide + 2 * HALO;
  float4 pt[kTileCap];  // WITH_FIT: .w holds the fit rank (int bits) instead of the PCL-order index
  unsigned tab[kTW * kTabStride];
  unsigned rowg[kTW + 1];   // global start of each halo row segment
  unsigned rowl[kTW + 1];   // local start (prefix of segment sizes)
  unsigned intg[kSuper + 1]; // global start of each interior row segment
  unsigned intp[kSuper + 1]; // prefix of interior row sizes
  unsigned colg[kSlabW * kSlabW]; // slab mode (regions of at most kSlabSide columns a side): global start of the slab part of each column
  unsigned zext[2];         // slab mode: ordered-int min / max z of the region
  unsigned total;
  unsigned nlist;           // slab mode: the interior points of the copy, compacted
  unsigned short list[kTileCap];
};

struct TileGeom {
  int x0, x1, y0, y1;      // interior columns [x0,x1) x [y0,y1)
  int xh0, xh1, yh0, yh1;  // with halo, clamped to the grid
  unsigned total;          // points in the LDS copy (tile + halo)
  unsigned n_int;          // points in the interior (all z)
  bool in_lds;             // the LDS copy is valid
  bool slab;               // the copy is a z slab: interior points also need z0 <= z < z1
  float z0, z1;
};

__device__ __forceinline__ void tile_extent(const FrameInfo& F, int x0, int y0, int size, int halo, TileGeom& g)
{
  const int dx = F.div_b[0], dy = F.div_b[1];
  g.x0 = x0;
  g.y0 = y0;
  g.x1 = min(x0 + size, dx);
  g.y1 = min(y0 + size, dy);
  g.xh0 = max(g.x0 - halo, 0);
  g.yh0 = max(g.y0 - halo, 0);
  g.xh1 = min(g.x1 + halo, dx);
  g.yh1 = min(g.y1 + halo, dy);
  g.slab = false;
  g.z0 = -INFINITY;
  g.z1 = INFINITY;
}

// Loads the size x size column tile at (x0, y0) of frame F with its halo, every z.  Returns
// false if the tile holds no point; g.in_lds tells whether the LDS copy was made (it fits).
template <bool WITH_FIT, int HALO>
__device__ __forceinline__ bool load_tile(const FrameInfo& F, int x0, int y0, int size, const float4* spt,
                                          const uint32_t* col_first, const int* fit_rank,
                                          TileShared<WITH_FIT, HALO>& S, TileGeom& g)
{
  constexpr int kTabStride = TileShared<WITH_FIT, HALO>::kTabStride;
  const int tid = threadIdx.x;
  const int dx = F.div_b[0];
  __syncthreads();  // the previous tile of this block is done with S
  tile_extent(F, x0, y0, size, HALO, g);
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
    for (int r = irows; r <= kSuper; ++r) S.intp[r] = acc;  // (padded: interior_point searches all kSuper rows)
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
  for (int r = tid / kWave; r < rows; r += kBlock / kWave) {  // a wave copies a row segment: coalesced, no search
    const unsigned gl = S.rowg[r], lo = S.rowl[r], n = S.rowl[r + 1] - lo;
    for (unsigned i = tid % kWave; i < n; i += kWave) {
      float4 p = spt[gl + i];
      if (WITH_FIT && fit_rank) p.w = __int_as_float(fit_rank[__float_as_uint(p.w)]);  // (null: .w is the rank already)
      S.pt[lo + i] = p;
    }
  }
  __syncthreads();
  return true;
}

// i-th interior point of the tile: position in the sorted array and in the LDS copy
template <bool WITH_FIT, int HALO>
__device__ __forceinline__ void interior_point(const TileShared<WITH_FIT, HALO>& S, const TileGeom& g, unsigned i,
                                               unsigned& e_glob, unsigned& e_loc)
{
  static_assert(kSuper == 32, "five-step search over the interior rows");
  int r = 0;  // last row whose first point is <= i (rows past the tile's last hold the total)
  if (S.intp[r + 16] <= i) r += 16;
  if (S.intp[r + 8] <= i) r += 8;
  if (S.intp[r + 4] <= i) r += 4;
  if (S.intp[r + 2] <= i) r += 2;
  if (S.intp[r + 1] <= i) r += 1;
  e_glob = S.intg[r] + (i - S.intp[r]);
  const int rh = r + (g.y0 - g.yh0);
  e_loc = S.rowl[rh] + (e_glob - S.rowg[rh]);
}

// first entry of [lo, hi) of the sorted array with z >= zt (a column ascends in z up to one cell)
__device__ __forceinline__ unsigned lower_bound_z_glob(const float4* spt, unsigned lo, unsigned hi, float zt)
{
  while (lo < hi) {
    const unsigned mid = (lo + hi) >> 1;
    if (spt[mid].z < zt) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// min / max z over the columns of the region with halo (first / last entry of every column)
template <bool WITH_FIT, int HALO>
__device__ __forceinline__ bool region_z_extent(const FrameInfo& F, const TileGeom& g, const float4* spt,
                                                const uint32_t* col_first, TileShared<WITH_FIT, HALO>& S, float& zmin,
                                                float& zmax, unsigned& count)
{
  const int tid = threadIdx.x;
  const int dx = F.div_b[0];
  const int rows = g.yh1 - g.yh0, width = g.xh1 - g.xh0;
  __syncthreads();
  if (tid == 0) {
    S.zext[0] = 0xffffffffu;
    S.zext[1] = 0u;
    S.total = 0u;
  }
  __syncthreads();
  unsigned lo_o = 0xffffffffu, hi_o = 0u, cnt = 0u;
  for (int c = tid; c < rows * width; c += kBlock) {
    const int r = c / width, cc = c - r * width;
    const unsigned col = (unsigned)F.col_off + (unsigned)((g.yh0 + r) * dx + g.xh0 + cc);
    const unsigned a = col_first[col], b = col_first[col + 1];
    if (b > a) {
      lo_o = min(lo_o, f2ord(spt[a].z));
      hi_o = max(hi_o, f2ord(spt[b - 1].z));
      cnt += b - a;
    }
  }
  if (cnt) {
    atomicMin(&S.zext[0], lo_o);
    atomicMax(&S.zext[1], hi_o);
    atomicAdd(&S.total, cnt);
  }
  __syncthreads();
  count = S.total;
  if (count == 0) return false;
  zmin = ord2f(S.zext[0]);
  zmax = ord2f(S.zext[1]);
  return true;
}

// Loads the z slab [zlo_h, zhi_h) of the region: every column is cut by two binary searches on
// z and the pieces are packed in (row, column) order, so that a stencil row is still one
// contiguous run and the column table works as in load_tile.  g.total > kTileCap: not loaded.
template <bool WITH_FIT, int HALO>
__device__ __forceinline__ void load_slab(const FrameInfo& F, float zlo_h, float zhi_h, const float4* spt,
                                          const uint32_t* col_first, const int* fit_rank,
                                          TileShared<WITH_FIT, HALO>& S, TileGeom& g)
{
  constexpr int kTabStride = TileShared<WITH_FIT, HALO>::kTabStride;
  const int tid = threadIdx.x;
  const int dx = F.div_b[0];
  const int rows = g.yh1 - g.yh0, width = g.xh1 - g.xh0;
  const int nc = rows * width;
  __syncthreads();
  for (int c = tid; c < nc; c += kBlock) {
    const int r = c / width, cc = c - r * width;
    const unsigned col = (unsigned)F.col_off + (unsigned)((g.yh0 + r) * dx + g.xh0 + cc);
    unsigned a = col_first[col], b = col_first[col + 1];
    if (b > a) {
      a = lower_bound_z_glob(spt, a, b, zlo_h);
      b = lower_bound_z_glob(spt, a, b, zhi_h);
    }
    S.colg[c] = a;
    S.tab[r * kTabStride + cc] = b - a;
  }
  __syncthreads();
  if (tid < kWave) {  // exclusive scan of the nc counts in (row, column) order by one wave
    const int per = (nc + kWave - 1) / kWave;
    const int c0 = tid * per, c1 = min(c0 + per, nc);
    unsigned sum = 0;
    for (int c = c0; c < c1; ++c) sum += S.tab[(c / width) * kTabStride + (c % width)];
    unsigned incl = sum;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const unsigned o = (unsigned)__shfl_up((int)incl, off);
      if (tid >= off) incl += o;
    }
    unsigned run = incl - sum;
    for (int c = c0; c < c1; ++c) {
      unsigned* t = &S.tab[(c / width) * kTabStride + (c % width)];
      const unsigned n = *t;
      *t = run;
      run += n;
    }
    if (tid == kWave - 1) S.total = incl;
  }
  __syncthreads();
  g.total = S.total;
  g.in_lds = g.total <= (unsigned)kTileCap;
  g.slab = true;
  g.n_int = g.total;
  for (int r = tid; r < rows; r += kBlock)
    S.tab[r * kTabStride + width] = (r + 1 < rows) ? S.tab[(r + 1) * kTabStride] : g.total;
  __syncthreads();
  if (!g.in_lds || g.total == 0) return;
  for (unsigned i = tid; i < g.total; i += kBlock) {
    // row: last one whose start is <= i; column likewise inside the row
    int r = 0;
    while (r + 1 < rows && S.tab[(r + 1) * kTabStride] <= i) ++r;
    const unsigned* t = &S.tab[r * kTabStride];
    int cc = 0;
    while (cc + 1 < width && t[cc + 1] <= i) ++cc;
    float4 p = spt[S.colg[r * width + cc] + (i - t[cc])];
    if (WITH_FIT && fit_rank) p.w = __int_as_float(fit_rank[__float_as_uint(p.w)]);  // (null: .w is the rank already)
    S.pt[i] = p;
  }
  __syncthreads();
}

// Walks the regions of one 32x32 super tile that fit in LDS, one per call of next():
//   * the whole super tile with its halo, every z, else each of its four 16x16 tiles:
//   * the whole tile with its halo, every z (horizontal surfaces: a few points per column), else
//   * its 8x8 quarters, every z (noisy far range: several z layers per column), and a quarter
//     that still does not fit
//   * in z slabs (walls, clutter): interior [z, z + h cells), halo zhalo on both sides; a slab
//     that does not fit halves h; at h = 1 the quarter is split into 4x4 regions, whose one-cell
//     slabs always fit (a cell holds one down-sampled point).
// Written as a state machine so that the caller's per-region body exists once in the kernel.
// Every decision is block-uniform.
template <bool WITH_FIT, int HALO, int SIDE>
struct TileWalker {
  const FrameInfo& F;
  const float4* spt;
  const uint32_t* col_first;
  const int* fit_rank;
  TileShared<WITH_FIT, HALO>& S;
  int halo_xy;   // xy halo of the slab regions (<= HALO)
  float zhalo, leaf_z;
  static constexpr int side = SIDE;  // kSuper: the block owns a super tile; kTile: one 16x16 tile (dense clouds; the host picks)
  int SX, SY;    // the block's tile
  int X0 = 0, Y0 = 0;  // the current 16x16 tile
  int stage = 0, a = 0, b = 0, sub = 0;
  bool in8 = false, in4 = false, in4walk = false, in16 = false;
  int x8 = 0, y8 = 0, x4 = 0, y4 = 0, h8 = 1, h4 = 1;
  float z8 = 0.f, zend8 = 0.f, zs8 = 0.f, ze8 = 0.f, z4 = 0.f, zend4 = 0.f;

  __device__ __forceinline__ TileWalker(const FrameInfo& F_, int tile, int halo_xy_, float zhalo_,
                                        float leaf_z_, const float4* spt_, const uint32_t* col_first_,
                                        const int* fit_rank_, TileShared<WITH_FIT, HALO>& S_)
      : F(F_), spt(spt_), col_first(col_first_), fit_rank(fit_rank_), S(S_), halo_xy(halo_xy_), zhalo(zhalo_),
        leaf_z(leaf_z_)
  {
    static_assert(SIDE == kSuper || SIDE == kTile, "tile side");
    const int ntx = (F.div_b[0] + side - 1) / side;
    const int ty = tile / ntx;
    SX = (tile - ty * ntx) * side;
    SY = ty * side;
    if (side != kSuper) stage = 1;  // (no super tile to try)
  }

  // slab height (cells) from the mean number of points per z layer of the region
  __device__ __forceinline__ int slab_cells(unsigned count, float zmin, float zmax) const
  {
    const float layers = fmaxf((zmax - zmin) / leaf_z + 1.f, 1.f);
    const float per_layer = fmaxf((float)count / layers, 1e-3f);
    return (int)fminf(fmaxf(0.6f * (float)kTileCap / per_layer - 2.f * zhalo / leaf_z, 1.f), 1.0e9f);
  }

  // after a slab that fitted: double the height for the next one if this one was sparse, or if it is
  // so thin next to its z halo (a z grid far finer than the search radius) that doubling adds next to nothing
  __device__ __forceinline__ bool grow(unsigned total, int h) const
  {
    return total * 4u <= (unsigned)kTileCap || (float)h * leaf_z * 8.f < zhalo;
  }

  __device__ __forceinline__ bool next(TileGeom& g)
  {
    for (int guard = 0; guard < (1 << 22); ++guard) {
      if (stage == 0) {
        stage = 1;
        if (!load_tile<WITH_FIT, HALO>(F, SX, SY, kSuper, spt, col_first, fit_rank, S, g)) break;
        if (g.in_lds) {
          stage = 2;
          return true;
        }
        continue;
      }
      if (stage == 2) break;
      if (!in16) {
        // the next 16x16 tile of the super tile (or the block's only one)
        if (sub == (side == kSuper ? 4 : 1)) break;
        X0 = SX + (sub & 1) * kTile;
        Y0 = SY + (sub >> 1) * kTile;
        ++sub;
        if (X0 >= F.div_b[0] || Y0 >= F.div_b[1]) continue;
        if (!load_tile<WITH_FIT, HALO>(F, X0, Y0, kTile, spt, col_first, fit_rank, S, g)) continue;
        if (g.in_lds) return true;
        in16 = true;  // it does not fit: its quarters
        a = 0;
        continue;
      }
      if (in4) {
        if (!in4walk) {
          if (b == 4) {
            in4 = false;
            z8 = ze8;
            continue;
          }
          x4 = x8 + (b & 1) * 4;
          y4 = y8 + (b >> 1) * 4;
          ++b;
          if (x4 >= F.div_b[0] || y4 >= F.div_b[1]) continue;
          tile_extent(F, x4, y4, 4, halo_xy, g);
          float zmin, zmax;
          unsigned count;
          if (!region_z_extent<WITH_FIT, HALO>(F, g, spt, col_first, S, zmin, zmax, count)) continue;
          z4 = fmaxf(zs8, zmin - 1.5f * leaf_z);
          zend4 = fminf(ze8, zmax + 1.5f * leaf_z);
          h4 = slab_cells(count, zmin, zmax);
          in4walk = true;
        }
        if (!(z4 < zend4)) {
          in4walk = false;
          continue;
        }
        const float z1 = fminf(z4 + (float)h4 * leaf_z, zend4);
        tile_extent(F, x4, y4, 4, halo_xy, g);
        load_slab<WITH_FIT, HALO>(F, z4 - zhalo, z1 + zhalo, spt, col_first, fit_rank, S, g);
        if (!g.in_lds && h4 > 1) {
          h4 = max(h4 / 2, 1);
          continue;
        }
        // (a one-cell slab that does not fit -- raw, not down-sampled, points -- is served from the global
        // arrays, which have no capacity: then the rest of the z range goes in the same step)
        const float z1u = g.in_lds ? z1 : zend4;
        g.z0 = z4;
        g.z1 = z1u;
        z4 = z1u;
        if (g.in_lds && grow(g.total, h4)) h4 = min(h4 * 2, 1 << 30);
        if (g.in_lds && g.total == 0) continue;
        return true;  // (in_lds false: the caller reads from the global arrays)
      }
      if (in8) {
        if (!(z8 < zend8)) {
          in8 = false;
          continue;
        }
        const float z1 = fminf(z8 + (float)h8 * leaf_z, zend8);
        tile_extent(F, x8, y8, 8, halo_xy, g);
        load_slab<WITH_FIT, HALO>(F, z8 - zhalo, z1 + zhalo, spt, col_first, fit_rank, S, g);
        if (!g.in_lds) {
          if (h8 > 1) {
            h8 = max(h8 / 2, 1);
            continue;
          }
          // the 4x4 regions take over the rest of the quarter's z range, each with a slab height of its own
          // (handing back one cell at a time would crawl through a dense stretch of a fine z grid)
          in4 = true;
          in4walk = false;
          b = 0;
          zs8 = z8;
          ze8 = zend8;
          continue;
        }
        g.z0 = z8;
        g.z1 = z1;
        z8 = z1;
        if (grow(g.total, h8)) h8 = min(h8 * 2, 1 << 30);
        if (g.total == 0) continue;
        return true;
      }
      if (a == 4) {
        in16 = false;
        continue;
      }
      x8 = X0 + (a & 1) * 8;
      y8 = Y0 + (a >> 1) * 8;
      ++a;
      if (x8 >= F.div_b[0] || y8 >= F.div_b[1]) continue;
      if (!load_tile<WITH_FIT, HALO>(F, x8, y8, 8, spt, col_first, fit_rank, S, g)) continue;
      if (g.in_lds) return true;
      tile_extent(F, x8, y8, 8, halo_xy, g);
      float zmin, zmax;
      unsigned count;
      if (!region_z_extent<WITH_FIT, HALO>(F, g, spt, col_first, S, zmin, zmax, count)) continue;
      z8 = zmin - 1.5f * leaf_z;
      zend8 = zmax + 1.5f * leaf_z;
      h8 = slab_cells(count, zmin, zmax);
      in8 = true;
    }
    stage = 2;
    return false;
  }
};

// Calls fn_lds(q, cqx, cqy) for every interior point of the tile / slab, q taken from the LDS
// copy.  A slab that could not be loaded (cannot happen with one point per cell; handled exactly
// anyway) calls fn_glob instead, with q from the sorted array (.w = PCL-order index).
template <bool WITH_FIT, int HALO, class FnLds, class FnGlob>
__device__ __forceinline__ void for_each_interior(const FrameInfo& F, const TileGeom& g,
                                                  TileShared<WITH_FIT, HALO>& S, const float4* spt,
                                                  const uint32_t* col_first, float inv_xy, float leaf_z,
                                                  FnLds&& fn_lds, FnGlob&& fn_glob)
{
  const int dx = F.div_b[0], dy = F.div_b[1];
  if (g.in_lds && g.slab) {
    // a slab copy is mostly halo (xy and z): list its interior points first, so that the queries
    // run on full waves (order is irrelevant: every query is independent)
    if (threadIdx.x == 0) S.nlist = 0u;
    __syncthreads();
    for (unsigned i = threadIdx.x; i < g.total; i += kBlock) {
      const float4 q = S.pt[i];
      const int cqx = clampi(cell_unclamped(q.x, inv_xy, F.min_b[0]), 0, dx - 1);
      const int cqy = clampi(cell_unclamped(q.y, inv_xy, F.min_b[1]), 0, dy - 1);
      if (cqx >= g.x0 && cqx < g.x1 && cqy >= g.y0 && cqy < g.y1 && q.z >= g.z0 && q.z < g.z1)
        S.list[atomicAdd(&S.nlist, 1u)] = (unsigned short)i;
    }
    __syncthreads();
    const unsigned n_list = S.nlist;
    for (unsigned k = threadIdx.x; k < n_list; k += kBlock) {
      const float4 q = S.pt[S.list[k]];
      const int cqx = clampi(cell_unclamped(q.x, inv_xy, F.min_b[0]), 0, dx - 1);
      const int cqy = clampi(cell_unclamped(q.y, inv_xy, F.min_b[1]), 0, dy - 1);
      fn_lds(q, cqx, cqy);
    }
  } else if (g.in_lds) {
    for (unsigned i = threadIdx.x; i < g.n_int; i += kBlock) {
      unsigned eg, el;
      interior_point(S, g, i, eg, el);
      const float4 q = S.pt[el];
      const int cqx = clampi(cell_unclamped(q.x, inv_xy, F.min_b[0]), 0, dx - 1);
      const int cqy = clampi(cell_unclamped(q.y, inv_xy, F.min_b[1]), 0, dy - 1);
      fn_lds(q, cqx, cqy);
    }
  } else {
    // (several threads share a column when the region has fewer columns than the block has threads)
    const int w = g.x1 - g.x0, ncol = w * (g.y1 - g.y0);
    const int share = ncol < kBlock ? kBlock / max(ncol, 1) : 1;
    for (int t = threadIdx.x; t < ncol * share; t += kBlock) {
      const int c = t / share;
      const int cx = g.x0 + c % w, cy = g.y0 + c / w;
      const unsigned col = (unsigned)F.col_off + (unsigned)(cy * dx + cx);
      unsigned a = col_first[col];
      const unsigned b = col_first[col + 1];
      a = lower_bound_z_glob(spt, a, b, g.z0 - 2.5f * leaf_z);
      for (unsigned e = a + (unsigned)(t % share); e < b; e += (unsigned)share) {
        const float4 q = spt[e];
        if (q.z >= g.z1 + 2.5f * leaf_z) break;
        if (q.z >= g.z0 && q.z < g.z1) fn_glob(q, cx, cy);
      }
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

// Entries are fetched four at a time (clamped indices, so the four reads are independent and in
// flight together: a scan is bound by the latency of its LDS / L2 reads, not by their number) and
// handed to fn in ascending order.
template <class Cols, class Fn>
__device__ __forceinline__ void for_run(const Cols& cols, unsigned lo, unsigned hi, Fn&& fn)
{
  for (unsigned e = lo; e < hi; e += 4u) {
    const unsigned last = hi - 1u;
    const float4 p0 = cols.pt(e), p1 = cols.pt(min(e + 1u, last)), p2 = cols.pt(min(e + 2u, last)),
                 p3 = cols.pt(min(e + 3u, last));
    fn(e, p0);
    if (e + 1u < hi) fn(e + 1u, p1);
    if (e + 2u < hi) fn(e + 2u, p2);
    if (e + 3u < hi) fn(e + 3u, p3);
  }
}

template <class Cols, class Fn>
__device__ __forceinline__ void for_window(const Cols& cols, int x0, int x1, int cy, float zlo, float zhi, Fn&& fn)
{
  unsigned lo, hi;
  cols.row_range(x0, x1, cy, lo, hi);
  if (hi - lo <= (unsigned)kRunDirect) {
    for_run(cols, lo, hi, fn);
    return;
  }
  for (int cx = x0; cx <= x1; ++cx) {
    cols.range(cx, cy, lo, hi);
    if (hi - lo > 8u) lo = lower_bound_z(cols, lo, hi, zlo);
    for (unsigned e = lo; e < hi; e += 4u) {
      const unsigned last = hi - 1u;
      const float4 p0 = cols.pt(e), p1 = cols.pt(min(e + 1u, last)), p2 = cols.pt(min(e + 2u, last)),
                   p3 = cols.pt(min(e + 3u, last));
      if (p0.z > zhi) break;
      fn(e, p0);
      if (e + 1u >= hi || p1.z > zhi) break;
      fn(e + 1u, p1);
      if (e + 2u >= hi || p2.z > zhi) break;
      fn(e + 2u, p2);
      if (e + 3u >= hi || p3.z > zhi) break;
      fn(e + 3u, p3);
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
__device__ __forceinline__ void accumulate_sorted(const Cols& cols, const float4 q, const typename Cols::cand_t* cand_e,
                                                  int total, float* accu)
{
  static_assert(N == 8 || N == 12 || N == 16 || N == 32, "sorting networks exist for 8, 12, 16 and 32 inputs");
  float cd[N];
  unsigned ce[N];
#pragma unroll
  for (int s = 0; s < N; ++s) ce[s] = (s < total ? (unsigned)cand_e[s * kBlock] : 0u) + cols.note_base();
#pragma unroll
  for (int s = 0; s < N; ++s) {
    const float4 p = cols.pt(ce[s]);
    cd[s] = s < total ? dist2(q.x, q.y, q.z, p.x, p.y, p.z) : INFINITY;
  }
  if (N == 8) {
    HOPE_SORT_NET8
  } else if (N == 12) {
    HOPE_SORT_NET12
  } else if (N == 16) {
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

// the normal of a point from the nine sums over its neighbours (computeMeanAndCovarianceMatrix's tail, pcl::eigen33's
// smallest eigenvector, curvature, flipNormalTowardsViewpoint towards the origin: PCL 1.8.1 centroid.hpp, normal_3d.h)
__device__ __forceinline__ float4 normal_from_sums(float* accu, int total, const float4 q)
{
  const float cnt = (float)total;
#pragma unroll
  for (int k = 0; k < 9; ++k) accu[k] /= cnt;
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

// columns / rows of the stencil that can hold a point within `radius` of q (see estimate_normal)
__device__ __forceinline__ void stencil_reach(const float4 q, int ring, float radius, float inv_xy, float slack, int& kl,
                                              int& kr, int& kd, int& ku)
{
  kl = kr = kd = ku = ring;
  const float cell = 1.0f / inv_xy, lim = radius * 1.00001f;
  const float sx = q.x * inv_xy, sy = q.y * inv_xy;
  const float u = sx - floorf(sx), v = sy - floorf(sy);
  while (kl > 1 && (u + (float)(kl - 1) - slack) * cell * 0.999999f > lim) --kl;
  while (kr > 1 && ((float)kr - u - slack) * cell * 0.999999f > lim) --kr;
  while (kd > 1 && (v + (float)(kd - 1) - slack) * cell * 0.999999f > lim) --kd;
  while (ku > 1 && ((float)ku - v - slack) * cell * 0.999999f > lim) --ku;
}

template <class Cols, int CAPL>
__device__ __forceinline__ float4 estimate_normal(const Cols& cols, const float4 q, int cqx, int cqy, int dx, int dy,
                                                  int ring, float radius, float r2, float leaf_z,
                                                  typename Cols::cand_t* cand_e, int ablate, float inv_xy, float slack)
{
  static_assert(CAPL <= kNbCap, "the largest sorting network has kNbCap inputs");
  // The stencil: `ring` columns to every side reach the radius from ANY place in the query's cell; from where the
  // query actually is, the outermost ones mostly cannot (radius 1.01 cells: the second column to the left only
  // within 0.01 cells of the left edge).  A point k columns to the left is at least (u + k - 1 - slack) cells away
  // in x, u the query's place in its cell, slack the float-rounding slack of a column index (FrameInfo::slack);
  // k columns to the right: (k - u - slack).  The factors leave 1e-5 to the roundings of the distance itself.
  int kl, kr, kd, ku;
  stencil_reach(q, ring, radius, inv_xy, slack, kl, kr, kd, ku);
  const int x0 = max(cqx - kl, 0), x1 = min(cqx + kr, dx - 1);
  const int y0 = max(cqy - kd, 0), y1 = min(cqy + ku, dy - 1);
  const float zlo = q.z - radius - 1.5f * leaf_z, zhi = q.z + radius + 1.5f * leaf_z;
  int total = 0;
  auto note = [&](unsigned e, const float4& p) {
    if (dist2(q.x, q.y, q.z, p.x, p.y, p.z) < r2) {
      if (Cols::kNotes && total < CAPL) cand_e[total * kBlock] = (typename Cols::cand_t)(e - cols.note_base());
      ++total;
    }
  };
  for (int cy = y0; cy <= y1; ++cy) for_window(cols, x0, x1, cy, zlo, zhi, note);
  if (total < 3) return make_float4(NAN, NAN, NAN, NAN);
  if (ablate == 2) return make_float4((float)total, 0.f, 0.f, 0.f);
  float accu[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (Cols::kNotes && total <= CAPL) {
    // wave-uniform choice of the network: the smallest one that holds the longest list of the wave (a horizontal
    // surface finds 5..10 neighbours)
    if (CAPL <= 8 || !__any(total > 8))
      accumulate_sorted<Cols, 8>(cols, q, cand_e, total, accu);
    else if (CAPL <= 12 || !__any(total > 12))
      accumulate_sorted<Cols, 12>(cols, q, cand_e, total, accu);
    else if (CAPL <= 16 || !__any(total > 16))
      accumulate_sorted<Cols, 16>(cols, q, cand_e, total, accu);
    else
      accumulate_sorted<Cols, 32>(cols, q, cand_e, total, accu);
  } else {
    accumulate_by_rescan(cols, q, x0, x1, y0, y1, zlo, zhi, r2, total, accu);
  }
  if (ablate == 3) {
    const float cnt = (float)total;
#pragma unroll
    for (int k = 0; k < 9; ++k) accu[k] /= cnt;
    return make_float4(accu[0] + accu[1] + accu[2], accu[3] + accu[4] + accu[5], accu[6] + accu[7], accu[8]);
  }
  return normal_from_sums(accu, total, q);
}

// grid = (max tiles per frame, n_frames)
template <int CAPL, int SIDE>
__global__ void __launch_bounds__(kBlock, 3)
k_normals_tile(const float4* spt, const uint32_t* col_first, const FrameInfo* fi, float inv_xy, float inv_z,
               float radius, float r2, int ring, unsigned own_lo, unsigned own_hi, float4* normals, int ablate)
{
  constexpr int side = SIDE;
  __shared__ TileShared<false, kHaloNormals> S;
  __shared__ unsigned short s_cand_e[CAPL * kBlock];
  const FrameInfo& F = fi[blockIdx.y];
  const int dx = F.div_b[0], dy = F.div_b[1];
  if (F.n_vox <= 0 || (int)blockIdx.x >= ((dx + side - 1) / side) * ((dy + side - 1) / side)) return;
  const float leaf_z = 1.0f / inv_z;
  unsigned short* cand_e = s_cand_e + threadIdx.x;
  const GlobalCols gc{spt, col_first, nullptr, (unsigned)F.col_off, dx};
  const float zhalo = radius + 2.5f * leaf_z;
  TileWalker<false, kHaloNormals, SIDE> walk(F, (int)blockIdx.x, kHaloNormals, zhalo, leaf_z, spt, col_first, nullptr, S);
  TileGeom g;
  while (walk.next(g)) {
    if (ablate == 1) continue;
    const TileCols<kHaloNormals> tc{S.pt, S.tab, g.xh0, g.yh0};
    for_each_interior(
        F, g, S, spt, col_first, inv_xy, leaf_z,
        [&](const float4& q, int cqx, int cqy) {
          const unsigned idx = __float_as_uint(q.w);
          if (idx < own_lo || idx >= own_hi) return;  // (staged multi-GPU call: another rank's point)
          normals[idx] = estimate_normal<TileCols<kHaloNormals>, CAPL>(tc, q, cqx, cqy, dx, dy, ring, radius, r2, leaf_z,
                                                                       cand_e, ablate, inv_xy, F.slack);
        },
        [&](const float4& q, int cqx, int cqy) {
          const unsigned idx = __float_as_uint(q.w);
          if (idx < own_lo || idx >= own_hi) return;
          normals[idx] =
              estimate_normal<GlobalCols, CAPL>(gc, q, cqx, cqy, dx, dy, ring, radius, r2, leaf_z, cand_e, ablate, inv_xy,
                                                F.slack);
        });
  }
}

// A6 without tiles, for ONE cloud whose columns are tall (walls, shelves, clutter: most of a tile copy would be
// halo in z): one thread per point in column order, the stencil read straight from the sorted array.  The lanes of
// a wave are neighbours in a column or in adjacent columns, so their scans read the same cache lines.
// (4 waves per SIMD at 128 VGPRs and 44 bytes of spills beat 3 at 135 and none: 1.29 against 1.52 ms on the frame
// batch; 5 at 96 VGPRs lose again)
//
// The sorted array is row-major in (y, x), so everything the 256 consecutive points of a workgroup can find lies in
// ONE stretch of it: from the first stencil row of the first point to the last stencil row of the last one -- the
// workgroup's own points and 2 * ring grid rows.  When that stretch fits kWinCap points (depth frames: rows of
// ~100 points) it is copied to LDS once and every scan and re-fetch reads it there (WindowCols): the kernel is
// bound by the 64 B/clk of the L1 otherwise, at 16 lines per wave-level float4 load.  Tall columns (walls,
// clutter) do not fit: straight from HBM then, block by block.
constexpr int kWinCap = 1024;  // 16 KiB of points + 16 KiB of 16-bit notes = the 32 KiB of 32-bit notes of the other path

template <int CAPL>
__global__ void __launch_bounds__(kBlock, 4)
k_normals_direct(const float4* spt, const uint32_t* col_first, const FrameInfo* fi, float inv_xy, float inv_z,
                 float radius, float r2, int ring, unsigned own_lo, unsigned own_hi, unsigned n, const int* vox_frame,
                 float4* normals, int ablate, float th_norm, uint32_t* fit_flag, uint32_t* fit_flag_sorted)
{
  // fit_flag != NULL (a call that computes every normal): the A7 gate |nz| > th_norm is written here too, in index
  // order and in sorted order, n + 1 entries each -- the inputs of the two compactions, which then need no pass of
  // their own over the normals
  if (fit_flag && blockIdx.x == 0 && threadIdx.x == 0) {
    fit_flag[n] = 0u;
    fit_flag_sorted[n] = 0u;
  }
  constexpr int kNoteBytes = CAPL * kBlock * 4;
  constexpr int kWinBytes = kWinCap * 16 + CAPL * kBlock * 2;
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[kNoteBytes > kWinBytes ? kNoteBytes : kWinBytes];
  __shared__ unsigned s_w[2];
  const unsigned e = xcd_block() * kBlock + threadIdx.x;
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned idx = 0u;
  bool valid = e < n;
  if (valid) {
    q = spt[e];
    idx = __float_as_uint(q.w);
    valid = idx >= own_lo && idx < own_hi;  // (staged multi-GPU call: another rank's point)
  }
  const FrameInfo& F = fi[(valid && vox_frame) ? vox_frame[idx] : 0];
  const int dx = F.div_b[0], dy = F.div_b[1];
  const int cqx = clampi(cell_unclamped(q.x, inv_xy, F.min_b[0]), 0, dx - 1);
  const int cqy = clampi(cell_unclamped(q.y, inv_xy, F.min_b[1]), 0, dy - 1);
  GlobalColsNotes gc;
  gc.spt = spt;
  gc.col_first = col_first;
  gc.fit_rank = nullptr;
  gc.col_off = (unsigned)F.col_off;
  gc.dx = dx;
  // the stretch of the sorted array that holds every stencil row of this workgroup
  if (threadIdx.x == 0) {
    s_w[0] = 0xffffffffu;
    s_w[1] = 0u;
  }
  __syncthreads();
  unsigned lo = 0xffffffffu, hi = 0u;
  if (valid) {
    int kl, kr, kd, ku;
    stencil_reach(q, ring, radius, inv_xy, F.slack, kl, kr, kd, ku);
    const int x0 = max(cqx - kl, 0), x1 = min(cqx + kr, dx - 1);
    unsigned t;
    gc.row_range(x0, x1, max(cqy - kd, 0), lo, t);
    gc.row_range(x0, x1, min(cqy + ku, dy - 1), t, hi);
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    lo = min(lo, (unsigned)__shfl_xor((int)lo, off));
    hi = max(hi, (unsigned)__shfl_xor((int)hi, off));
  }
  if ((threadIdx.x & (kWave - 1)) == 0 && hi > 0u) {
    atomicMin(&s_w[0], lo);
    atomicMax(&s_w[1], hi);
  }
  __syncthreads();
  const unsigned w0 = s_w[0], w1 = s_w[1];
  float4 nrm;
  if (w1 > w0 && w1 - w0 <= (unsigned)kWinCap) {  // (block-uniform)
    float4* s_pt = reinterpret_cast<float4*>(s_raw);
    unsigned short* s_note = reinterpret_cast<unsigned short*>(s_raw + kWinCap * 16);
    for (unsigned i = threadIdx.x; i < w1 - w0; i += kBlock) s_pt[i] = spt[w0 + i];
    __syncthreads();
    if (!valid) return;
    const WindowCols wc{s_pt, w0, col_first, (unsigned)F.col_off, dx};
    nrm = estimate_normal<WindowCols, CAPL>(wc, q, cqx, cqy, dx, dy, ring, radius, r2, 1.0f / inv_z,
                                            s_note + threadIdx.x, ablate, inv_xy, F.slack);
  } else {
    if (!valid) return;
    nrm = estimate_normal<GlobalColsNotes, CAPL>(gc, q, cqx, cqy, dx, dy, ring, radius, r2, 1.0f / inv_z,
                                                 reinterpret_cast<unsigned*>(s_raw) + threadIdx.x, ablate, inv_xy,
                                                 F.slack);
  }
  normals[idx] = nrm;
  if (fit_flag) {
    const uint32_t f = fabsf(nrm.z) > th_norm ? 1u : 0u;  // NaN fails (k_fit_flags)
    fit_flag[idx] = f;
    fit_flag_sorted[e] = f;
  }
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

constexpr uint32_t kTodoFromRing0 = 0x80000000u;  // todo flag: only part of rings 0..2 has been seen

// Phase A, grid = (max tiles per frame, n_frames): every query whose k neighbours provably lie
// inside the LDS copy of its tile is answered here; the others (sparse neighbourhoods) are
// appended to `todo` and answered by k_knn_todo.
template <int KMAX, int SIDE>
__global__ void __launch_bounds__(kBlock, KMAX <= 8 ? 4 : 1)
k_knn_tile(const float4* spt, const uint32_t* col_first, const int* fit_rank, const FrameInfo* fi, float inv_xy,
           float inv_z, float cell_xy, int K, const uint32_t* fit_idx, unsigned own_lo, unsigned own_hi, int* knn_out,
           uint32_t* todo, unsigned* n_todo, int ablate)
{
  constexpr int side = SIDE;
  typedef TileCols<kHaloKnn> Tc;
  __shared__ TileShared<true, kHaloKnn> S;
  const FrameInfo& F = fi[blockIdx.y];
  const int dx = F.div_b[0], dy = F.div_b[1];
  if (F.n_fit <= 0 || (int)blockIdx.x >= ((dx + side - 1) / side) * ((dy + side - 1) / side)) return;
  const float leaf_z = 1.0f / inv_z;
  const int k = min(K, F.n_fit);
  // Fast pass: the 5x5 square is ring 2; a list that settles there holds only neighbours closer
  // than reach2, so a z window of that height loses nothing.
  const float reach2 = ring_reach(2, F.slack, cell_xy);
  const float zwin = reach2 + 1.5f * leaf_z;
  TileWalker<true, kHaloKnn, SIDE> walk(F, (int)blockIdx.x, 2, zwin + leaf_z, leaf_z, spt, col_first, fit_rank, S);
  TileGeom g;
  while (walk.next(g)) {
    const Tc tc{S.pt, S.tab, g.xh0, g.yh0};
    if (ablate == 1) continue;
    auto leftover = [&](const float4& q, int, int) {
      const int me = fit_rank ? fit_rank[__float_as_uint(q.w)] : __float_as_int(q.w);
      if (me < 0) return;
      if (fit_idx[me] < own_lo || fit_idx[me] >= own_hi) return;
      for (int j = 0; j < K; ++j) knn_out[(size_t)me * K + j] = -1;
      todo[atomicAdd(n_todo, 1u)] = (uint32_t)me | kTodoFromRing0;
    };
    for_each_interior(F, g, S, spt, col_first, inv_xy, leaf_z, [&](const float4& q, int cqx, int cqy) {
      const int me = __float_as_int(q.w);
      if (me < 0) return;
      if (own_hi - own_lo != 0xffffffffu) {  // staged multi-GPU call: only this rank's share of the queries
        const unsigned idx = fit_idx[me];
        if (idx < own_lo || idx >= own_hi) return;
      }
      knn_key lst[KMAX];
#pragma unroll
      for (int j = 0; j < KMAX; ++j) lst[j] = kKnnEmpty;
      const int max_ring = max(max(cqx, dx - 1 - cqx), max(cqy, dy - 1 - cqy));
      bool done = false;
      if (reach2 > 0.f) {
        const int x0 = max(cqx - 2, 0), x1 = min(cqx + 2, dx - 1);
        const float zlo = q.z - zwin, zhi = q.z + zwin;
        // nearest columns first: the 3x3 core fills the list with near points, after which most of
        // the outer ring is turned away by the single compare at the top of knn_offer
        auto offer = [&](unsigned e, const float4& p) { knn_offer<Tc, KMAX>(tc, q, e, p, k, lst); };
        const int xc0 = max(cqx - 1, 0), xc1 = min(cqx + 1, dx - 1);
#pragma unroll 1
        for (int s = 0; s < 3; ++s) {
          const int cy = cqy + (s == 0 ? 0 : (s == 1 ? -1 : 1));
          if (cy < 0 || cy >= dy) continue;
          for_window(tc, xc0, xc1, cy, zlo, zhi, offer);
        }
#pragma unroll 1
        for (int s = 0; s < 5; ++s) {
          const int cy = cqy + s - 2;
          if (cy < 0 || cy >= dy) continue;
          if (s == 0 || s == 4) {
            for_window(tc, x0, x1, cy, zlo, zhi, offer);
          } else {
            if (cqx - 2 >= 0) for_window(tc, cqx - 2, cqx - 2, cy, zlo, zhi, offer);
            if (cqx + 2 < dx) for_window(tc, cqx + 2, cqx + 2, cy, zlo, zhi, offer);
          }
        }
        done = knn_kth<KMAX>(k, lst) < reach2 * reach2 * 0.999999f;
      }
      if (ablate == 2) done = true;
      if (!done && !g.slab) {
        // sparse neighbourhood: the exact ring expansion over the LDS copy (every z; points the
        // fast pass has already listed are skipped)
        done = knn_rings<Tc, KMAX, true>(tc, q, cqx, cqy, dx, dy, 0, min(kHaloKnn, max_ring), k, F.slack, cell_xy,
                                         leaf_z, lst);
        if (max_ring <= kHaloKnn) done = true;  // every column of the frame has been visited
      }
      // settled: the answer; not settled: the list so far, which phase B takes up again
#pragma unroll
      for (int j = 0; j < KMAX; ++j)
        if (j < K) knn_out[(size_t)me * K + j] = (j < k) ? (int)(unsigned)lst[j] : -1;
      if (!done) todo[atomicAdd(n_todo, 1u)] = (uint32_t)me | (g.slab ? kTodoFromRing0 : 0u);
    }, leftover);
  }
}

// Phase A without tiles: one thread per fit point in column order, the columns read straight from the column index
// of the fit points (.w = fit rank).  Same fast pass, same ring expansion up to kHaloKnn, same leftovers as
// k_knn_tile; the lanes of a wave are neighbours on a surface, so their scans share cache lines.
// the fast pass and the ring expansion of one query; FULL: k == KMAX, known at compile time (the list code then has
// no selects on k)
template <int KMAX, bool FULL, class Fast>
__device__ __forceinline__ bool knn_direct_query(const Fast& fc, const CoarseCols& tc, const float4 q, int cqx, int cqy,
                                                 int dx, int dy, int k_any, float slack, float cell_xy, float leaf_z,
                                                 knn_key (&lst)[KMAX])
{
  const int k = FULL ? KMAX : k_any;
  const float reach2 = ring_reach(2, slack, cell_xy);
  const float zwin = reach2 + 1.5f * leaf_z;
  const int max_ring = max(max(cqx, dx - 1 - cqx), max(cqy, dy - 1 - cqy));
  bool done = false;
  if (reach2 > 0.f) {
    const int x0 = max(cqx - 2, 0), x1 = min(cqx + 2, dx - 1);
    const float zlo = q.z - zwin, zhi = q.z + zwin;
    auto offer = [&](unsigned e, const float4& p) { knn_offer<Fast, KMAX>(fc, q, e, p, k, lst); };
    // whole rows of five columns, nearest row first: five scans of 5..7 points each (the tile kernel's order --
    // 3x3 core, then the ring around it as two rows and six single columns -- turns more candidates away with one
    // compare, but its eleven scans of mostly one point each run the list code once per SLOT of four)
#pragma unroll 1
    for (int s = 0; s < 5; ++s) {
      const int cy = cqy + (s == 0 ? 0 : (s == 1 ? -1 : (s == 2 ? 1 : (s == 3 ? -2 : 2))));
      if (cy < 0 || cy >= dy) continue;
      for_window(fc, x0, x1, cy, zlo, zhi, offer);
    }
    done = knn_kth<KMAX>(k, lst) < reach2 * reach2 * 0.999999f;
  }
  if (!done) {
    done = knn_rings<CoarseCols, KMAX, true>(tc, q, cqx, cqy, dx, dy, 0, min(kHaloKnn, max_ring), k, slack, cell_xy,
                                             leaf_z, lst);
    if (max_ring <= kHaloKnn) done = true;  // every column of the frame has been visited
  }
  return done;
}

// (K <= 8: 5 waves per SIMD at 96 VGPRs; 6 at 80 spill too much: 2.4 against 1.33 ms)
// The fast pass reads the 5x5 columns from an LDS copy of the workgroup's stretch of the sorted fit points when
// that fits (see k_normals_direct); the ring expansion of the few queries it does not settle reads HBM.
template <int KMAX>
__global__ void __launch_bounds__(kBlock, KMAX <= 8 ? 5 : 1)
k_knn_direct(const float4* spt_fit, const uint32_t* col_first_fit, const FrameInfo* fi, const int* fit_frame,
             float inv_xy, float inv_z, float cell_xy, int K, const uint32_t* fit_idx, unsigned own_lo, unsigned own_hi,
             unsigned n_fit, int* knn_out, uint32_t* todo, unsigned* n_todo)
{
  constexpr int kWinCap = 1536;  // (k-NN: no notes, so LDS has room for more than the normals kernel's 1024)
  __shared__ float4 s_pt[kWinCap];
  __shared__ unsigned s_w[2];
  const unsigned e0 = xcd_block() * kBlock + threadIdx.x;
  bool valid = e0 < n_fit;
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  int me = 0;
  if (valid) {
    q = spt_fit[e0];
    me = __float_as_int(q.w);
    if (own_hi - own_lo != 0xffffffffu) {  // staged multi-GPU call: only this rank's share of the queries
      const unsigned idx = fit_idx[me];
      valid = idx >= own_lo && idx < own_hi;
    }
  }
  const FrameInfo& F = fi[(valid && fit_frame) ? fit_frame[me] : 0];
  const int dx = F.div_b[0], dy = F.div_b[1];
  const int cqx = clampi(cell_unclamped(q.x, inv_xy, F.min_b[0]), 0, dx - 1);
  const int cqy = clampi(cell_unclamped(q.y, inv_xy, F.min_b[1]), 0, dy - 1);
  const float leaf_z = 1.0f / inv_z;
  const int k = min(K, F.n_fit);
  const CoarseCols tc{spt_fit, col_first_fit, nullptr, (unsigned)F.col_off, dx};
  if (threadIdx.x == 0) {
    s_w[0] = 0xffffffffu;
    s_w[1] = 0u;
  }
  __syncthreads();
  unsigned lo = 0xffffffffu, hi = 0u;
  if (valid) {
    const int x0 = max(cqx - 2, 0), x1 = min(cqx + 2, dx - 1);
    unsigned t;
    tc.row_range(x0, x1, max(cqy - 2, 0), lo, t);
    tc.row_range(x0, x1, min(cqy + 2, dy - 1), t, hi);
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    lo = min(lo, (unsigned)__shfl_xor((int)lo, off));
    hi = max(hi, (unsigned)__shfl_xor((int)hi, off));
  }
  if ((threadIdx.x & (kWave - 1)) == 0 && hi > 0u) {
    atomicMin(&s_w[0], lo);
    atomicMax(&s_w[1], hi);
  }
  __syncthreads();
  const unsigned w0 = s_w[0], w1 = s_w[1];
  const bool window = w1 > w0 && w1 - w0 <= (unsigned)kWinCap;  // (block-uniform)
  if (window) {
    for (unsigned i = threadIdx.x; i < w1 - w0; i += kBlock) s_pt[i] = spt_fit[w0 + i];
    __syncthreads();
  }
  if (!valid) return;
  knn_key lst[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) lst[j] = kKnnEmpty;
  const bool full = __all(k == KMAX);  // wave-uniform: the usual case (K = 8 and frames of more than 8 fit points)
  bool done;
  if (window) {
    const WindowCols wc{s_pt, w0, col_first_fit, (unsigned)F.col_off, dx};
    if (full)
      done = knn_direct_query<KMAX, true>(wc, tc, q, cqx, cqy, dx, dy, k, F.slack, cell_xy, leaf_z, lst);
    else
      done = knn_direct_query<KMAX, false>(wc, tc, q, cqx, cqy, dx, dy, k, F.slack, cell_xy, leaf_z, lst);
  } else {
    if (full)
      done = knn_direct_query<KMAX, true>(tc, tc, q, cqx, cqy, dx, dy, k, F.slack, cell_xy, leaf_z, lst);
    else
      done = knn_direct_query<KMAX, false>(tc, tc, q, cqx, cqy, dx, dy, k, F.slack, cell_xy, leaf_z, lst);
  }
#pragma unroll
  for (int j = 0; j < KMAX; ++j)
    if (j < K) knn_out[(size_t)me * K + j] = (j < k) ? (int)(unsigned)lst[j] : -1;
  if (!done) todo[atomicAdd(n_todo, 1u)] = (uint32_t)me;
}

// Phase B: the queries of `todo`, one WAVE each, by ring expansion over a whole-frame column
// index: the fine index of all points (few queries; rings 0..kHaloKnn are done unless flagged) or
// the coarse index of the fit points (many queries).  The 64 lanes take the columns of a ring in
// parallel, each keeping a private sorted list of candidates that beat the current k-th
// distance; the private lists are then merged, smallest first, into the list every lane holds.
// A far search therefore costs one memory round trip per ring instead of one per column.  The
// list phase A left in knn_out is taken up again (distances recomputed), so the z windows are
// narrow from the first ring on; points already listed are skipped.
template <int KMAX>
__device__ __forceinline__ void knn_insert(knn_key key, int k, knn_key (&lst)[KMAX])
{
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

__device__ __forceinline__ knn_key wave_min_key(knn_key v)
{
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, off);
    const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), off);
    const knn_key o = ((knn_key)hi << 32) | lo;
    v = o < v ? o : v;
  }
  return v;
}

template <class Cols, int KMAX>
__global__ void __launch_bounds__(kBlock)
k_knn_todo(const uint32_t* todo, const unsigned* n_todo, const float4* fit_pt, const int* fit_frame,
           const float4* idx_pt, const uint32_t* idx_first, const int* fit_rank, const FrameInfo* fi, float inv_xy,
           float inv_z, float cell_xy, int coarse, int K, int* knn_out)
{
  const unsigned t = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;  // wave-uniform
  if (t >= *n_todo) return;
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t entry = todo[t];
  const int me = (int)(entry & ~kTodoFromRing0);
  const FrameInfo& F = fi[fit_frame[me]];
  const float4 q = fit_pt[me];
  const int dx = (F.div_b[0] + coarse - 1) / coarse, dy = (F.div_b[1] + coarse - 1) / coarse;
  const int cqx = clampi(cell_unclamped(q.x, inv_xy, F.min_b[0]), 0, F.div_b[0] - 1) / coarse;
  const int cqy = clampi(cell_unclamped(q.y, inv_xy, F.min_b[1]), 0, F.div_b[1] - 1) / coarse;
  const int k = min(K, F.n_fit);
  const float leaf_z = 1.0f / inv_z;
  // a coarse column is `coarse` cells wide; the rounding slack of a cell index shrinks with it
  const float slack = F.slack / (float)coarse + (coarse > 1 ? 1e-6f : 0.f);
  const float cell = cell_xy * (float)coarse;
  const Cols cols{idx_pt, idx_first, fit_rank, (unsigned)(coarse > 1 ? F.ccol_off : F.col_off), dx};
  knn_key lst[KMAX];  // the same in every lane
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    lst[j] = kKnnEmpty;
    if (j < k) {
      const int r = knn_out[(size_t)me * K + j];
      if (r >= 0) {
        const float4 p = fit_pt[r];
        lst[j] = ((knn_key)__float_as_uint(dist2(q.x, q.y, q.z, p.x, p.y, p.z)) << 32) | (unsigned)r;
      }
    }
  }
  const int max_ring = max(max(cqx, dx - 1 - cqx), max(cqy, dy - 1 - cqy));
  const int first_ring = (coarse > 1 || (entry & kTodoFromRing0)) ? 0 : kHaloKnn + 1;
  for (int R = first_ring; R <= max_ring; ++R) {
    const int x0 = cqx - R, x1 = cqx + R, y0 = cqy - R, y1 = cqy + R;
    const int side = 2 * R + 1;
    const int n_pos = R == 0 ? 1 : 8 * R;
    const knn_key worst = knn_worst<KMAX>(k, lst);
    const float wd = key_dist(worst);
    const float rz = wd < INFINITY ? __builtin_sqrtf(wd) * 1.000001f + 1.5f * leaf_z : INFINITY;
    const float zlo = q.z - rz, zhi = q.z + rz;
    knn_key mine[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) mine[j] = kKnnEmpty;
    for (int pos = lane; pos < n_pos; pos += kWave) {
      int cx, cy;
      if (pos < side) {
        cx = x0 + pos;
        cy = y0;
      } else if (pos < 2 * side) {
        cx = x0 + (pos - side);
        cy = y1;
      } else if (pos < 2 * side + (side - 2)) {
        cx = x0;
        cy = y0 + 1 + (pos - 2 * side);
      } else {
        cx = x1;
        cy = y0 + 1 + (pos - 2 * side - (side - 2));
      }
      if (cx < 0 || cx >= dx || cy < 0 || cy >= dy) continue;
      unsigned lo, hi;
      cols.range(cx, cy, lo, hi);
      if (hi - lo > 8u) lo = lower_bound_z(cols, lo, hi, zlo);
      for (unsigned e = lo; e < hi; ++e) {
        const float4 p = cols.pt(e);
        if (p.z > zhi) break;
        const int pr = cols.fit(e, p);
        if (pr < 0) continue;
        const knn_key key = ((knn_key)__float_as_uint(dist2(q.x, q.y, q.z, p.x, p.y, p.z)) << 32) | (unsigned)pr;
        if (!(key < worst) || !(key < knn_worst<KMAX>(k, mine))) continue;
        bool dup = false;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) dup |= (lst[j] == key);
        if (dup) continue;
        knn_insert<KMAX>(key, k, mine);
      }
    }
    // merge: at most k of this ring's candidates can enter, taken in ascending order
    for (int it = 0; it < k; ++it) {
      const knn_key gmin = wave_min_key(mine[0]);
      if (!(gmin < knn_worst<KMAX>(k, lst))) break;
      knn_insert<KMAX>(gmin, k, lst);
      if (mine[0] == gmin) {  // keys are unique per point: exactly one lane owns it
#pragma unroll
        for (int j = 0; j + 1 < KMAX; ++j) mine[j] = mine[j + 1];
        mine[KMAX - 1] = kKnnEmpty;
      }
    }
    const float reach = ring_reach(R, slack, cell);
    if (reach > 0.f && knn_kth<KMAX>(k, lst) < reach * reach * 0.999999f) break;
  }
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
      if (j < K) knn_out[(size_t)me * K + j] = (j < k) ? (int)(unsigned)lst[j] : -1;
  }
}

// Phase B for MANY queries (a scene full of clutter leaves 10^5 of them): one THREAD per query over the coarse
// index, the rings one after the other (knn_rings, as in phase A).  A query then waits for every column of its rings
// in turn, but 64 queries share a wave and thousands of waves are in flight: throughput, where k_knn_todo buys the
// latency of a single query with a whole wave.
template <int KMAX>
__global__ void __launch_bounds__(kBlock)
k_knn_todo_threads(const uint32_t* todo, const unsigned* n_todo, const float4* fit_pt, const int* fit_frame,
                   const float4* idx_pt, const uint32_t* idx_first, const FrameInfo* fi, float inv_xy, float inv_z,
                   float cell_xy, int coarse, int K, int* knn_out)
{
  const unsigned t = xcd_block() * kBlock + threadIdx.x;  // (neighbours in the list are neighbours in space)
  if (t >= *n_todo) return;
  const uint32_t entry = todo[t];
  const int me = (int)(entry & ~kTodoFromRing0);
  const FrameInfo& F = fi[fit_frame[me]];
  const float4 q = fit_pt[me];
  const int dx = (F.div_b[0] + coarse - 1) / coarse, dy = (F.div_b[1] + coarse - 1) / coarse;
  const int cqx = clampi(cell_unclamped(q.x, inv_xy, F.min_b[0]), 0, F.div_b[0] - 1) / coarse;
  const int cqy = clampi(cell_unclamped(q.y, inv_xy, F.min_b[1]), 0, F.div_b[1] - 1) / coarse;
  const int k = min(K, F.n_fit);
  const float leaf_z = 1.0f / inv_z;
  const float slack = F.slack / (float)coarse + (coarse > 1 ? 1e-6f : 0.f);
  const float cell = cell_xy * (float)coarse;
  const CoarseCols cols{idx_pt, idx_first, nullptr, (unsigned)(coarse > 1 ? F.ccol_off : F.col_off), dx};
  knn_key lst[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    lst[j] = kKnnEmpty;
    if (j < k) {
      const int r = knn_out[(size_t)me * K + j];
      if (r >= 0) {
        const float4 p = fit_pt[r];
        lst[j] = ((knn_key)__float_as_uint(dist2(q.x, q.y, q.z, p.x, p.y, p.z)) << 32) | (unsigned)r;
      }
    }
  }
  const int max_ring = max(max(cqx, dx - 1 - cqx), max(cqy, dy - 1 - cqy));
  const int first_ring = (coarse > 1 || (entry & kTodoFromRing0)) ? 0 : kHaloKnn + 1;
  knn_rings<CoarseCols, KMAX, true>(cols, q, cqx, cqy, dx, dy, first_ring, max_ring, k, slack, cell, leaf_z, lst);
#pragma unroll
  for (int j = 0; j < KMAX; ++j)
    if (j < K) knn_out[(size_t)me * K + j] = (j < k) ? (int)(unsigned)lst[j] : -1;
}

}  // namespace hope
