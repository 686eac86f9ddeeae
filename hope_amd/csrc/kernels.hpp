// kernels.hpp — HIP kernels of the HoPE PlaneSegment hot path for gfx950 (MI355X).
//
// Everything here is HBM/L2-bound integer and fp32 work: no MFMA.
// Arithmetic that feeds a discontinuous decision (voxel index, |nz| gate, dz gate,
// normalAnalysis thresholds) uses only IEEE + - * / sqrt in a fixed order, and this
// translation unit is compiled with -ffp-contract=off, so results are bit-identical to
// the scalar CPU statement of the reference.  Reference citations (file:line, relative
// to DrawZeroPoint/hope) are given per kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/hope_detmath.h"
#include "../../include/hope_gpu.h"
#include "prims.hpp"

namespace hope {

constexpr int kBlock = 256;
constexpr int kWave = 64;
constexpr int kMaxKnn = 32;
constexpr int kCoarse = 4;   // coarse k-NN index: columns of kCoarse x kCoarse cells (1e8-point cloud, leftovers stage: 8 -> 0.90 ms, 4 -> 0.67, 3 -> 0.72)

// Per-frame state, resident in HBM for the whole call (one per depth frame / cloud).
struct FrameInfo {
  float aff[12];        // rotation r[9] row-major + translation t[3]   (A3)
  unsigned bb_min[3];   // order-preserving encodings of the bbox        (A5 getMinMax3D)
  unsigned bb_max[3];
  int n_raw;            // points that survive A1/A2
  int min_b[3];         // VoxelGrid min_b_                              (A5)
  int div_b[3];         // VoxelGrid div_b_
  int overflow;         // dx*dy*dz > INT_MAX (PCL copies the input through)
  long long cells;      // div_b product
  float slack;          // float-rounding slack of a column index, in cells (kNN bound)
  int col_off;          // first entry of this frame in the column table
  int ccol_off;         // first entry in the coarse (8x8 columns) table of the fit points
  int vox_start, n_vox; // range in the global down-sampled arrays
  int fit_start, n_fit; // range in the global normal-filtered arrays
  int seg_start, n_segments;
  int cl_start, n_clusters;
  int status;
};

struct Globals {  // scalars produced on device, read back at the sync points
  unsigned long long max_cells;
  unsigned int total_cols;
  unsigned int max_div_z;
  unsigned int any_overflow;
  unsigned int n_runs;
  unsigned int changed;
  unsigned int n_vox_total;
  unsigned int pad;
  unsigned int n_groups;  // depth path: number of (chunk, voxel) group records
  unsigned int max_frame_clusters;  // k_frame_cluster_ranges: most clusters in one frame
  unsigned int max_tiles; // largest per-frame count of 32x32-column super tiles (neighbours.hpp)
  unsigned int max_tiles16;  // ... of 16x16-column tiles
  unsigned int sum_tiles;    // super tiles of all frames (the host picks the tile side from the points per super tile)
  unsigned int max_cols;     // largest per-frame column table (columns + 1)
  unsigned int total_ccols;  // entries of the coarse column table
  unsigned int n_todo;       // k-NN queries left over by the tile pass
  unsigned int max_fit;      // largest per-frame count of fit points
  unsigned int code_overflow;  // k_bbox_records: a cell lies outside the packed code range
  unsigned int rank_overflow;  // k_voxel_rank: a frame has more voxels than the LDS counters hold
  unsigned int max_vox;        // k_frame_vox_scan: largest per-frame count of voxels
  unsigned int n_big;          // k_big_count: clusters of verify_big_from members and more ...
  unsigned int n_big_sb;       // ... and their 1024-member super blocks (verify_big.hpp)
  unsigned int grid_unsorted;  // k_grid_mark: the derived cells do not strictly ascend along the down-sampled cloud
};

struct DepthSrc {
  const uint16_t* depth;
  int w, h;
  unsigned ppf;  // pixels per frame
  float fx, fy, cx, cy, scale, min_d, max_d;
  float rcp_fx, rcp_fy, rcp_scale;  // 1/fx, 1/fy, 1/scale rounded to float (div_const)
};

struct CloudSrc {
  const char* base;
  unsigned long long stride;
  unsigned n;
  int apply;  // 1: PassThrough z in [min_d, max_d] + pose transform
  float min_d, max_d;
};

// ---- small device helpers ------------------------------------------------------------
__device__ __forceinline__ unsigned f2ord(float f)
{
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u)
{
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ bool finite3(float x, float y, float z)
{
  return isfinite(x) && isfinite(y) && isfinite(z);
}

// A3: Eigen Affine3f * Vector3f = t + (r0*x + (r1*y + r2*z))   (transform.cpp:168-185)
__device__ __forceinline__ void apply_affine(const float* a, float x, float y, float z, float& ox, float& oy,
                                             float& oz)
{
  ox = a[9] + (a[0] * x + (a[1] * y + a[2] * z));
  oy = a[10] + (a[3] * x + (a[4] * y + a[5] * z));
  oz = a[11] + (a[6] * x + (a[7] * y + a[8] * z));
}

// A1 + A2 + A3 for one pixel of a TUM-style depth frame (get_cloud.cpp:220-233,
// utilities.cpp:662-675, transform.cpp:152-191).  Returns false for a dropped pixel.
__device__ __forceinline__ bool load_point(const DepthSrc& s, const FrameInfo* fi, int frame, unsigned pix,
                                           float& ox, float& oy, float& oz)
{
  const uint16_t d = s.depth[(size_t)frame * s.ppf + pix];
  const int col = (int)(pix % (unsigned)s.w);
  const int row = (int)(pix / (unsigned)s.w);
  const float z = (float)d / s.scale;
  if (!(z < s.max_d && z > s.min_d)) return false;
  const float x = ((float)col - s.cx) * z / s.fx;
  const float y = ((float)row - s.cy) * z / s.fy;
  apply_affine(fi[frame].aff, x, y, z, ox, oy, oz);
  return true;
}

// A2 + A3 for one record of a point cloud (plane_segment.cpp:118-127,791-795)
__device__ __forceinline__ bool load_point(const CloudSrc& s, const FrameInfo* fi, int frame, unsigned i,
                                           float& ox, float& oy, float& oz)
{
  const float* p = reinterpret_cast<const float*>(s.base + (size_t)i * s.stride);
  const float x = p[0], y = p[1], z = p[2];
  if (!finite3(x, y, z)) return false;
  if (s.apply) {
    if (z < s.min_d || z > s.max_d) return false;
    apply_affine(fi[frame].aff, x, y, z, ox, oy, oz);
    return finite3(ox, oy, oz);
  }
  ox = x;
  oy = y;
  oz = z;
  return true;
}

__device__ __forceinline__ unsigned src_count(const DepthSrc& s) { return s.ppf; }
__device__ __forceinline__ unsigned src_count(const CloudSrc& s) { return s.n; }

// lanes of the wave whose `ok` is set and whose low `bits` bits of d equal this lane's
__device__ __forceinline__ unsigned long long match_lanes(unsigned d, int bits, bool ok)
{
  unsigned long long same = __builtin_amdgcn_ballot_w64(ok);
  for (int b = 0; b < bits; ++b) {
    const bool bit = (d >> b) & 1u;
    const unsigned long long v = __builtin_amdgcn_ballot_w64(bit);
    same &= bit ? v : ~v;
  }
  return same;
}

// Workgroups are dealt to the eight XCDs in turn, each with its own L2: kernels whose neighbouring
// blocks read the same lines (windows of the sorted array, the rows and roots of neighbours in index) let XCD x take
// the x-th eighth of the blocks, in order.  The grid must be a multiple of 8 blocks (grid8).
__device__ __forceinline__ unsigned xcd_block()
{
  const unsigned per = (gridDim.x + 7u) / 8u;
  return (blockIdx.x % 8u) * per + blockIdx.x / 8u;  // (may be >= gridDim.x: its queries are then out of range)
}

// ---- K1: bbox + count per frame ------------------------------------------------------
// pcl::getMinMax3D inside VoxelGrid::applyFilter.  grid = (blocks, n_frames).
template <class Src>
__global__ void __launch_bounds__(kBlock) k_bbox(Src src, FrameInfo* fi)
{
  const int frame = blockIdx.y;
  const unsigned n = src_count(src);
  float mn[3] = {INFINITY, INFINITY, INFINITY};
  float mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  int cnt = 0;
  for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    float x, y, z;
    if (load_point(src, fi, frame, i, x, y, z)) {
      mn[0] = fminf(mn[0], x);
      mn[1] = fminf(mn[1], y);
      mn[2] = fminf(mn[2], z);
      mx[0] = fmaxf(mx[0], x);
      mx[1] = fmaxf(mx[1], y);
      mx[2] = fmaxf(mx[2], z);
      ++cnt;
    }
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      mn[d] = fminf(mn[d], __shfl_down(mn[d], off));
      mx[d] = fmaxf(mx[d], __shfl_down(mx[d], off));
    }
    cnt += __shfl_down(cnt, off);
  }
  __shared__ float s_mn[kBlock / kWave][3], s_mx[kBlock / kWave][3];
  __shared__ int s_cnt[kBlock / kWave];
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  if (lane == 0) {
    for (int d = 0; d < 3; ++d) {
      s_mn[wv][d] = mn[d];
      s_mx[wv][d] = mx[d];
    }
    s_cnt[wv] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / kWave; ++w) {
      for (int d = 0; d < 3; ++d) {
        mn[d] = fminf(mn[d], s_mn[w][d]);
        mx[d] = fmaxf(mx[d], s_mx[w][d]);
      }
      cnt += s_cnt[w];
    }
    if (cnt > 0) {
      for (int d = 0; d < 3; ++d) {
        atomicMin(&fi[frame].bb_min[d], f2ord(mn[d]));
        atomicMax(&fi[frame].bb_max[d], f2ord(mx[d]));
      }
      atomicAdd(&fi[frame].n_raw, cnt);
    }
  }
}

// ---- K2: VoxelGrid grid set-up per frame (PCL 1.8.1 voxel_grid.hpp applyFilter) -------
static __global__ void __launch_bounds__(1024) k_frame_grid(FrameInfo* fi, int n_frames, float inv_xy, float inv_z, unsigned n_per_frame, Globals* g)
{
  // one 1024-thread block: per-frame grid set-up in parallel, then a block-wide exclusive
  // scan of the per-frame column-table sizes
  __shared__ unsigned s_scan[1024];
  __shared__ unsigned long long s_max_cells;
  __shared__ unsigned long long s_all_cols;
  __shared__ unsigned s_max_dz, s_any_over, s_carry, s_carry2, s_max_tiles, s_max_tiles16, s_sum_tiles, s_max_cols;
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_all_cols = 0;
    s_max_cells = 1;
    s_max_dz = 1;
    s_any_over = 0;
    s_carry = 0;
    s_carry2 = 0;
    s_max_tiles = 0;
    s_max_tiles16 = 0;
    s_sum_tiles = 0;
    s_max_cols = 0;
  }
  __syncthreads();
  for (int base = 0; base < n_frames; base += 1024) {
    const int f = base + tid;
    unsigned ncols = 0, nccols = 0;
    if (f < n_frames) {
      FrameInfo& F = fi[f];
      int min_b[3] = {0, 0, 0}, div_b[3] = {0, 0, 0};
      int overflow = 0;
      long long cells = 0;
      float slack = 0.f;
      ncols = 1;
      nccols = 1;
      if (F.n_raw > 0) {
        const float inv[3] = {inv_xy, inv_xy, inv_z};
        long long dxyz = 1;
        float maxabs = 0.f, maxabs3 = 0.f;
        for (int d = 0; d < 3; ++d) {
          const float mn = ord2f(F.bb_min[d]), mx = ord2f(F.bb_max[d]);
          dxyz *= (long long)((mx - mn) * inv[d]) + 1;
          const float m = fmaxf(fabsf(mn), fabsf(mx)) * inv[d];
          maxabs3 = fmaxf(maxabs3, m);
          if (d < 2) maxabs = fmaxf(maxabs, m);
        }
        // (a cell coordinate beyond 2^30 has no int representation worth computing: PCL's own cast
        // would be undefined there, and such a frame cannot be indexed by columns either)
        const bool indexable = maxabs3 < 1073741824.f;
        if (indexable) {
          for (int d = 0; d < 3; ++d) {
            min_b[d] = (int)floorf(ord2f(F.bb_min[d]) * inv[d]);
            div_b[d] = (int)floorf(ord2f(F.bb_max[d]) * inv[d]) - min_b[d] + 1;
          }
        }
        const long long cols = (long long)div_b[0] * div_b[1];
        if (!indexable || (dxyz > 2147483647LL && cols > 0x3fffffffLL)) {
          // no column table of a sane size: the call is refused
          overflow = 1;
          atomicOr(&s_any_over, 2u);
          for (int d = 0; d < 3; ++d) min_b[d] = div_b[d] = 0;
        } else {
          if (dxyz > 2147483647LL) {
            // "Leaf size is too small for the input dataset": PCL copies the input through
            // (voxel_grid.hpp applyFilter).  Every surviving point becomes a voxel of its own (key =
            // its index in the frame); min_b / div_b only shape the column index of A6 / A8
            overflow = 1;
            atomicOr(&s_any_over, 1u);
            cells = 0;
            atomicMax(&s_max_cells, (unsigned long long)n_per_frame);
          } else {
            cells = (long long)div_b[0] * div_b[1] * div_b[2];
            atomicMax(&s_max_cells, (unsigned long long)cells);
          }
          // |fl(x*inv) - x*inv| <= 2^-24 * |x*inv|; two such errors enter a column difference
          slack = maxabs * 2.4e-7f + 1e-6f;
          atomicMax(&s_max_dz, (unsigned)div_b[2]);
          ncols = (unsigned)(div_b[0] * div_b[1]) + 1u;
          nccols = (unsigned)(((div_b[0] + kCoarse - 1) / kCoarse) * ((div_b[1] + kCoarse - 1) / kCoarse)) + 1u;
          const unsigned t32 = (unsigned)(((div_b[0] + 31) / 32) * ((div_b[1] + 31) / 32));  // (neighbours.hpp: kSuper)
          atomicMax(&s_max_tiles, t32);
          atomicAdd(&s_sum_tiles, t32);
          atomicMax(&s_max_tiles16, (unsigned)(((div_b[0] + 15) / 16) * ((div_b[1] + 15) / 16)));  // (kTile)
        }
      }
      for (int d = 0; d < 3; ++d) {
        F.min_b[d] = min_b[d];
        F.div_b[d] = div_b[d];
      }
      F.overflow = overflow;
      F.cells = cells;
      F.slack = slack;
    }
    if (ncols > 1u) atomicAdd(&s_all_cols, (unsigned long long)ncols);
    // inclusive Hillis-Steele scan of ncols over the chunk
    s_scan[tid] = ncols;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const unsigned add = tid >= off ? s_scan[tid - off] : 0u;
      __syncthreads();
      s_scan[tid] += add;
      __syncthreads();
    }
    const unsigned carry = s_carry;
    if (f < n_frames) {
      fi[f].col_off = (int)(carry + s_scan[tid] - ncols);
      atomicMax(&s_max_cols, ncols);
    }
    __syncthreads();
    if (tid == 1023) s_carry = carry + s_scan[1023];
    __syncthreads();
    // the same scan for the coarse column table
    s_scan[tid] = nccols;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const unsigned add = tid >= off ? s_scan[tid - off] : 0u;
      __syncthreads();
      s_scan[tid] += add;
      __syncthreads();
    }
    const unsigned carry2 = s_carry2;
    if (f < n_frames) fi[f].ccol_off = (int)(carry2 + s_scan[tid] - nccols);
    __syncthreads();
    if (tid == 1023) s_carry2 = carry2 + s_scan[1023];
    __syncthreads();
  }
  if (tid == 0) {
    g->max_cells = s_max_cells;
    g->total_cols = s_carry;
    g->max_div_z = s_max_dz;
    g->any_overflow = s_any_over | (s_all_cols > 0x7fffffffull ? 4u : 0u);  // (col_off is an int)
    g->max_tiles = s_max_tiles;
    g->max_tiles16 = s_max_tiles16;
    g->sum_tiles = s_sum_tiles;
    g->max_cols = s_max_cols;
    g->total_ccols = s_carry2;
  }
}

// ---- K3: voxel keys -------------------------------------------------------------------
// key = frame << cell_bits | idx, idx = i + j*div_x + k*div_x*div_y with
// ijk = (int)(floor(p*inv) - (float)min_b)   (voxel_grid.hpp, second pass)
template <class Src, class KeyT>
__global__ void __launch_bounds__(kBlock)
k_voxel_keys(Src src, const FrameInfo* fi, float inv_xy, float inv_z, int cell_bits, int n_frames, KeyT* keys,
             uint32_t* vals)
{
  const int frame = blockIdx.y;
  const unsigned n = src_count(src);
  const FrameInfo& F = fi[frame];
  for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    float x, y, z;
    KeyT key = (KeyT)n_frames << cell_bits;  // sentinel: sorts behind every frame
    if (load_point(src, fi, frame, i, x, y, z)) {
      const int i0 = (int)(floorf(x * inv_xy) - (float)F.min_b[0]);
      const int i1 = (int)(floorf(y * inv_xy) - (float)F.min_b[1]);
      const int i2 = (int)(floorf(z * inv_z) - (float)F.min_b[2]);
      // (pass-through frame, F.overflow: every point is a voxel of its own, in input order)
      const unsigned idx = F.overflow ? i : (unsigned)(i0 + i1 * F.div_b[0] + i2 * (F.div_b[0] * F.div_b[1]));
      key = ((KeyT)frame << cell_bits) | (KeyT)idx;
    }
    const size_t o = (size_t)frame * n + i;
    keys[o] = key;
    vals[o] = (uint32_t)o;
  }
}

// ---- depth-frame fast paths: 8 pixels (one 16-byte load) per thread ----------------------
// Same arithmetic as load_point(DepthSrc); requires w % 8 == 0 and a 16-byte aligned buffer.
struct Px8 {
  float x[8], y[8], z[8];
  unsigned valid;  // bit j set: pixel j survives A1/A2
};

__device__ __forceinline__ void load_px8(const DepthSrc& s, const float* aff, int frame, unsigned p0, Px8& o)
{
  const uint4 raw = *reinterpret_cast<const uint4*>(s.depth + (size_t)frame * s.ppf + p0);
  const unsigned wds[4] = {raw.x, raw.y, raw.z, raw.w};
  const int row = (int)(p0 / (unsigned)s.w);
  const int col0 = (int)(p0 - (unsigned)row * (unsigned)s.w);
  const float fr = (float)row - s.cy;
  o.valid = 0u;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const unsigned d = (wds[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
    const float z = (float)d / s.scale;
    if (z < s.max_d && z > s.min_d) {
      const float x = ((float)(col0 + j) - s.cx) * z / s.fx;
      const float y = fr * z / s.fy;
      apply_affine(aff, x, y, z, o.x[j], o.y[j], o.z[j]);
      o.valid |= 1u << j;
    }
  }
}

static __global__ void __launch_bounds__(kBlock) k_bbox_depth8(DepthSrc src, FrameInfo* fi)
{
  const int frame = blockIdx.y;
  const unsigned n8 = src.ppf / 8;
  const float* aff = fi[frame].aff;
  float mn[3] = {INFINITY, INFINITY, INFINITY};
  float mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  int cnt = 0;
  for (unsigned t = blockIdx.x * kBlock + threadIdx.x; t < n8; t += gridDim.x * kBlock) {
    Px8 p;
    load_px8(src, aff, frame, t * 8, p);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (p.valid & (1u << j)) {
        mn[0] = fminf(mn[0], p.x[j]);
        mn[1] = fminf(mn[1], p.y[j]);
        mn[2] = fminf(mn[2], p.z[j]);
        mx[0] = fmaxf(mx[0], p.x[j]);
        mx[1] = fmaxf(mx[1], p.y[j]);
        mx[2] = fmaxf(mx[2], p.z[j]);
      }
    }
    cnt += __popc(p.valid);
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      mn[d] = fminf(mn[d], __shfl_down(mn[d], off));
      mx[d] = fmaxf(mx[d], __shfl_down(mx[d], off));
    }
    cnt += __shfl_down(cnt, off);
  }
  if ((threadIdx.x & (kWave - 1)) == 0 && cnt > 0) {
    for (int d = 0; d < 3; ++d) {
      atomicMin(&fi[frame].bb_min[d], f2ord(mn[d]));
      atomicMax(&fi[frame].bb_max[d], f2ord(mx[d]));
    }
    atomicAdd(&fi[frame].n_raw, cnt);
  }
}

template <class KeyT>
__global__ void __launch_bounds__(kBlock)
k_voxel_keys_depth8(DepthSrc src, const FrameInfo* fi, float inv_xy, float inv_z, int cell_bits, int n_frames,
                    KeyT* keys, uint32_t* vals)
{
  const int frame = blockIdx.y;
  const unsigned n8 = src.ppf / 8;
  const FrameInfo& F = fi[frame];
  const float* aff = F.aff;
  const int mb0 = F.min_b[0], mb1 = F.min_b[1], mb2 = F.min_b[2];
  const int d0 = F.div_b[0], d01 = F.div_b[0] * F.div_b[1];
  const KeyT sentinel = (KeyT)n_frames << cell_bits;
  const KeyT fbits = (KeyT)frame << cell_bits;
  const bool over = F.overflow != 0;
  for (unsigned t = blockIdx.x * kBlock + threadIdx.x; t < n8; t += gridDim.x * kBlock) {
    Px8 p;
    load_px8(src, aff, frame, t * 8, p);
    const size_t o = (size_t)frame * src.ppf + (size_t)t * 8;
    KeyT k[8];
    uint32_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      k[j] = sentinel;
      if (p.valid & (1u << j)) {
        const int i0 = (int)(floorf(p.x[j] * inv_xy) - (float)mb0);
        const int i1 = (int)(floorf(p.y[j] * inv_xy) - (float)mb1);
        const int i2 = (int)(floorf(p.z[j] * inv_z) - (float)mb2);
        k[j] = fbits | (KeyT)(over ? t * 8 + (unsigned)j : (unsigned)(i0 + i1 * d0 + i2 * d01));
      }
      v[j] = (uint32_t)(o + j);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) keys[o + j] = k[j];
    uint4* vp = reinterpret_cast<uint4*>(vals + o);
    vp[0] = make_uint4(v[0], v[1], v[2], v[3]);
    vp[1] = make_uint4(v[4], v[5], v[6], v[7]);
  }
}

// ---- K5: voxel centroids (CentroidPoint<PointXYZ>: sequential float sum / n), per-point path ---
// A voxel's points arrive in ascending source index (the stable sort keeps them so) and are
// re-derived from the source instead of re-read at 12 B/pt.
// A wave owns 64 consecutive voxels (lane = voxel).  Per round it fetches the next kR entries
// of every voxel with coalesced loads (16 lanes per voxel), re-derives the points in parallel,
// parks them in LDS and then every lane adds ITS voxel's kR points sequentially — the same
// ascending-index chain as one thread per voxel would run, but with 64 chains per wave
// running from LDS instead of 64 dependent uncoalesced gathers.
constexpr int kR = 16;          // entries per voxel per round
constexpr int kRStride = 65;    // LDS row stride (dwords): conflict-free reads, <=4-way writes

// one round of the load phase; UNI: every voxel of the wave lies in frame f0, so the pose is
// fetched with scalar loads.  Branch-free (clamped indices) so that all 16 index loads and all
// 16 depth gathers of a round are in flight together.
template <bool UNI, class Src>
__device__ __forceinline__ void centroid_load_round(const Src& src, const FrameInfo* fi, const uint32_t* vals_sorted,
                                                    unsigned n, unsigned r0, int f0, int frame, unsigned my_start,
                                                    unsigned my_cnt, int sub, int grp, float* s_x, float* s_y,
                                                    float* s_z)
{
  unsigned o[kR];
  int fv[kR];
#pragma unroll
  for (int q = 0; q < kR; ++q) {
    const int vl = q * (kWave / kR) + grp;
    const unsigned cnt_v = (unsigned)__shfl((int)my_cnt, vl);
    const unsigned start_v = (unsigned)__shfl((int)my_start, vl);
    fv[q] = UNI ? f0 : __shfl(frame, vl);
    const unsigned e = min(r0 + (unsigned)sub, cnt_v > 0u ? cnt_v - 1u : 0u);
    o[q] = vals_sorted[start_v + e];
  }
#pragma unroll
  for (int q = 0; q < kR; ++q) {
    const int vl = q * (kWave / kR) + grp;
    const unsigned i = o[q] - (unsigned)((size_t)fv[q] * n);
    float x = 0.f, y = 0.f, z = 0.f;
    load_point(src, fi, fv[q], i < n ? i : 0u, x, y, z);
    s_x[sub * kRStride + vl] = x;
    s_y[sub * kRStride + vl] = y;
    s_z[sub * kRStride + vl] = z;
  }
}

template <class Src, class KeyT>
__global__ void __launch_bounds__(kWave)
k_centroids_w(Src src, const FrameInfo* fi, const KeyT* unique, const uint32_t* run_start, const uint32_t* run_count,
              const uint32_t* vals_sorted, unsigned n_vox, int cell_bits, float4* vox, int* vox_frame)
{
  __shared__ float s_x[kR * kRStride], s_y[kR * kRStride], s_z[kR * kRStride];
  const int lane = threadIdx.x;
  const unsigned v = blockIdx.x * kWave + lane;
  const bool live = v < n_vox;
  const int frame = live ? (int)(unique[v] >> cell_bits) : 0;
  const unsigned my_start = live ? run_start[v] : 0u;
  const unsigned my_cnt = live ? run_count[v] : 0u;
  const unsigned n = src_count(src);
  unsigned max_cnt = my_cnt;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) max_cnt = max(max_cnt, (unsigned)__shfl_xor((int)max_cnt, off));
  const int f0 = __builtin_amdgcn_readfirstlane(frame);
  const bool uni = __all(!live || frame == f0) != 0;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  const int sub = lane & (kR - 1);   // entry within the round
  const int grp = lane / kR;         // which of the 4 voxels this load serves
  for (unsigned r0 = 0; r0 < max_cnt; r0 += kR) {
    if (uni)
      centroid_load_round<true>(src, fi, vals_sorted, n, r0, f0, frame, my_start, my_cnt, sub, grp, s_x, s_y, s_z);
    else
      centroid_load_round<false>(src, fi, vals_sorted, n, r0, f0, frame, my_start, my_cnt, sub, grp, s_x, s_y, s_z);
    __syncthreads();
    const int k = (int)min((unsigned)kR, my_cnt > r0 ? my_cnt - r0 : 0u);
    for (int j = 0; j < k; ++j) {
      sx += s_x[j * kRStride + lane];
      sy += s_y[j * kRStride + lane];
      sz += s_z[j * kRStride + lane];
    }
    __syncthreads();
  }
  if (live) {
    const float c = (float)my_cnt;
    float4 out = make_float4(sx / c, sy / c, sz / c, 0.f);
    if (fi[frame].overflow) {
      // pass-through frame: the point itself (0.f + x would lose the sign of a -0.f)
      load_point(src, fi, frame, vals_sorted[my_start] - (unsigned)((size_t)frame * n), out.x, out.y, out.z);
    }
    vox[v] = out;
    vox_frame[v] = frame;
  }
}

// ---- column index over the down-sampled cloud -----------------------------------------
// Index function (any monotone function is exact, it only steers the search):
// col = clamp(floor(x*inv) - min_b), cz likewise; key2 = (col_off + cy*div_x + cx) << zbits | cz
__device__ __forceinline__ int cell_unclamped(float x, float inv, int min_b)
{
  return (int)(floorf(x * inv) - (float)min_b);
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <class KeyT>
__global__ void __launch_bounds__(kBlock)
k_column_keys(const float4* vox, const int* vox_frame, const FrameInfo* fi, unsigned n_vox, float inv_xy,
              float inv_z, int zbits, KeyT* keys, uint32_t* vals)
{
  const unsigned v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= n_vox) return;
  const FrameInfo& F = fi[vox_frame[v]];
  const float4 p = vox[v];
  const int cx = clampi(cell_unclamped(p.x, inv_xy, F.min_b[0]), 0, F.div_b[0] - 1);
  const int cy = clampi(cell_unclamped(p.y, inv_xy, F.min_b[1]), 0, F.div_b[1] - 1);
  const int cz = clampi(cell_unclamped(p.z, inv_z, F.min_b[2]), 0, F.div_b[2] - 1);
  const unsigned col = (unsigned)F.col_off + (unsigned)(cy * F.div_b[0] + cx);
  keys[v] = zbits ? (((KeyT)col << zbits) | (KeyT)(unsigned)cz) : (KeyT)col;  // (zbits 0: by column alone)
  vals[v] = v;
}

// col_first[c] = first sorted entry whose column is >= c (c = 0 .. total_cols).  Entry i opens the
// columns (column(i-1), column(i)]; the thread one past the end closes the table.
template <class KeyT>
__global__ void __launch_bounds__(kBlock)
k_column_table(const KeyT* skeys, unsigned n_vox, unsigned total_cols, int zbits, uint32_t* col_first)
{
  const unsigned i = blockIdx.x * kBlock + threadIdx.x;
  if (i > n_vox) return;
  const long long prev = i > 0u ? (long long)(skeys[i - 1u] >> zbits) : -1ll;
  const long long cur = i < n_vox ? (long long)(skeys[i] >> zbits) : (long long)total_cols;
  for (long long c = prev + 1; c <= cur; ++c) col_first[c] = i;
}

// sorted copy: xyz + original (PCL-order) index in .w
static __global__ void __launch_bounds__(kBlock)
k_gather_sorted(const float4* vox, const uint32_t* svals, unsigned n_vox, float4* spt)
{
  const unsigned s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n_vox) return;
  const uint32_t v = svals[s];
  float4 p = vox[v];
  p.w = __uint_as_float(v);
  spt[s] = p;
}

// ---- coarse index over the FIT points only (k-NN queries that leave the LDS tile) -------------
// Same layout as the column index, with columns of kCoarse x kCoarse cells: a ring of coarse
// columns covers kCoarse times the distance per table lookup, and points that failed the
// normal gate are not in it at all.
template <class KeyT>
__global__ void __launch_bounds__(kBlock)
k_coarse_keys(const float4* fit_pt, const int* fit_frame, const FrameInfo* fi, unsigned n_fit, float inv_xy,
              float inv_z, int zbits, KeyT* keys, uint32_t* vals)
{
  const unsigned g = blockIdx.x * kBlock + threadIdx.x;
  if (g >= n_fit) return;
  const FrameInfo& F = fi[fit_frame[g]];
  const float4 p = fit_pt[g];
  const int cx = clampi(cell_unclamped(p.x, inv_xy, F.min_b[0]), 0, F.div_b[0] - 1) / kCoarse;
  const int cy = clampi(cell_unclamped(p.y, inv_xy, F.min_b[1]), 0, F.div_b[1] - 1) / kCoarse;
  const int cz = clampi(cell_unclamped(p.z, inv_z, F.min_b[2]), 0, F.div_b[2] - 1);
  const int dxc = (F.div_b[0] + kCoarse - 1) / kCoarse;
  const unsigned col = (unsigned)F.ccol_off + (unsigned)(cy * dxc + cx);
  keys[g] = ((KeyT)col << zbits) | (KeyT)(unsigned)cz;
  vals[g] = g;
}

// sorted copy of the fit points: xyz + fit rank in .w
static __global__ void __launch_bounds__(kBlock)
k_gather_coarse(const float4* fit_pt, const uint32_t* svals, unsigned n_fit, float4* cspt)
{
  const unsigned s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n_fit) return;
  const uint32_t v = svals[s];
  float4 p = fit_pt[v];
  p.w = __int_as_float((int)v);
  cspt[s] = p;
}

// ---- column index of the FIT points only (k-NN) ---------------------------------------------------
// A stable compaction of the sorted copy keeps the (column, z) order, so the fit points need no sort
// of their own: flags in sorted order, their scan, the compacted copy (.w = fit rank) and the column
// table looked up through the scan.
static __global__ void __launch_bounds__(kBlock)
k_fit_sorted_flags(const float4* spt, const int* fit_rank, unsigned n_vox, uint32_t* flags)
{
  const unsigned i = blockIdx.x * kBlock + threadIdx.x;
  if (i > n_vox) return;
  flags[i] = (i < n_vox && fit_rank[__float_as_uint(spt[i].w)] >= 0) ? 1u : 0u;
}

static __global__ void __launch_bounds__(kBlock)
k_fit_sorted_scatter(const float4* spt, const int* fit_rank, const uint32_t* flags, const uint32_t* fpos,
                     unsigned n_vox, float4* spt_fit)
{
  const unsigned i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_vox || !flags[i]) return;
  float4 p = spt[i];
  p.w = __int_as_float(fit_rank[__float_as_uint(p.w)]);
  spt_fit[fpos[i]] = p;
}

static __global__ void __launch_bounds__(kBlock)
k_fit_col_table(const uint32_t* col_first, const uint32_t* fpos, unsigned total_cols, uint32_t* col_first_fit)
{
  const unsigned c = blockIdx.x * kBlock + threadIdx.x;
  if (c <= total_cols) col_first_fit[c] = fpos[col_first[c]];
}

// FLANN L2_Simple<float>: ((dx*dx + dy*dy) + dz*dz)
__device__ __forceinline__ float dist2(float ax, float ay, float az, float bx, float by, float bz)
{
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return (dx * dx + dy * dy) + dz * dz;
}

__device__ __forceinline__ bool di_less(float d1, unsigned i1, float d2, unsigned i2)
{
  return d1 < d2 || (d1 == d2 && i1 < i2);
}

// ---- pcl::computeRoots2 / computeRoots / eigen33 (PCL 1.8.1 common/impl/eigen.hpp) -----
__device__ inline void compute_roots2(float b, float c, float* roots)
{
  roots[0] = 0.f;
  float d = (float)((double)(b * b) - 4.0 * (double)c);
  if (d < 0.0) d = 0.0;
  const float sd = (float)__builtin_sqrt((double)d);
  roots[2] = 0.5f * (b + sd);
  roots[1] = 0.5f * (b - sd);
}

__device__ inline void swapf(float& a, float& b)
{
  const float t = a;
  a = b;
  b = t;
}

__device__ inline void compute_roots(const float* m, float* roots)
{
  const float m00 = m[0], m01 = m[1], m02 = m[2], m11 = m[4], m12 = m[5], m22 = m[8];
  const float c0 = m00 * m11 * m22 + 2.f * m01 * m02 * m12 - m00 * m12 * m12 - m11 * m02 * m02 -
                   m22 * m01 * m01;
  const float c1 = m00 * m11 - m01 * m01 + m00 * m22 - m02 * m02 + m11 * m22 - m12 * m12;
  const float c2 = m00 + m11 + m22;
  if (fabsf(c0) < 1.1920928955078125e-7f) {
    compute_roots2(c2, c1, roots);
    return;
  }
  const float s_inv3 = (float)(1.0 / 3.0);
  const float s_sqrt3 = __builtin_sqrtf(3.0f);
  const float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.f) a_over_3 = 0.f;
  const float half_b = 0.5f * (c0 + c2_over_3 * (2.f * c2_over_3 * c2_over_3 - c1));
  float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.f) q = 0.f;
  const float rho = __builtin_sqrtf(-a_over_3);
  const float theta = hope_det_atan2f(__builtin_sqrtf(-q), half_b) * s_inv3;
  const float cos_theta = hope_det_cosf(theta);
  const float sin_theta = hope_det_sinf(theta);
  roots[0] = c2_over_3 + 2.f * rho * cos_theta;
  roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  if (roots[0] >= roots[1]) swapf(roots[0], roots[1]);
  if (roots[1] >= roots[2]) {
    swapf(roots[1], roots[2]);
    if (roots[0] >= roots[1]) swapf(roots[0], roots[1]);
  }
  if (roots[0] <= 0.f) compute_roots2(c2, c1, roots);
}

__device__ inline float sqnorm3(const float* v) { return v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]); }

__device__ inline void cross3(const float* a, const float* b, float* o)
{
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

__device__ inline void eigen33_smallest(const float* mat, float& eigenvalue, float* evec)
{
  float scale = 0.f;
#pragma unroll
  for (int i = 0; i < 9; ++i) scale = fmaxf(scale, fabsf(mat[i]));
  if (scale <= 1.17549435e-38f) scale = 1.f;
  float sm[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) sm[i] = mat[i] / scale;
  float ev[3];
  compute_roots(sm, ev);
  eigenvalue = ev[0] * scale;
  sm[0] -= ev[0];
  sm[4] -= ev[0];
  sm[8] -= ev[0];
  float v1[3], v2[3], v3[3];
  cross3(sm + 0, sm + 3, v1);
  cross3(sm + 0, sm + 6, v2);
  cross3(sm + 3, sm + 6, v3);
  const float l1 = sqnorm3(v1), l2 = sqnorm3(v2), l3 = sqnorm3(v3);
  float vx, vy, vz, l;
  if (l1 >= l2 && l1 >= l3) {
    vx = v1[0]; vy = v1[1]; vz = v1[2]; l = l1;
  } else if (l2 >= l1 && l2 >= l3) {
    vx = v2[0]; vy = v2[1]; vz = v2[2]; l = l2;
  } else {
    vx = v3[0]; vy = v3[1]; vz = v3[2]; l = l3;
  }
  const float s = __builtin_sqrtf(l);
  evec[0] = vx / s;
  evec[1] = vy / s;
  evec[2] = vz / s;
}

// ---- K7: |nz| > th_norm gate (utilities.cpp:629-645) + order-preserving compaction -----
static __global__ void __launch_bounds__(kBlock)
k_fit_flags(const float4* normals, unsigned n_vox, float th_norm, uint32_t* flags)
{
  const unsigned v = blockIdx.x * kBlock + threadIdx.x;
  if (v > n_vox) return;
  flags[v] = (v < n_vox && fabsf(normals[v].z) > th_norm) ? 1u : 0u;  // NaN fails
}

static __global__ void __launch_bounds__(kBlock)
k_fit_scatter(const uint32_t* flags, const uint32_t* scan, unsigned n_vox, const float4* vox, const float4* normals,
              const int* vox_frame, int* fit_rank, uint32_t* fit_idx, float4* fit_pt, float4* fit_nrm,
              int* fit_frame)
{
  const unsigned v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= n_vox) return;
  if (flags[v]) {
    const unsigned g = scan[v];
    fit_rank[v] = (int)g;
    fit_idx[g] = v;
    fit_pt[g] = vox[v];
    fit_nrm[g] = normals[v];
    fit_frame[g] = vox_frame[v];
  } else {
    fit_rank[v] = -1;
  }
}

static __global__ void k_frame_fit_ranges(FrameInfo* fi, int n_frames, const uint32_t* scan, Globals* g)
{
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_frames) return;
  const unsigned a = scan[fi[f].vox_start];
  const unsigned b = scan[fi[f].vox_start + fi[f].n_vox];
  fi[f].fit_start = (int)a;
  fi[f].n_fit = (int)(b - a);
  atomicMax(&g->max_fit, b - a);
}

// ---- K9: ZGrowing as min-rank reachability (z_growing.cpp:229-364) ----------------------
// The reference grows regions breadth-first from seeds taken in the order std::sort
// leaves n all-equal keys.  Because every finished region is closed under the (directed)
// neighbour relation, the segment of a point is the smallest seed rank among all points
// that can reach it; that fixed point is computed here by union-find over mutual edges
// plus monotone atomicMin sweeps over the one-way edges.
//
// libstdc++ introsort on all-equal keys is data independent: each level swaps first<->mid
// and reverses [first+1, last), then splits at first+1+(n-1)/2 while n > 16.  The element
// that ends at position p is found by walking the split chain down and undoing the
// levels bottom-up (tests/test_seed_order.py checks this against std::sort itself).
// rank_of walks the permutation forwards: the position at which element s of n ends up (its seed
// rank).  Top-down, one level at a time, so no per-level state is kept (no scratch memory).
__device__ __forceinline__ int rank_of(int n, int s)
{
  int f = 0, l = n, j = s;
  while (l - f > 16) {
    const int mid = f + (l - f) / 2;
    if (j == f) j = mid; else if (j == mid) j = f;
    if (j >= f + 1) j = l + f - j;
    const int cut = f + 1 + (l - f - 1) / 2;
    if (j >= cut) f = cut; else l = cut;
  }
  return j;
}

static __global__ void __launch_bounds__(kBlock)
k_seed_ranks(const FrameInfo* fi, const int* fit_frame, unsigned n_fit, int* rank, int* seed_node, int* parent,
             int* L)
{
  const unsigned g = blockIdx.x * kBlock + threadIdx.x;
  if (g >= n_fit) return;
  const FrameInfo& F = fi[fit_frame[g]];
  const int r = F.fit_start + rank_of(F.n_fit, (int)g - F.fit_start);
  rank[g] = r;       // global rank of node g
  seed_node[r] = (int)g;  // node seeded at global rank r
  parent[g] = (int)g;
  L[g] = 0x7fffffff;
}

__device__ __forceinline__ int uf_find(int* parent, int x)
{
  int p = parent[x];
  while (p != x) {
    const int gp = parent[p];
    if (gp != p) parent[x] = gp;  // path halving (benign race: only ever points nearer the root)
    x = p;
    p = gp;
  }
  return x;
}

// the roots of two nodes, the two chains walked side by side: one round trip per step for both
__device__ __forceinline__ void uf_find2(int* parent, int& a, int& b)
{
  int pa = parent[a], pb = parent[b];
  while (pa != a || pb != b) {
    const int ga = parent[pa], gb = parent[pb];  // (at a root: its own entry again)
    if (ga != pa) parent[a] = ga;                // path halving, as in uf_find
    if (gb != pb) parent[b] = gb;
    a = pa;
    pa = ga;
    b = pb;
    pb = gb;
  }
}

__device__ __forceinline__ bool z_edge(const float4* fit_pt, int a, int b, float th_z)
{
  return fabsf(fit_pt[a].z - fit_pt[b].z) < th_z;  // validatePoint, smooth mode
}

// Union over the mutual valid edges, two levels.  Fit points are numbered in VoxelGrid order
// (z layer, then y, then x), so the points of a horizontal surface sit in a few long index runs
// and most of their edges join two points of the same 1024-point block: those are united in an
// LDS union-find (no global atomics, no long pointer chases through HBM); what leaves the block
// is noted as a bit mask per point and united in global memory by k_union_cross.
constexpr int kUfBlock = 1024;
constexpr int kUfPer = 24;
constexpr int kUfTile = kUfBlock * kUfPer;  // 24576 points, 96 KiB of LDS: one workgroup per CU (the largest tile; the
                                            // launch chooses: a single cloud takes smaller tiles, one per CU and more)

__device__ __forceinline__ int uf_find_lds(int* parent, int x)
{
  int p = parent[x];
  while (p != x) {
    const int gp = parent[p];
    if (gp != p) parent[x] = gp;
    x = p;
    p = gp;
  }
  return x;
}

// bit j set: neighbour j of g is a mutual valid edge to a larger index (each pair once, from its
// smaller end).  K8: rows of the neighbour table are two 16-byte loads, all issued together.
template <bool K8>
__device__ __forceinline__ uint32_t mutual_edges_up(const int* knn, const float4* fit_pt, int g, int K, float th_z, int* nb)
{
  const float zg = fit_pt[g].z;
  uint32_t m = 0u;
  if (K8) {
    const int4* row = reinterpret_cast<const int4*>(knn + (size_t)g * 8);
    const int4 r0 = row[0], r1 = row[1];
    nb[0] = r0.x; nb[1] = r0.y; nb[2] = r0.z; nb[3] = r0.w;
    nb[4] = r1.x; nb[5] = r1.y; nb[6] = r1.z; nb[7] = r1.w;
    float zv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) zv[j] = nb[j] > g ? fit_pt[nb[j]].z : NAN;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (fabsf(zg - zv[j]) < th_z) m |= 1u << j;  // validatePoint, smooth mode (NaN: not an upward edge)
    int4 q0[8], q1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int4* rv = reinterpret_cast<const int4*>(knn + (size_t)(nb[j] > g ? nb[j] : g) * 8);
      q0[j] = rv[0];
      q1[j] = rv[1];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool has = q0[j].x == g || q0[j].y == g || q0[j].z == g || q0[j].w == g || q1[j].x == g || q1[j].y == g ||
                       q1[j].z == g || q1[j].w == g;
      if (!has) m &= ~(1u << j);
    }
  } else {
    for (int j = 0; j < K; ++j) {
      const int v = knn[(size_t)g * K + j];
      if (v <= g) continue;  // -1 (no neighbour), self, or the pair's larger end
      if (!(fabsf(zg - fit_pt[v].z) < th_z)) continue;
      bool mutual = false;
      for (int t = 0; t < K; ++t) mutual |= (knn[(size_t)v * K + t] == g);
      if (mutual) m |= 1u << j;
    }
  }
  return m;
}

// grid = (tiles per frame, n_frames): tiles start at the first fit point of a frame, so a frame of
// up to kUfTile fit points is united entirely in LDS.  (Fit points are in VoxelGrid order, z layer
// major: a cut through a frame separates layers of the same noisy surface, whose points interleave
// everywhere -- thousands of edges across the cut and fragmented components on both sides.)
template <bool K8>
__global__ void __launch_bounds__(kUfBlock)
k_union_local(const int* knn, const float4* fit_pt, const FrameInfo* fi, int K, float th_z, int* parent, uint32_t* cross,
              uint2* edges, unsigned* n_edges, const uint32_t* order, unsigned tile)
{
  // (dynamic: as many entries as the batch's largest tile has points, so that frames of a few thousand fit points
  // leave room for two workgroups per CU -- the kernel waits on gathers and LDS chases, not on issue)
  extern __shared__ int s_par[];
  const FrameInfo& F = fi[order ? order[blockIdx.y] : blockIdx.y];
  if (blockIdx.x * tile >= (unsigned)F.n_fit) return;
  const unsigned base = (unsigned)F.fit_start + blockIdx.x * tile;
  const unsigned end = min(base + tile, (unsigned)(F.fit_start + F.n_fit));
  for (unsigned i = threadIdx.x; i < end - base; i += kUfBlock) s_par[i] = (int)i;
  __syncthreads();
  for (int i = 0; i < kUfPer; ++i) {
    const int l = threadIdx.x + i * kUfBlock;
    const unsigned g = base + l;
    if (g >= end) break;
    int nb[8];  // K8 only (registers: every index is a compile-time constant)
    uint32_t m = mutual_edges_up<K8>(knn, fit_pt, (int)g, K, th_z, nb);
    uint32_t crossmask = 0u;
    while (m) {
      const int j = __ffs((int)m) - 1;
      m &= m - 1u;
      int v = 0;
      if (K8) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (t == j) v = nb[t];
      } else {
        v = knn[(size_t)g * K + j];
      }
      if ((unsigned)v >= end) {  // (v > g >= base)
        crossmask |= 1u << j;
        continue;
      }
      int a = l, b = v - (int)base;
      while (true) {
        a = uf_find_lds(s_par, a);  // (the two chains side by side, as in k_union_cross: 0.527 -> 0.547 ms)
        b = uf_find_lds(s_par, b);
        if (a == b) break;
        if (a > b) {
          const int t = a;
          a = b;
          b = t;
        }
        if (atomicCAS(&s_par[b], b, a) == b) break;  // hook the larger root under the smaller one
      }
    }
    cross[g] = crossmask;
  }
  __syncthreads();
  for (int i = 0; i < kUfPer; ++i) {
    const int l = threadIdx.x + i * kUfBlock;
    const unsigned g = base + l;
    if (g >= end) break;
    int r = l;
    while (s_par[r] != r) r = s_par[r];
    parent[g] = (int)base + r;
    s_par[l] = r;  // (a walker that meets this entry jumps straight to its root)
  }
  // A frame that fits one tile has its final components here, in LDS: the valid one-way edges
  // between two different components (all the label sweeps need) are found with LDS look-ups of the
  // roots instead of k_cross_edges' gathers.
  if (!K8 || (unsigned)F.n_fit > tile) return;
  __syncthreads();
  const int lane = threadIdx.x & (kWave - 1);
  for (int i = 0; i < kUfPer; ++i) {  // (uniform trip count: the loop holds wave-wide votes)
    const int l = threadIdx.x + i * kUfBlock;
    const unsigned g = base + l;
    const bool live = g < end;
    if (__builtin_amdgcn_ballot_w64(live) == 0ull) break;
    const int r = live ? s_par[l] : 0;
    const float zg = live ? fit_pt[g].z : 0.f;
    int nb[8], rv[8];
    float zv[8];
    const int4* row = reinterpret_cast<const int4*>(knn + (size_t)(live ? g : base) * 8);
    const int4 r0 = row[0], r1 = row[1];
    nb[0] = r0.x; nb[1] = r0.y; nb[2] = r0.z; nb[3] = r0.w;
    nb[4] = r1.x; nb[5] = r1.y; nb[6] = r1.z; nb[7] = r1.w;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool use = live && nb[j] >= 0 && nb[j] != (int)g;
      rv[j] = use ? s_par[nb[j] - (int)base] : r;
      zv[j] = (use && rv[j] != r) ? fit_pt[nb[j]].z : NAN;
    }
    bool emit[8];
    unsigned long long ems[8];
    unsigned total = 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      emit[j] = rv[j] != r && fabsf(zg - zv[j]) < th_z;
      ems[j] = __builtin_amdgcn_ballot_w64(emit[j]);
      total += (unsigned)__popcll(ems[j]);
    }
    if (total == 0u) continue;
    unsigned o = 0u;  // (one place in the list for the edges of all eight columns)
    if (lane == 0) o = atomicAdd(n_edges, total);
    o = (unsigned)__builtin_amdgcn_readfirstlane((int)o);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (emit[j])
        edges[o + (unsigned)__popcll(ems[j] & ((1ull << lane) - 1ull))] =
            make_uint2(base + (unsigned)r, base + (unsigned)rv[j]);
      o += (unsigned)__popcll(ems[j]);
    }
  }
}

constexpr int kKeyVotes = 3;  // keys of a chunk that are gathered by a vote each; the lanes left after these go alone

// The mutual edges that leave a tile.  k_union_local has left parent[] flattened per tile, so an
// edge (g, v) only says "unite tile component parent[g] with tile component parent[v]"; the many
// edges along the cut of one surface all name the same pair, so each wave unites a pair once.
static __global__ void __launch_bounds__(kBlock)
k_union_cross(const int* knn, const uint32_t* cross, unsigned n_fit, int K, int* parent, int ablate)
{
  const unsigned g = xcd_block() * kBlock + threadIdx.x;
  const uint32_t m = g < n_fit ? cross[g] : 0u;
  if (__builtin_amdgcn_ballot_w64(m != 0u) == 0ull || ablate == 2) return;
  const int lane = threadIdx.x & (kWave - 1);
  const int ra = m ? parent[g] : 0;
  for (int j = 0; j < K; ++j) {
    const bool on = (m >> j) & 1u;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(on);
    if (!todo) continue;
    const int rb = on ? parent[knn[(size_t)g * K + j]] : 0;
    bool first = false;  // the first lane of each distinct pair: those unite, all at once (the finds are chains of loads)
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int la = __builtin_amdgcn_readlane(ra, leader), lb = __builtin_amdgcn_readlane(rb, leader);
      todo &= ~__builtin_amdgcn_ballot_w64(on && ra == la && rb == lb);
      first |= lane == leader;
    }
    if (first && ra != rb && ablate != 1) {
      int a = ra, b = rb;
      while (true) {
        uf_find2(parent, a, b);
        if (a == b) break;
        if (a > b) {
          const int t = a;
          a = b;
          b = t;
        }
        if (atomicCAS(&parent[b], b, a) == b) break;
      }
    }
  }
}

// One atomic per distinct key per wave: neighbouring points mostly share a root / label, so
// thousands of same-address atomics collapse to a handful.
__device__ __forceinline__ void wave_atomic_min_by_key(int* table, int key, int val, bool active)
{
  unsigned long long todo = __builtin_amdgcn_ballot_w64(active);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int k = __shfl(key, leader);
    const bool mine = active && key == k;
    const unsigned long long grp = __builtin_amdgcn_ballot_w64(mine);
    int m = mine ? val : 0x7fffffff;
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) m = min(m, __shfl_xor(m, off));
    if ((threadIdx.x & (kWave - 1)) == leader) atomicMin(&table[k], m);
    todo &= ~grp;
  }
}

__device__ __forceinline__ void wave_atomic_count_by_key(uint32_t* table, int key, bool active)
{
  unsigned long long todo = __builtin_amdgcn_ballot_w64(active);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int k = __shfl(key, leader);
    const unsigned long long grp = __builtin_amdgcn_ballot_w64(active && key == k);
    if ((threadIdx.x & (kWave - 1)) == leader) atomicAdd(&table[k], (uint32_t)__popcll(grp));
    todo &= ~grp;
  }
}

// A wave takes kSizeRun consecutive chunks of 64 points: the chains of the run are chased together, and a minimum goes
// to L[] when the root changes, not per chunk -- neighbouring points share their component, and the components of a
// single cloud are few and large (one address for the atomics of every wave of the floor).
constexpr int kSizeRun = 8;
static __global__ void __launch_bounds__(kBlock)
k_flatten_roots(int* parent, const int* rank, unsigned n_fit, int* root, int* L)
{
  const int lane = threadIdx.x & (kWave - 1);
  const unsigned wave = xcd_block() * (kBlock / kWave) + threadIdx.x / kWave;
  if (wave * (unsigned)(kWave * kSizeRun) >= n_fit) return;
  const unsigned g0 = wave * (unsigned)(kWave * kSizeRun) + (unsigned)lane;
  int r[kSizeRun], rk[kSizeRun];
#pragma unroll
  for (int i = 0; i < kSizeRun; ++i) {
    const unsigned g = g0 + i * kWave;
    r[i] = g < n_fit ? parent[g] : -1;
    rk[i] = g < n_fit ? rank[g] : 0x7fffffff;
  }
  bool more = true;
  while (more) {
    more = false;
    int p[kSizeRun];
#pragma unroll
    for (int i = 0; i < kSizeRun; ++i) p[i] = r[i] >= 0 ? parent[r[i]] : -1;
#pragma unroll
    for (int i = 0; i < kSizeRun; ++i) {
      more |= p[i] != r[i];
      r[i] = p[i];
    }
  }
  int run_key = -1, run_min = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < kSizeRun; ++i) {
    const bool active = r[i] >= 0;
    if (active) root[g0 + i * kWave] = r[i];
    unsigned long long todo = __builtin_amdgcn_ballot_w64(active);
    for (int vote = 0; todo && vote < kKeyVotes; ++vote) {
      const int k = __builtin_amdgcn_readlane(r[i], __ffsll((long long)todo) - 1);
      const bool mine = active && r[i] == k;
      int m = mine ? rk[i] : 0x7fffffff;
#pragma unroll
      for (int off = kWave / 2; off > 0; off >>= 1) m = min(m, __shfl_xor(m, off));
      if (k != run_key) {
        if (run_key >= 0 && lane == 0) atomicMin(&L[run_key], run_min);
        run_key = k;
        run_min = 0x7fffffff;
      }
      run_min = min(run_min, m);
      todo &= ~__builtin_amdgcn_ballot_w64(mine);
    }
    if ((todo >> lane) & 1ull) atomicMin(&L[r[i]], rk[i]);  // (clutter: as many components as points, no two alike)
  }
  if (run_key >= 0 && lane == 0) atomicMin(&L[run_key], run_min);
}

// The valid directed edges that join two DIFFERENT mutual components are the only ones the label
// sweeps need; on real scenes they are a few percent of all edges (region fringes).  Collected
// once (order is irrelevant: atomicMin is commutative), then swept until nothing changes.
template <bool K8>
__global__ void __launch_bounds__(kBlock)
k_cross_edges(const int* knn, const float4* fit_pt, const int* root, const int* fit_frame, const FrameInfo* fi,
              unsigned n_fit, int K, float th_z, uint2* edges, unsigned* n_edges, unsigned tile)
{
  const unsigned g = xcd_block() * kBlock + threadIdx.x;
  // (K8: the edges of a frame that fits one union tile were found by k_union_local)
  const bool live = g < n_fit && !(K8 && (unsigned)fi[fit_frame[g]].n_fit <= tile);
  if (__builtin_amdgcn_ballot_w64(live) == 0ull) return;
  const int r = live ? root[g] : 0;
  const float zg = live ? fit_pt[g].z : 0.f;
  const int lane = threadIdx.x & (kWave - 1);
  auto emit_edge = [&](bool emit, int rv) {
    const unsigned long long em = __builtin_amdgcn_ballot_w64(emit);
    if (em) {
      unsigned o = 0;
      if (lane == __ffsll((long long)em) - 1) o = atomicAdd(n_edges, (unsigned)__popcll(em));
      o = (unsigned)__shfl((int)o, __ffsll((long long)em) - 1);
      if (emit) edges[o + (unsigned)__popcll(em & ((1ull << lane) - 1ull))] = make_uint2((unsigned)r, (unsigned)rv);
    }
  };
  if (K8) {  // the row in two 16-byte loads, all sixteen gathers in flight together
    int nb[8], rv[8];
    float zv[8];
    const int4* row = reinterpret_cast<const int4*>(knn + (size_t)(live ? g : 0u) * 8);
    const int4 r0 = row[0], r1 = row[1];
    nb[0] = r0.x; nb[1] = r0.y; nb[2] = r0.z; nb[3] = r0.w;
    nb[4] = r1.x; nb[5] = r1.y; nb[6] = r1.z; nb[7] = r1.w;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool use = live && nb[j] >= 0 && nb[j] != (int)g;
      rv[j] = use ? root[nb[j]] : r;
      zv[j] = use ? fit_pt[nb[j]].z : NAN;
    }
    bool em[8];
    bool any = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      em[j] = rv[j] != r && fabsf(zg - zv[j]) < th_z;
      any |= em[j];
    }
    if (__builtin_amdgcn_ballot_w64(any) == 0ull) return;
    // one place in the list for the wave's edges of all eight columns (the counter is one address for every wave)
    unsigned long long ems[8];
    unsigned total = 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ems[j] = __builtin_amdgcn_ballot_w64(em[j]);
      total += (unsigned)__popcll(ems[j]);
    }
    unsigned o = 0u;
    if (lane == 0) o = atomicAdd(n_edges, total);
    o = (unsigned)__builtin_amdgcn_readfirstlane((int)o);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (em[j]) edges[o + (unsigned)__popcll(ems[j] & ((1ull << lane) - 1ull))] = make_uint2((unsigned)r, (unsigned)rv[j]);
      o += (unsigned)__popcll(ems[j]);
    }
  } else {
    for (int j = 0; j < K; ++j) {
      bool emit = false;
      int rv = 0;
      if (live) {
        const int v = knn[(size_t)g * K + j];
        if (v >= 0 && v != (int)g) {
          rv = root[v];
          emit = rv != r && fabsf(zg - fit_pt[v].z) < th_z;
        }
      }
      emit_edge(emit, rv);
    }
  }
}

// one sweep over the component-level edges (source root, target root)
static __global__ void __launch_bounds__(kBlock)
k_label_sweep_edges(const uint2* edges, const unsigned* n_edges, int* L, unsigned* changed)
{
  const unsigned n = *n_edges;
  bool ch = false;
  for (unsigned e = blockIdx.x * kBlock + threadIdx.x; e < n; e += gridDim.x * kBlock) {
    const uint2 ed = edges[e];
    const int a = L[ed.x];
    if (a < L[ed.y]) {
      const int old = atomicMin(&L[ed.y], a);
      ch |= (a < old);
    }
  }
  if (ch) *changed = 1u;
}

// shortcut: the seed that labels a component may itself carry a smaller label
static __global__ void __launch_bounds__(kBlock)
k_label_shortcut(const int* root, const int* seed_node, unsigned n_fit, int* L, unsigned* changed)
{
  const unsigned g = blockIdx.x * kBlock + threadIdx.x;
  if (g >= n_fit || root[g] != (int)g) return;
  const int a = L[g];
  const int aw = L[root[seed_node[a]]];
  if (aw < a) {
    atomicMin(&L[g], aw);
    *changed = 1u;
  }
}

// per rank value: number of points whose segment was seeded at that rank.  A wave takes kSizeRun consecutive chunks of
// 64 points and adds a count when the label changes, not per chunk: neighbouring points share their segment, and the
// segments of a single cloud are few and large (one address for the adds of every wave of the floor).
static __global__ void __launch_bounds__(kBlock)
k_segment_sizes(const int* root, const int* L, unsigned n_fit, int* label, uint32_t* cnt)
{
  const int lane = threadIdx.x & (kWave - 1);
  const unsigned wave = xcd_block() * (kBlock / kWave) + threadIdx.x / kWave;
  const unsigned g0 = wave * (unsigned)(kWave * kSizeRun) + (unsigned)lane;
  if (wave * (unsigned)(kWave * kSizeRun) >= n_fit) return;
  int a[kSizeRun];
#pragma unroll
  for (int i = 0; i < kSizeRun; ++i) a[i] = g0 + i * kWave < n_fit ? root[g0 + i * kWave] : -1;
#pragma unroll
  for (int i = 0; i < kSizeRun; ++i) a[i] = a[i] >= 0 ? L[a[i]] : -1;
  int run_key = -1;
  uint32_t run_cnt = 0u;
#pragma unroll
  for (int i = 0; i < kSizeRun; ++i) {
    const bool active = a[i] >= 0;  // (labels are seed ranks, >= 0)
    if (active) label[g0 + i * kWave] = a[i];
    unsigned long long todo = __builtin_amdgcn_ballot_w64(active);
    for (int vote = 0; todo && vote < kKeyVotes; ++vote) {
      const int k = __builtin_amdgcn_readlane(a[i], __ffsll((long long)todo) - 1);
      const unsigned long long grp = __builtin_amdgcn_ballot_w64(active && a[i] == k);
      if (k != run_key) {
        if (run_cnt && lane == 0) atomicAdd(&cnt[run_key], run_cnt);
        run_key = k;
        run_cnt = 0u;
      }
      run_cnt += (uint32_t)__popcll(grp);
      todo &= ~grp;
    }
    if ((todo >> lane) & 1ull) atomicAdd(&cnt[a[i]], 1u);
  }
  if (run_cnt && lane == 0) atomicAdd(&cnt[run_key], run_cnt);
}

static __global__ void __launch_bounds__(kBlock)
k_segment_flags(const uint32_t* cnt, unsigned n_fit, int min_cluster, U3* flags)
{
  // per seed rank: (.a is a segment, .b is a cluster, .c points of the cluster); one scan for the three
  const unsigned r = blockIdx.x * kBlock + threadIdx.x;
  if (r > n_fit) return;
  const uint32_t c = r < n_fit ? cnt[r] : 0u;
  const bool cl = c >= (uint32_t)min_cluster;  // and <= INT_MAX, z_growing.cpp:153-154
  flags[r] = U3{c > 0u ? 1u : 0u, cl ? 1u : 0u, cl ? c : 0u};
}

// The kernels that run one workgroup per frame (k_voxel_rank, k_column_index, k_union_local, k_members_by_frame) take
// 4 frames per CU on a 1024-frame batch whose frames differ 5x in size: dealt in frame order, a large frame that comes
// up last is a tail as long as itself.  order[] = the frames by descending key (key: a count per frame at a stride,
// e.g. its group records), ties by frame number; n^2 compares, a wave per frame (n_frames <= 4096).
static __global__ void __launch_bounds__(kBlock)
k_frame_order(const uint32_t* key, unsigned stride, int n_frames, uint32_t* order)
{
  // one wave per frame: its place = the frames with a larger key, or the same key and a smaller number
  const int f = (int)(blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave);
  const int lane = threadIdx.x & (kWave - 1);
  if (f >= n_frames) return;
  const unsigned k = key[(size_t)f * stride];
  unsigned before = 0u;
  for (int o = lane; o < n_frames; o += kWave) {
    const unsigned ko = key[(size_t)o * stride];
    before += (ko > k || (ko == k && o < f)) ? 1u : 0u;
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) before += (unsigned)__shfl_down((int)before, off);
  if (lane == 0) order[before] = (uint32_t)f;
}

static __global__ void k_frame_cluster_ranges(FrameInfo* fi, int n_frames, const U3* scan, Globals* g)
{
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_frames) return;
  const U3 a = scan[fi[f].fit_start], b = scan[fi[f].fit_start + fi[f].n_fit];
  fi[f].seg_start = (int)a.a;
  fi[f].n_segments = (int)(b.a - a.a);
  fi[f].cl_start = (int)a.b;
  fi[f].n_clusters = (int)(b.b - a.b);
  atomicMax(&g->max_frame_clusters, b.b - a.b);
}

// cluster id of every fit point (-1: its segment was dropped by the size filter) and the
// sort key that groups members by cluster
static __global__ void __launch_bounds__(kBlock)
k_point_clusters(const int* label, const U3* flags, const U3* scan, unsigned n_fit, unsigned n_cl_total, int* cid,
                 uint32_t* mkey, uint32_t* mval)
{
  const unsigned g = blockIdx.x * kBlock + threadIdx.x;
  if (g >= n_fit) return;
  const int a = label[g];
  const int c = flags[a].b ? (int)scan[a].b : -1;
  cid[g] = c;
  if (mkey) {  // (null: the member lists are made frame by frame, no pairs to sort)
    mkey[g] = c >= 0 ? (uint32_t)c : n_cl_total;
    mval[g] = g;
  }
}

// The member lists of a batch of frames without a sort: a cluster's members are the fit points of ONE frame in
// ascending index, so one workgroup walks its frame's fit points, each of its four waves a quarter of them, 64 at a
// time, and hands every point the next place of its cluster (the lanes of one step that share a cluster are ranked by
// a vote; a first walk counts every cluster's members per quarter, which gives the quarters their first places).
// Replaces the stable radix sort of (cluster, index) pairs when no frame has more than kMemberClusters clusters.
// grid = n_frames, block = kBlock; bits: enough for the largest per-frame cluster count - 1.
constexpr int kMemberClusters = 4096;
template <bool PLACE>
__device__ __forceinline__ void members_walk(const int* cid, unsigned first, unsigned end, int cl_start, int bits,
                                             int lane, unsigned* next, uint32_t* members)
{
  // (four steps' cluster ids are fetched while the four before them are handled: a step is otherwise one round trip
  // to memory)
  constexpr int kU = 4;
  const unsigned long long lt = (1ull << lane) - 1ull;
  int c_next[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const unsigned i = first + (unsigned)(u * kWave + lane);
    c_next[u] = i < end ? cid[i] : -1;
  }
  for (unsigned i0 = first; i0 < end; i0 += kU * kWave) {
    int cc[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      cc[u] = c_next[u];
      const unsigned i = i0 + (unsigned)((kU + u) * kWave + lane);
      c_next[u] = i < end ? cid[i] : -1;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const unsigned g = i0 + (unsigned)(u * kWave + lane);
      const int c = cc[u];
      const bool ok = c >= 0;
      const unsigned local = ok ? (unsigned)(c - cl_start) : 0u;
      const unsigned long long same = match_lanes(local, bits, ok);
      const unsigned at = ok ? next[local] : 0u;
      if (PLACE && ok) members[at + (unsigned)__popcll(same & lt)] = g;
      // (LDS runs a wave's instructions in order, and only this wave touches its row of counters; the barriers keep
      // the compiler from reordering the read and the write)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (ok && (same & lt) == 0ull) next[local] = at + (unsigned)__popcll(same);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

static __global__ void __launch_bounds__(kBlock)
k_members_by_frame(const FrameInfo* fi, const int* cid, const uint32_t* cl_first, int bits, uint32_t* members,
                   const uint32_t* order)
{
  constexpr int kW = kBlock / kWave;
  __shared__ unsigned s_next[kW][kMemberClusters];
  const FrameInfo& F = fi[order ? order[blockIdx.x] : blockIdx.x];
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  const unsigned ncl = (unsigned)F.n_clusters;
  if (ncl == 0u) return;
  for (unsigned c = threadIdx.x; c < ncl; c += kBlock)
    for (int ww = 0; ww < kW; ++ww) s_next[ww][c] = 0u;
  __syncthreads();
  const unsigned base = (unsigned)F.fit_start, n = (unsigned)F.n_fit;
  const unsigned per = ((n + kW - 1u) / kW + kWave - 1u) / kWave * kWave;  // a quarter, in whole steps
  const unsigned first = min(base + (unsigned)w * per, base + n), end = min(first + per, base + n);
  members_walk<false>(cid, first, end, F.cl_start, bits, lane, s_next[w], members);
  __syncthreads();
  for (unsigned c = threadIdx.x; c < ncl; c += kBlock) {
    unsigned run = cl_first[(unsigned)F.cl_start + c];
    for (int ww = 0; ww < kW; ++ww) {
      const unsigned t = s_next[ww][c];
      s_next[ww][c] = run;
      run += t;
    }
  }
  __syncthreads();
  members_walk<true>(cid, first, end, F.cl_start, bits, lane, s_next[w], members);
}

struct ClusterRec {  // == hope_cluster
  int size;
  int pass;
  float z_mean;
  float min_x, min_y, max_x, max_y;
  int seed_rank;
};

static __global__ void __launch_bounds__(kBlock)
k_cluster_heads(const uint32_t* cnt, const U3* flags, const U3* scan, const int* seed_node, const int* fit_frame,
                const FrameInfo* fi, unsigned n_fit, uint32_t* cl_first, ClusterRec* rec)
{
  const unsigned r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= n_fit || !flags[r].b) return;
  const unsigned c = scan[r].b;
  cl_first[c] = scan[r].c;
  rec[c].size = (int)cnt[r];
  rec[c].seed_rank = (int)r - fi[fit_frame[seed_node[r]]].fit_start;
}

// Utilities::normalAnalysis's 2x2 principal axis (closed form; see oracle header note)
__device__ inline void principal_axis_2x2(float a, float b, float d, float& ax, float& ay)
{
  const float p = 0.5f * (a - d);
  const float z = __builtin_sqrtf(p * p + b * b);
  if (z == 0.f) {
    ax = 0.f;
    ay = 1.f;
    return;
  }
  float vx, vy;
  if (p >= 0.f) {
    vx = p + z;
    vy = b;
  } else {
    vx = b;
    vy = z - p;
  }
  const float nrm = __builtin_sqrtf(vx * vx + vy * vy);
  ax = vx / nrm;
  ay = vy / nrm;
}

// The scalar steps of Utilities::normalAnalysis between its sums (utilities.cpp:151-255), shared by the two
// kernels that evaluate the sums (k_cluster_verify_b: chains; verify_big.hpp: exact integer blocks).
// (i)-(ii): mean vector and its tilt; false = the cluster fails here
__device__ __forceinline__ bool verify_mean_gate(int sz, float s0, float s1, float s2, float th_angle, float& m0,
                                                 float& m1)
{
  if (sz <= 2) return false;
  const float fsz = (float)sz;
  m0 = s0 / fsz;
  m1 = s1 / fsz;
  const float m2 = s2 / fsz;
  const float proj_norm = __builtin_sqrtf(m0 * m0 + m1 * m1);
  const float mean_norm = __builtin_sqrtf(m0 * m0 + (m1 * m1 + m2 * m2));
  const float grad = hope_det_asinf(proj_norm / mean_norm);
  return !(grad > th_angle);
}
// (iii): principal axis of the normals' xy covariance
__device__ __forceinline__ void verify_axis(int sz, float sxx, float sxy, float syy, float& ax, float& ay)
{
  const float denom = (float)(sz - 1);
  principal_axis_2x2(sxx / denom, sxy / denom, syy / denom, ax, ay);
}
// (iv)-(v): the two halves' mean normals and the angle between them
__device__ __forceinline__ int verify_split(const float* s6, unsigned na, unsigned nb, float th_angle)
{
  if (na == 0 || nb == 0) return 0;
  const float fa = (float)na, fb = (float)nb;
  const float p1[3] = {s6[0] / fa, s6[1] / fa, s6[2] / fa};
  const float p2[3] = {s6[3] / fb, s6[4] / fb, s6[5] / fb};
  const float mu = p2[0] * p1[0] + (p2[1] * p1[1] + p2[2] * p1[2]);
  const float n1 = __builtin_sqrtf(sqnorm3(p1));
  const float n2 = __builtin_sqrtf(sqnorm3(p2));
  const float rad_mu = hope_det_acosf(mu / (n1 * n2));
  return (rad_mu <= th_angle) ? 1 : 0;
}

// ---- K10: A10 mean z (utilities.cpp:751-769), A12 bbox (plane_segment.cpp:408-420),
// A11 normalAnalysis (utilities.cpp:151-282).  Every sum is the reference's sequential float sum
// over the members in ascending index.
// One block per cluster.  The reference's sums are
// chains of dependent float adds in member order; a chain cannot be split, but the work around it
// can: waves 1..3 of the block are LOADERS -- they fetch the members' points / normals (coalesced
// index loads one tile ahead, 16-byte gathers) and park the per-point terms in LDS, double
// buffered -- while wave 0 only runs the chains, one lane per accumulator, over the previous tile
// with 16-byte LDS reads.  A chain then costs about one dependent add (4 cycles) per member
// instead of a memory round trip, which matters for the one huge cluster (a floor) that bounds
// the kernel.  Terms that the reference skips (a non-finite z, the other side of the split)
// enter as +0.0f, which leaves a running sum that started at +0.0f bit-identical.
constexpr int kVTile = 384;                    // members per tile ...
constexpr int kVTileBig = 1152;                // ... and for the clusters of kVBigFrom members and more: a tile costs
constexpr int kVBigFrom = 32768;               // ~0.25 us of barrier and pipeline fill on top of its adds (measured)
constexpr int kVLoaders = kBlock - kWave;      // 192 loader threads (2 or 6 members each per tile)

// acc + sv[0] + sv[1] + ... in that order.  The adds are one dependent chain (4 cycles each); the LDS
// reads that feed it are issued a whole block (32 values, 8 x 16 bytes) ahead, so the chain never waits
// for a read it has just issued: a plain "read 16, add 16" loop costs a read latency per block on top of
// the adds (measured 4.2 ns per add against the 1.7 ns of the add itself).
__device__ __forceinline__ void chain_load(const float* p, float4 (&r)[8], float& acc)
{
#pragma unroll
  for (int q = 0; q < 8; ++q) r[q] = *reinterpret_cast<const float4*>(p + 4 * q);
  // the reads are issued in front of the adds they overlap with: the running sum passes through this
  // statement, so no add moves above it, and no read moves below it
  asm volatile("" : "+v"(acc) : : "memory");
}
__device__ __forceinline__ float chain_add(const float4 (&r)[8], float acc)
{
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    acc += r[q].x; acc += r[q].y; acc += r[q].z; acc += r[q].w;
  }
  return acc;
}
__device__ __forceinline__ float chain_sum(const float* sv, int n, float acc)
{
  constexpr int kB = 32;
  int j = 0;
  if (n >= kB) {
    float4 ra[8], rb[8];  // two register sets, no copies: while one is added the other is in flight
    chain_load(sv, ra, acc);
    while (true) {
      // ra holds [j, j + kB)
      if (j + 2 * kB > n) {
        acc = chain_add(ra, acc);
        j += kB;
        break;
      }
      chain_load(sv + j + kB, rb, acc);
      acc = chain_add(ra, acc);
      j += kB;
      if (j + 2 * kB > n) {
        acc = chain_add(rb, acc);
        j += kB;
        break;
      }
      chain_load(sv + j + kB, ra, acc);
      acc = chain_add(rb, acc);
      j += kB;
    }
  }
  for (; j + 4 <= n; j += 4) {
    const float4 a = *reinterpret_cast<const float4*>(sv + j);
    acc += a.x; acc += a.y; acc += a.z; acc += a.w;
  }
  for (; j < n; ++j) acc += sv[j];
  return acc;
}

// One pass over the members: loaders call term(point, normal, slot) for every member of a tile,
// wave 0 sums rows 0..n_chains-1 of the tile in member order.  Returns this lane's chain sum.
template <int kVTile, bool NEED_POINT, class Term>
__device__ __forceinline__ float verify_pass(const uint32_t* m, int sz, const float4* fit_pt, const float4* fit_nrm,
                                             float (*s_v)[6][kVTile + 4], int n_chains, Term&& term)
{
  constexpr int kVPer = kVTile / kVLoaders;
  static_assert(kVTile % kVLoaders == 0 && kVTile % 32 == 0, "whole loader rounds, whole chain blocks");
  const int tid = threadIdx.x;
  const bool loader = tid >= kWave;
  const int lt = tid - kWave;
  const int lane = tid & (kWave - 1);
  const int n_tiles = (sz + kVTile - 1) / kVTile;
  float acc = 0.f;
  unsigned idx[kVPer];
  float4 rp[kVPer], rn[kVPer];
  auto fetch_idx = [&](int ti) {
#pragma unroll
    for (int i = 0; i < kVPer; ++i) {
      const int e = ti * kVTile + i * kVLoaders + lt;
      idx[i] = e < sz ? m[e] : 0u;
    }
  };
  auto fetch_data = [&]() {
#pragma unroll
    for (int i = 0; i < kVPer; ++i) {
      if (NEED_POINT) rp[i] = fit_pt[idx[i]];
      rn[i] = fit_nrm[idx[i]];
    }
  };
  __syncthreads();
  if (loader) {
    fetch_idx(0);
    fetch_data();
    if (n_tiles > 1) fetch_idx(1);
  }
  for (int ti = 0; ti < n_tiles; ++ti) {
    const int buf = ti & 1;
    const int tile = min(kVTile, sz - ti * kVTile);
    if (loader) {
#pragma unroll
      for (int i = 0; i < kVPer; ++i) {
        const int t = i * kVLoaders + lt;
        if (t < tile) term(rp[i], rn[i], &s_v[buf][0][t]);
      }
    }
    __syncthreads();
    if (loader) {
      if (ti + 1 < n_tiles) {
        fetch_data();                              // tile ti+1, indices already here
        if (ti + 2 < n_tiles) fetch_idx(ti + 2);
      }
    } else if (lane < n_chains) {
      acc = chain_sum(s_v[buf][lane], tile, acc);
    }
  }
  return acc;
}

__device__ __forceinline__ float block_reduce_minmax(float v, bool is_min, float* s_red)
{
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    const float o = __shfl_xor(v, off);
    v = is_min ? fminf(v, o) : fmaxf(v, o);
  }
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) s_red[threadIdx.x / kWave] = v;
  __syncthreads();
  float r = s_red[0];
  for (int w = 1; w < kBlock / kWave; ++w) r = is_min ? fminf(r, s_red[w]) : fmaxf(r, s_red[w]);
  return r;
}

__device__ __forceinline__ unsigned block_sum_u32(unsigned v, unsigned* s_cnt)
{
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += (unsigned)__shfl_xor((int)v, off);
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) s_cnt[threadIdx.x / kWave] = v;
  __syncthreads();
  unsigned r = 0;
  for (int w = 0; w < kBlock / kWave; ++w) r += s_cnt[w];
  return r;
}

// The kernel below takes as long as its longest chain (a frame's floor: 10-20 000 members, three dependent passes of
// 2.74 ns per member) -- when that chain starts at once.  In cluster order it starts whenever its frame comes up, so
// the clusters are dealt to the workgroups long ones first: clusters of kVLong members and more from the front of
// `order`, the others from the back (ends[0], ends[1]: zeroed by the host).
constexpr int kVLong = 2048;
static __global__ void __launch_bounds__(kBlock)
k_cluster_order(const ClusterRec* rec, unsigned n_cl, unsigned* ends, uint32_t* order)
{
  const unsigned c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= n_cl) return;
  const bool is_long = rec[c].size >= kVLong;
  const unsigned pos = is_long ? atomicAdd(&ends[0], 1u) : n_cl - 1u - atomicAdd(&ends[1], 1u);
  order[pos] = c;
}

template <int kVTile>
__global__ void __launch_bounds__(kBlock)
k_cluster_verify_b(const uint32_t* members, const uint32_t* cl_first, const float4* fit_pt, const float4* fit_nrm,
                   unsigned n_cl, float th_angle, int size_from, int size_below, ClusterRec* rec, const uint32_t* order)
{
  constexpr int kVStride = kVTile + 4;  // LDS row stride: the 6 chain lanes hit different banks
  __shared__ __attribute__((aligned(16))) float s_v[2][6][kVStride];
  __shared__ float s_bc[8];
  __shared__ float s_red[kBlock / kWave];
  __shared__ unsigned s_cnt[kBlock / kWave];
  if (blockIdx.x >= n_cl) return;
  const unsigned c = order[blockIdx.x];  // (k_cluster_order: the long chains start first)
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const bool chain_wave = tid < kWave;
  const uint32_t* m = members + cl_first[c];
  const int sz = rec[c].size;
  if (sz < size_from || sz >= size_below) return;  // (the launch with the other tile size has this cluster)

  // ---- pass 1: sum z (lane 0), sum nx, ny, |nz| (lanes 1..3); bbox + finite count by the loaders --
  unsigned ct = 0;
  float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
  float acc = verify_pass<kVTile, true>(m, sz, fit_pt, fit_nrm, s_v, 4, [&](const float4& p, const float4& nn, float* slot) {
    const bool fin = isfinite(p.z);
    slot[0 * kVStride] = fin ? p.z : 0.f;
    slot[1 * kVStride] = nn.x;
    slot[2 * kVStride] = nn.y;
    slot[3 * kVStride] = fabsf(nn.z);
    ct += fin ? 1u : 0u;
    mnx = fminf(mnx, p.x);
    mny = fminf(mny, p.y);
    mxx = fmaxf(mxx, p.x);
    mxy = fmaxf(mxy, p.y);
  });
  mnx = block_reduce_minmax(mnx, true, s_red);
  mny = block_reduce_minmax(mny, true, s_red);
  mxx = block_reduce_minmax(mxx, false, s_red);
  mxy = block_reduce_minmax(mxy, false, s_red);
  const unsigned ctz = block_sum_u32(ct, s_cnt);
  __syncthreads();
  if (chain_wave && lane < 4) s_bc[lane] = acc;
  __syncthreads();
  const float mid = s_bc[0], s0 = s_bc[1], s1 = s_bc[2], s2 = s_bc[3];

  int pass = 0;
  float m0 = 0.f, m1 = 0.f;
  if (verify_mean_gate(sz, s0, s1, s2, th_angle, m0, m1)) {  // block-uniform
    // ---- pass 2: covariance sums cxx, cxy, cyy (lanes 0..2) ---------------------------------
    const float cacc =
        verify_pass<kVTile, false>(m, sz, fit_pt, fit_nrm, s_v, 3, [&](const float4&, const float4& nn, float* slot) {
          const float tx = nn.x - m0, ty = nn.y - m1;
          slot[0 * kVStride] = tx * tx;
          slot[1 * kVStride] = tx * ty;
          slot[2 * kVStride] = ty * ty;
        });
    __syncthreads();
    if (chain_wave && lane < 3) s_bc[lane] = cacc;
    __syncthreads();
    float ax, ay;
    verify_axis(sz, s_bc[0], s_bc[1], s_bc[2], ax, ay);
    // ---- pass 3: split sums; lanes 0..2 own part 1 (nx, ny, |nz|), lanes 3..5 part 2 ----------
    unsigned na = 0;
    const float pacc =
        verify_pass<kVTile, false>(m, sz, fit_pt, fit_nrm, s_v, 6, [&](const float4&, const float4& nn, float* slot) {
          const float tx = nn.x - m0, ty = nn.y - m1;
          const bool side = (ax * tx + ay * ty > 0.f);
          const float nz = fabsf(nn.z);
          slot[0 * kVStride] = side ? nn.x : 0.f;
          slot[1 * kVStride] = side ? nn.y : 0.f;
          slot[2 * kVStride] = side ? nz : 0.f;
          slot[3 * kVStride] = side ? 0.f : nn.x;
          slot[4 * kVStride] = side ? 0.f : nn.y;
          slot[5 * kVStride] = side ? 0.f : nz;
          na += side ? 1u : 0u;
        });
    na = block_sum_u32(na, s_cnt);
    __syncthreads();
    if (chain_wave && lane < 6) s_bc[lane] = pacc;
    __syncthreads();
    const float s6[6] = {s_bc[0], s_bc[1], s_bc[2], s_bc[3], s_bc[4], s_bc[5]};
    pass = verify_split(s6, na, (unsigned)sz - na, th_angle);
  }
  if (tid == 0) {
    rec[c].z_mean = mid / (float)ctz;
    rec[c].min_x = mnx;
    rec[c].min_y = mny;
    rec[c].max_x = mxx;
    rec[c].max_y = mxy;
    rec[c].pass = pass;
  }
}

// ---- one cloud across several GPUs (hope_shard_*) ------------------------
// The global bounding box (all-reduced by the caller) replaces the local one, so that every rank
// derives the same VoxelGrid min_b / div_b.
static __global__ void k_force_bbox(FrameInfo* fi, float mnx, float mny, float mnz, float mxx, float mxy, float mxz)
{
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  fi[0].bb_min[0] = f2ord(mnx);
  fi[0].bb_min[1] = f2ord(mny);
  fi[0].bb_min[2] = f2ord(mnz);
  fi[0].bb_max[0] = f2ord(mxx);
  fi[0].bb_max[1] = f2ord(mxy);
  fi[0].bb_max[2] = f2ord(mxz);
}

// the frame record of a call that brings pose and bounding box as kernel arguments (no host-to-device copy):
// all of it is written, so nothing of an earlier call survives in it
struct StaticFrame {
  float aff[12];
  float bbox[6];
};
static __global__ void k_static_frame(FrameInfo* fi, StaticFrame sf)
{
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  FrameInfo F = {};
  for (int k = 0; k < 12; ++k) F.aff[k] = sf.aff[k];
  for (int d = 0; d < 3; ++d) {
    F.bb_min[d] = f2ord(sf.bbox[d]);
    F.bb_max[d] = f2ord(sf.bbox[3 + d]);
  }
  F.n_raw = 1;  // (k_frame_grid only builds a grid for a frame with points)
  *fi = F;
}

// z-layer histogram of the surviving points: layer = VoxelGrid's k index.  Votes go to LDS-private
// bins of the workgroup (one ds_add per point; layers of a horizontal surface repeat within a
// wave, which the LDS atomic unit serialises cheaply) and are merged into the 64-bit global bins
// once per workgroup.  More layers than the LDS table holds: straight to the global bins.
constexpr int kHistLds = 8192;

template <class Src>
__global__ void __launch_bounds__(kBlock)
k_z_histogram(Src src, const FrameInfo* fi, float inv_z, int n_bins, unsigned long long* hist)
{
  __shared__ unsigned s_bins[kHistLds];
  const bool lds = n_bins <= kHistLds;
  if (lds)
    for (int b = threadIdx.x; b < n_bins; b += kBlock) s_bins[b] = 0u;
  __syncthreads();
  const unsigned n = src_count(src);
  const float mbz = (float)fi[0].min_b[2];
  for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    float x, y, z;
    if (!load_point(src, fi, 0, i, x, y, z)) continue;
    const int k = (int)(floorf(z * inv_z) - mbz);
    if (k < 0 || k >= n_bins) continue;
    if (lds) atomicAdd(&s_bins[k], 1u); else atomicAdd(&hist[k], 1ull);
  }
  __syncthreads();
  if (lds)
    for (int b = threadIdx.x; b < n_bins; b += kBlock)
      if (s_bins[b]) atomicAdd(&hist[b], (unsigned long long)s_bins[b]);
}

// destination rank of every point = owner of its z layer (points that do not survive A2: n_ranks)
template <class Src>
__global__ void __launch_bounds__(kBlock)
k_route_keys(Src src, const FrameInfo* fi, float inv_z, const int* layer_owner, int n_bins, int n_ranks,
             uint32_t* keys, uint32_t* vals)
{
  const unsigned n = src_count(src);
  const unsigned i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  float x, y, z;
  uint32_t key = (uint32_t)n_ranks;
  if (load_point(src, fi, 0, i, x, y, z)) {
    const int k = (int)(floorf(z * inv_z) - (float)fi[0].min_b[2]);
    if (k >= 0 && k < n_bins) key = (uint32_t)layer_owner[k];
  }
  keys[i] = key;
  vals[i] = i;
}

// transformed points in routed order (destination major, source index ascending) + per-rank counts
template <class Src>
__global__ void __launch_bounds__(kBlock)
k_route_export(Src src, const FrameInfo* fi, const uint32_t* skeys, const uint32_t* svals, int n_ranks, float* out)
{
  const unsigned n = src_count(src);
  const unsigned i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n || skeys[i] >= (uint32_t)n_ranks) return;
  float x = 0.f, y = 0.f, z = 0.f;
  load_point(src, fi, 0, svals[i], x, y, z);
  out[3 * (size_t)i] = x;
  out[3 * (size_t)i + 1] = y;
  out[3 * (size_t)i + 2] = z;
}

static __global__ void k_route_counts(const uint32_t* skeys, unsigned n, int n_ranks, unsigned long long* counts)
{
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n_ranks) return;
  unsigned lo = 0, hi = n;
  while (lo < hi) {  // first position with key >= r
    const unsigned mid = (lo + hi) >> 1;
    if (skeys[mid] < (uint32_t)r) lo = mid + 1; else hi = mid;
  }
  counts[r] = lo;  // prefix: counts[r+1] - counts[r] points go to rank r
}

// an already down-sampled cloud (VoxelGrid order) becomes the voxel stage's output as it is
template <class Src>
__global__ void __launch_bounds__(kBlock)
k_adopt_downsampled(Src src, FrameInfo* fi, float4* vox, int* vox_frame)
{
  const unsigned n = src_count(src);
  const unsigned i = blockIdx.x * kBlock + threadIdx.x;
  if (i == 0) {
    fi[0].vox_start = 0;
    fi[0].n_vox = (int)n;
  }
  if (i >= n) return;
  float x = 0.f, y = 0.f, z = 0.f;
  load_point(src, fi, 0, i, x, y, z);
  vox[i] = make_float4(x, y, z, 0.f);
  vox_frame[i] = 0;
}

// x, y, z of the down-sampled points, packed (hope_get_downsampled_to)
static __global__ void __launch_bounds__(kBlock) k_pack_xyz(const float4* vox, unsigned n, float* xyz)
{
  const unsigned i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float4 p = vox[i];
  xyz[3 * (size_t)i] = p.x;
  xyz[3 * (size_t)i + 1] = p.y;
  xyz[3 * (size_t)i + 2] = p.z;
}

// ---- on-demand export of the transformed cloud (hope_get_transformed) -------------------
template <class Src>
__global__ void __launch_bounds__(kBlock)
k_valid_flags(Src src, const FrameInfo* fi, int frame, uint32_t* flags)
{
  const unsigned n = src_count(src);
  const unsigned i = blockIdx.x * kBlock + threadIdx.x;
  if (i > n) return;
  float x, y, z;
  flags[i] = (i < n && load_point(src, fi, frame, i, x, y, z)) ? 1u : 0u;
}

template <class Src>
__global__ void __launch_bounds__(kBlock)
k_export_transformed(Src src, const FrameInfo* fi, int frame, const uint32_t* scan, float* xyz, int* src_index)
{
  const unsigned n = src_count(src);
  const unsigned i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  float x, y, z;
  if (load_point(src, fi, frame, i, x, y, z)) {
    const unsigned o = scan[i];
    xyz[3 * (size_t)o] = x;
    xyz[3 * (size_t)o + 1] = y;
    xyz[3 * (size_t)o + 2] = z;
    src_index[o] = (int)i;
  }
}

}  // namespace hope
