// voxel_rank.hpp — A5 (pcl::VoxelGrid, utilities.cpp:455-468) for depth batches without a sort.
//
// k_bbox_records has left one record per (chunk, voxel) group and k_group_keys the dense (voxel index,
// record slot) pairs of every frame, in chunk order.  VoxelGrid's output order is ascending voxel index,
// and a frame's grid has a few hundred thousand to a few million cells: a bitmap of the occupied cells
// fits in LDS (in one to a few cell ranges), and the rank of a set bit IS the voxel's place in the
// output.  One 1024-thread workgroup per frame therefore replaces the segmented radix sort of the
// records, the run-length encoding and the scans around them:
//   per cell range:   set the bit of every record's cell (LDS atomic OR); popcount scan -> prefix per
//                     256 bits; voxel of a record = rank of its bit (written next to the record)
//   per voxel range:  records per voxel (LDS atomic add); scan -> first place of every voxel's records;
//                     each record takes the next free place of its voxel (returning LDS atomic add)
// The places inside a voxel come out in arrival order; k_centroids_lists sorts each voxel's few records
// by slot number (= chunk order: band of rows, then left to right) in LDS before it walks them, which is
// the order the stable sort used to leave.
#pragma once
#include "depth_records.hpp"
#include "sortnet.hpp"

namespace hope {

constexpr int kRankBlock = 1024;

// exclusive scan of one value per thread over a kRankBlock-thread workgroup; s_w: 17 words of LDS
__device__ __forceinline__ unsigned block_scan_excl_1024(unsigned v, unsigned* s_w, unsigned& total)
{
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  unsigned inc = v;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const unsigned t = (unsigned)__shfl_up((int)inc, off);
    if (lane >= off) inc += t;
  }
  __syncthreads();  // (s_w may still be read by the previous scan's callers)
  if (lane == kWave - 1) s_w[wv] = inc;
  __syncthreads();
  if (threadIdx.x < kWave) {
    const unsigned w = lane < kRankBlock / kWave ? s_w[lane] : 0u;
    unsigned wi = w;
#pragma unroll
    for (int off = 1; off < kRankBlock / kWave; off <<= 1) {
      const unsigned t = (unsigned)__shfl_up((int)wi, off);
      if (lane >= off) wi += t;
    }
    if (lane < kRankBlock / kWave) s_w[lane] = wi - w;
    if (lane == kRankBlock / kWave - 1) s_w[kRankBlock / kWave] = wi;
  }
  __syncthreads();
  total = s_w[kRankBlock / kWave];
  return s_w[wv] + inc - v;
}

constexpr unsigned kBmWords = 32768;              // bitmap words per cell range: 2^20 cells, 128 KiB
constexpr unsigned kBmBits = kBmWords * 32u;
constexpr unsigned kPfxEvery = 8;                 // one prefix entry per 8 bitmap words (two 16-byte LDS reads)
constexpr unsigned kPfxWords = kBmWords / kPfxEvery;
constexpr unsigned kRankLds = 40704;              // words of LDS (159 KiB): bitmap + prefix, later the counters
constexpr unsigned kCntCap = kRankLds;            // voxels per voxel range
static_assert(kBmWords + kPfxWords <= kRankLds, "bitmap and prefix table share the counters' LDS");

// every record o of [0, n) once: four loads in flight per thread, then body(o, a[o] (, b[o])) for each
template <class Body>
__device__ __forceinline__ void for_each_record(unsigned n, const uint32_t* a, Body&& body)
{
  for (unsigned base = threadIdx.x; base < n; base += 4u * kRankBlock) {
    unsigned x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned o = base + (unsigned)u * kRankBlock;
      x[u] = a[o < n ? o : n - 1u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned o = base + (unsigned)u * kRankBlock;
      if (o < n) body(o, x[u]);
    }
  }
}

// the same over two arrays (b may be null: its values are then 0)
template <class Body>
__device__ __forceinline__ void for_each_record2(unsigned n, const uint32_t* a, const uint32_t* b, Body&& body)
{
  for (unsigned base = threadIdx.x; base < n; base += 4u * kRankBlock) {
    unsigned x[4], y[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned o = base + (unsigned)u * kRankBlock;
      x[u] = a[o < n ? o : n - 1u];
      y[u] = b ? b[o < n ? o : n - 1u] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned o = base + (unsigned)u * kRankBlock;
      if (o < n) body(o, x[u], y[u]);
    }
  }
}

__device__ __forceinline__ unsigned popc4(const uint4& q)
{
  return (unsigned)(__popc(q.x) + __popc(q.y) + __popc(q.z) + __popc(q.w));
}

// grid = n_frames, block = kRankBlock.  keys: cell index of every record (low cell_bits bits of k_group_keys'
// sort key), vals: its slot; both dense, in emission (chunk) order, the frame's records at
// [grp_off[f * cpf], grp_off[(f + 1) * cpf]).
// DENSE (k_bbox_records<.., true>): the frame's records at [f * cap_f, f * cap_f + rec_count[f]) in no particular
// order, keys = their packed cell codes relative to the camera's cell (decoded here with the frame's grid), and a
// record's value is its own index.
template <bool DENSE>
__global__ void __launch_bounds__(kRankBlock)
k_voxel_rank(const FrameInfo* fi, unsigned cpf, const uint32_t* grp_off, const uint32_t* keys, const uint32_t* vals,
             unsigned cell_mask, uint32_t* rec_vox, uint32_t* rec_list, uint32_t* vox_end, uint32_t* vox_cnt,
             unsigned cap_f, const uint32_t* rec_count, float inv_xy, float inv_z, int id_bits,
             const uint32_t* order)  // order: the frame of every workgroup (k_frame_order), or null
{
  __shared__ __attribute__((aligned(16))) unsigned s_mem[kRankLds];
  __shared__ unsigned s_w[kRankBlock / kWave + 1];
  unsigned* bitmap = s_mem;             // kBmWords
  unsigned* prefix = s_mem + kBmWords;  // kPfxWords
  unsigned* counters = s_mem;           // kCntCap (after the ranks have been written)
  const int f = order ? (int)order[blockIdx.x] : (int)blockIdx.x;
  const unsigned tid = threadIdx.x;
  const unsigned rec_base = DENSE ? (unsigned)f * cap_f : grp_off[(size_t)f * cpf];
  const unsigned n_rec = DENSE ? min(rec_count[(size_t)f * kCountStride], cap_f) : grp_off[(size_t)(f + 1) * cpf] - rec_base;
  if (n_rec == 0u) {
    if (tid == 0) vox_cnt[f] = 0u;
    return;
  }
  // (DENSE) cell index of a packed code: the same arithmetic as k_group_keys
  int b0 = 0, b1 = 0, b2 = 0, d0 = 0, d01 = 0;
  if (DENSE) {
    const FrameInfo& F = fi[f];
    float rx, ry, rz;
    frame_ref_cell(F.aff, inv_xy, inv_z, rx, ry, rz);
    b0 = (int)rx - 1024 - F.min_b[0];
    b1 = (int)ry - 1024 - F.min_b[1];
    b2 = (int)rz - 512 - F.min_b[2];
    d0 = F.div_b[0];
    d01 = F.div_b[0] * F.div_b[1];
  }
  const auto cell_of = [&](unsigned k) -> unsigned {
    if (!DENSE) return k & cell_mask;
    const int i0 = (int)(k & 2047u) + b0, i1 = (int)((k >> 11) & 2047u) + b1, i2 = (int)(k >> 22) + b2;
    return (unsigned)(i0 + i1 * d0 + i2 * d01);
  };
  const unsigned cells = (unsigned)fi[f].cells;
  const uint32_t* key = keys + rec_base;
  const uint32_t* val = DENSE ? nullptr : vals + rec_base;
  uint32_t* rvox = rec_vox + rec_base;
  const uint4* bm4 = reinterpret_cast<const uint4*>(bitmap);
  // ---- voxel of every record = rank of its cell among the occupied cells, one cell range at a time ----
  unsigned n_vox = 0u;
  for (unsigned lo = 0u; lo < cells; lo += kBmBits) {  // (uniform)
    const unsigned span = min(kBmBits, cells - lo);
    const unsigned n_grp = (span + 32u * kPfxEvery - 1u) / (32u * kPfxEvery);  // prefix entries in use
    for (unsigned i = tid; i < n_grp * kPfxEvery; i += kRankBlock) bitmap[i] = 0u;
    __syncthreads();
    for_each_record(n_rec, key, [&](unsigned, unsigned k) {
      const unsigned idx = cell_of(k) - lo;
      if (idx < span) atomicOr(&bitmap[idx >> 5], 1u << (idx & 31u));
    });
    __syncthreads();
    const unsigned per = (n_grp + kRankBlock - 1) / kRankBlock;
    unsigned mine = 0u;
    for (unsigned k = 0; k < per; ++k) {
      const unsigned gi = tid * per + k;
      if (gi < n_grp) mine += popc4(bm4[2 * gi]) + popc4(bm4[2 * gi + 1]);
    }
    unsigned total;
    unsigned ex = block_scan_excl_1024(mine, s_w, total);
    for (unsigned k = 0; k < per; ++k) {
      const unsigned gi = tid * per + k;
      if (gi < n_grp) {
        prefix[gi] = n_vox + ex;
        ex += popc4(bm4[2 * gi]) + popc4(bm4[2 * gi + 1]);
      }
    }
    __syncthreads();
    for_each_record(n_rec, key, [&](unsigned o, unsigned k) {
      const unsigned idx = cell_of(k) - lo;
      if (idx < span) {
        const unsigned w = idx >> 5, sub = w & 7u;
        const uint4 qa = bm4[(w >> 3) * 2u], qb = bm4[(w >> 3) * 2u + 1u];
        const unsigned wd[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
        unsigned r = prefix[w >> 3];
        unsigned cur = 0u;
#pragma unroll
        for (unsigned j = 0; j < 8u; ++j) {
          r += j < sub ? (unsigned)__popc(wd[j]) : 0u;
          cur = j == sub ? wd[j] : cur;
        }
        rvox[o] = r + (unsigned)__popc(cur & ((1u << (idx & 31u)) - 1u));
      }
    });
    n_vox += total;
    __syncthreads();  // (the next range clears the bitmap; the counters below reuse it)
  }
  // ---- the records of every voxel, one voxel range at a time (counters alias the bitmap) ----------------
  unsigned list_base = 0u;  // places taken by the voxels of earlier ranges
  for (unsigned vlo = 0u; vlo < n_vox; vlo += kCntCap) {  // (uniform)
    const unsigned nv = min(kCntCap, n_vox - vlo);
    for (unsigned i = tid; i < nv; i += kRankBlock) counters[i] = 0u;
    __syncthreads();
    for_each_record(n_rec, rvox, [&](unsigned, unsigned rv) {
      const unsigned v = rv - vlo;
      if (v < nv) atomicAdd(&counters[v], 1u);
    });
    __syncthreads();
    const unsigned perv = (nv + kRankBlock - 1) / kRankBlock;
    unsigned sum = 0u;
    for (unsigned k = 0; k < perv; ++k) {
      const unsigned v = tid * perv + k;
      if (v < nv) sum += counters[v];
    }
    unsigned total;
    unsigned at = list_base + block_scan_excl_1024(sum, s_w, total);
    for (unsigned k = 0; k < perv; ++k) {
      const unsigned v = tid * perv + k;
      if (v < nv) {
        const unsigned cnt = counters[v];
        counters[v] = at;
        at += cnt;
      }
    }
    __syncthreads();
    // DENSE: a sort word, chunk number (the record's rank in pixel order) above the record's index in the frame's
    // range (the second half of `keys` holds the chunk numbers): k_centroids_lists sorts a voxel's words
    for_each_record2(n_rec, rvox, DENSE ? keys + (size_t)gridDim.x * cap_f + rec_base : nullptr,
                     [&](unsigned o, unsigned rv, unsigned chunk) {
      const unsigned v = rv - vlo;
      if (v < nv) rec_list[rec_base + atomicAdd(&counters[v], 1u)] = DENSE ? ((chunk << id_bits) | o) : val[o];
    });
    __syncthreads();
    // counters[v] is now one past voxel v's last place
    for (unsigned i = tid; i < nv; i += kRankBlock) vox_end[rec_base + vlo + i] = counters[i];
    list_base += total;
    __syncthreads();
  }
  if (tid == 0) vox_cnt[f] = n_vox;
}

// range of every frame in the global down-sampled arrays (one workgroup)
// also block_first[f] = first block of kBlock voxels of frame f in k_centroids_lists' grid (entry n_frames: all)
__global__ void __launch_bounds__(kRankBlock)
k_frame_vox_scan(FrameInfo* fi, int n_frames, const uint32_t* vox_cnt, uint32_t* block_first, Globals* g,
                 const uint32_t* rec_count = nullptr)  // (dense records: their total goes to Globals::n_groups)
{
  __shared__ unsigned s_w[kRankBlock / kWave + 1];
  unsigned carry = 0u, bcarry = 0u;
  for (int base = 0; base < n_frames; base += kRankBlock) {
    const int f = base + (int)threadIdx.x;
    const unsigned v = f < n_frames ? vox_cnt[f] : 0u;
    unsigned total, btotal;
    const unsigned ex = block_scan_excl_1024(v, s_w, total);
    const unsigned bex = block_scan_excl_1024((v + kBlock - 1u) / kBlock, s_w, btotal);
    if (f < n_frames) {
      fi[f].vox_start = (int)(carry + ex);
      fi[f].n_vox = (int)v;
      block_first[f] = bcarry + bex;
    }
    carry += total;
    bcarry += btotal;
  }
  if (threadIdx.x == 0) {
    g->n_vox_total = carry;
    g->max_vox = bcarry;  // (blocks of the centroid kernel)
    block_first[n_frames] = bcarry;
  }
  if (rec_count) {
    unsigned mine = 0u;
    for (int f = (int)threadIdx.x; f < n_frames; f += kRankBlock) mine += rec_count[(size_t)f * kCountStride];
    unsigned total;
    block_scan_excl_1024(mine, s_w, total);
    if (threadIdx.x == 0) g->n_groups = total;
  }
}

// ---- the column index of A6 / A8 without a sort ------------------------------------------------------------------
// The searches want the down-sampled points of a frame grouped by xy column, every column ascending in z cell, and
// the table of column starts.  VoxelGrid's output already ascends in z cell, so a STABLE counting sort by column
// does it, and a frame's columns (a few ten thousand) fit in LDS as counters: one workgroup per frame counts the
// points of every column, scans, and gives each point the next free place of its column -- in arrival order, so
// a last pass moves every point to its place in index order (idx_tmp: n_vox indices of scratch).
// Replaces k_column_keys + the device-wide radix sort + k_column_table + k_gather_sorted for frames whose table
// (columns + 1) fits kCntCap counters.  grid = n_frames, block = kRankBlock.
__global__ void __launch_bounds__(kRankBlock)
k_column_index(const float4* vox, const FrameInfo* fi, int n_frames, float inv_xy, unsigned n_vox_total,
               unsigned total_cols, uint32_t* col_first, uint32_t* idx_tmp, float4* spt, const uint32_t* order)
{
  extern __shared__ unsigned s_cnt[];  // the batch's largest table (Globals::max_cols entries, at most kCntCap): two workgroups per CU when it is small
  __shared__ unsigned s_w[kRankBlock / kWave + 1];
  const FrameInfo& F = fi[order ? order[blockIdx.x] : blockIdx.x];
  const unsigned tid = threadIdx.x;
  const int dx = F.div_b[0], dy = F.div_b[1];
  const unsigned T = (unsigned)(dx * dy) + 1u;  // (the last entry closes the frame's part of the table)
  const unsigned n = (unsigned)F.n_vox, base = (unsigned)F.vox_start;
  if (blockIdx.x == (unsigned)n_frames - 1u && tid == 0) col_first[total_cols] = n_vox_total;
  for (unsigned i = tid; i < T; i += kRankBlock) s_cnt[i] = 0u;
  __syncthreads();
  const float4* pv = vox + base;
  auto column = [&](const float4& p) {
    const int cx = clampi(cell_unclamped(p.x, inv_xy, F.min_b[0]), 0, dx - 1);
    const int cy = clampi(cell_unclamped(p.y, inv_xy, F.min_b[1]), 0, dy - 1);
    return (unsigned)(cy * dx + cx);
  };
  // (four loads in flight per thread: a pass is a handful of dependent round trips to HBM otherwise)
  for (unsigned i0 = tid; i0 < n; i0 += 4u * kRankBlock) {
    float4 p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) p[u] = pv[min(i0 + (unsigned)u * kRankBlock, n - 1u)];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + (unsigned)u * kRankBlock < n) atomicAdd(&s_cnt[column(p[u])], 1u);
  }
  __syncthreads();
  const unsigned per = (T + kRankBlock - 1u) / kRankBlock;
  unsigned sum = 0u;
  for (unsigned k = 0; k < per; ++k) {
    const unsigned c = tid * per + k;
    if (c < T) sum += s_cnt[c];
  }
  unsigned total;
  unsigned at = block_scan_excl_1024(sum, s_w, total);
  uint32_t* cf = col_first + (unsigned)F.col_off;
  for (unsigned k = 0; k < per; ++k) {
    const unsigned c = tid * per + k;
    if (c < T) {
      const unsigned cnt = s_cnt[c];
      s_cnt[c] = at;
      at += cnt;
    }
  }
  __syncthreads();
  for (unsigned c = tid; c < T; c += kRankBlock) cf[c] = base + s_cnt[c];  // (coalesced)
  __syncthreads();  // (the placement below advances the counters)
  // every point takes the next free place of its column (arrival order) and leaves its index there ...
  uint32_t* pi = idx_tmp + base;
  for (unsigned i0 = tid; i0 < n; i0 += 4u * kRankBlock) {
    float4 p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) p[u] = pv[min(i0 + (unsigned)u * kRankBlock, n - 1u)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned i = i0 + (unsigned)u * kRankBlock;
      if (i < n) pi[atomicAdd(&s_cnt[column(p[u])], 1u)] = i;
    }
  }
  __syncthreads();  // (s_cnt[c] is now one past column c's last place; the indices are visible to the workgroup)
  // ... then moves to its place in INDEX order: first place of the column + the number of the column's points with
  // a smaller index (columns hold one point, often two, rarely a wall's worth: no chain of dependent loads either way)
  float4* ps = spt + base;
  for (unsigned i0 = tid; i0 < n; i0 += 4u * kRankBlock) {
    float4 p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) p[u] = pv[min(i0 + (unsigned)u * kRankBlock, n - 1u)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned i = i0 + (unsigned)u * kRankBlock;
      if (i < n) {
        const unsigned c = column(p[u]);
        const unsigned a = c > 0u ? s_cnt[c - 1u] : 0u, b = s_cnt[c];
        unsigned rank = 0u;
        if (b - a > 1u)
          for (unsigned j = a; j < b; ++j) rank += pi[j] < i ? 1u : 0u;
        p[u].w = __uint_as_float(base + i);  // (the PCL-order index, as k_gather_sorted leaves it)
        ps[a + rank] = p[u];
      }
    }
  }
}

// Centroids (CentroidPoint<PointXYZ>: sequential float sum / n) from the per-voxel record lists: one lane per
// voxel, one block per kBlock voxels of a frame (k_frame_vox_scan lays the blocks out).  The lists of a block's voxels are one contiguous range
// of rec_list: copied to LDS, each sorted by its lane (a few entries), then walked exactly as
// k_centroids_groups walks the sorted records.
constexpr int kListCap = 2048;  // entries of LDS per block (8 KiB: 256 voxels of ~3 records; occupancy is then bounded by registers, not LDS); a block with more sorts and walks in HBM

template <class ListPtr>
__device__ __forceinline__ void sort_small(ListPtr list, unsigned n)
{
  for (unsigned i = 1; i < n; ++i) {
    const unsigned x = list[i];
    unsigned j = i;
    while (j > 0u && list[j - 1u] > x) {
      list[j] = list[j - 1u];
      --j;
    }
    list[j] = x;
  }
}

// the pixels of one image row of a record, or of two records side by side (m: bit b = column col0 + b), in ascending
// column order, four depth
// reads in flight per step.  All of it in 32-bit lane arithmetic: the kernel is bound by VALU issue, and the 64-bit
// mask walk of centroid_add_bits costs twice the instructions per pixel.
template <bool FAST>
__device__ __forceinline__ void centroid_add_row(const DepthSrc& src, const float (&aff)[12], const uint16_t* dp,
                                                 unsigned row, unsigned col0, unsigned m, float& sx, float& sy,
                                                 float& sz)
{
  const uint16_t* rp = dp + (row * (unsigned)src.w + col0);  // (row, col0) is inside the image: the row has pixels
  while (m) {
    unsigned bb[4], dd[4];
    bool on[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      on[i] = m != 0u;
      bb[i] = on[i] ? (unsigned)__builtin_ctz(m) : 0u;
      m &= m - 1u;  // (0 stays 0)
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) dd[i] = rp[bb[i]];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (on[i]) {
        float x, y, z;
        depth_code_point<FAST>(src, aff, dd[i], row, col0 + bb[i], x, y, z);
        sx += x;
        sy += y;
        sz += z;
      }
    }
  }
}

template <bool FAST, class ListPtr>
__device__ __forceinline__ void centroid_walk(const DepthSrc& src, const float (&aff)[12], int frame, ListPtr list,
                                              unsigned cnt, const uint4* rec, float& sx, float& sy, float& sz,
                                              unsigned& n)
{
  static_assert(kChunkCols == 16 && kChunkRows == 4, "row masks are taken as 16-bit quarters of the record's mask");
  const uint16_t* dp = src.depth + (size_t)frame * src.ppf;
  unsigned r_lo, r_hi, r_rc;
  {
    const uint4 t = rec[list[0]];
    r_lo = t.y;
    r_hi = t.z;
    r_rc = t.w;
  }
  for (unsigned e = 0; e < cnt;) {
    // [e, e1): the records of one band of image rows; the second and third stay in registers
    unsigned e1 = e + 1u;
    unsigned n_lo = r_lo, n_hi = r_hi, n_rc = r_rc;
    unsigned b_lo = 0u, b_hi = 0u, b_rc = 0u, c_lo = 0u, c_hi = 0u, c_rc = 0u;
    while (e1 < cnt) {
      const uint4 t = rec[list[e1]];
      n_lo = t.y;
      n_hi = t.z;
      n_rc = t.w;
      if ((n_rc >> 16) != (r_rc >> 16)) break;
      if (e1 == e + 1u) {
        b_lo = n_lo;
        b_hi = n_hi;
        b_rc = n_rc;
      } else if (e1 == e + 2u) {
        c_lo = n_lo;
        c_hi = n_hi;
        c_rc = n_rc;
      }
      ++e1;
    }
    const unsigned row0 = r_rc >> 16, col0 = r_rc & 0xffffu;
    // a voxel's second record of the band is nearly always the next chunk to the right: its row joins the first
    // record's row as the upper half of one 32-bit mask, and the lanes of a wave stay in step (one walk per row)
    const bool join = e1 >= e + 2u && (b_rc & 0xffffu) == col0 + (unsigned)kChunkCols;
    const unsigned j_lo = join ? b_lo : 0u, j_hi = join ? b_hi : 0u;
    const unsigned rest = e + (join ? 2u : 1u);
    for (unsigned row = 0; row < (unsigned)kChunkRows; ++row) {
      const unsigned sh = (row & 1u) * 16u;
      const unsigned m = ((((row & 2u) ? r_hi : r_lo) >> sh) & 0xffffu) | ((((row & 2u) ? j_hi : j_lo) >> sh) << 16);
      n += (unsigned)__popc(m);
      centroid_add_row<FAST>(src, aff, dp, row0 + row, col0, m, sx, sy, sz);
      for (unsigned ee = rest; ee < e1; ++ee) {
        unsigned m_lo, m_hi, m_rc;
        if (ee == e + 1u) {
          m_lo = b_lo; m_hi = b_hi; m_rc = b_rc;
        } else if (ee == e + 2u) {
          m_lo = c_lo; m_hi = c_hi; m_rc = c_rc;
        } else {
          const uint4 t = rec[list[ee]];
          m_lo = t.y; m_hi = t.z; m_rc = t.w;
        }
        const unsigned mr = (((row & 2u) ? m_hi : m_lo) >> sh) & 0xffffu;
        n += (unsigned)__popc(mr);
        centroid_add_row<FAST>(src, aff, dp, row0 + row, m_rc & 0xffffu, mr, sx, sy, sz);
      }
    }
    e = e1;
    r_lo = n_lo;
    r_hi = n_hi;
    r_rc = n_rc;
  }
}

// DENSE: the frame's records are those of k_bbox_records<.., true> (range f * cap_f of `rec`, any order)
// a voxel's sort words (DENSE) into ascending order, then each replaced by the record index it carries.  Up to 16
// words go through a sorting network in registers (the words arrive in no order: an insertion sort would move half
// of them, lane by lane); all lanes of the wave call this.
template <class ListPtr>
__device__ __forceinline__ void sort_words(ListPtr list, unsigned n, unsigned id_mask, unsigned rec_base)
{
  const bool over8 = __any(n > 8u), over16 = __any(n > 16u);
  if (n > 16u) {
    sort_small(list, n);
    for (unsigned i = 0; i < n; ++i) list[i] = rec_base + (list[i] & id_mask);
    return;
  }
  (void)over16;
  unsigned w[16];
#define HOPE_CE(a, b)                         \
  {                                           \
    const unsigned lo_ = min(w[a], w[b]);     \
    w[b] = max(w[a], w[b]);                   \
    w[a] = lo_;                               \
  }
  if (!over8) {
#pragma unroll
    for (unsigned i = 0; i < 8u; ++i) w[i] = i < n ? list[i] : 0xffffffffu;
    HOPE_SORT_NET8
#pragma unroll
    for (unsigned i = 0; i < 8u; ++i)
      if (i < n) list[i] = rec_base + (w[i] & id_mask);
  } else {
#pragma unroll
    for (unsigned i = 0; i < 16u; ++i) w[i] = i < n ? list[i] : 0xffffffffu;
    HOPE_SORT_NET16
#pragma unroll
    for (unsigned i = 0; i < 16u; ++i)
      if (i < n) list[i] = rec_base + (w[i] & id_mask);
  }
#undef HOPE_CE
}

template <bool FAST, bool DENSE>
__global__ void __launch_bounds__(kBlock)
k_centroids_lists(DepthSrc src, const FrameInfo* fi, int n_frames, const uint32_t* block_first, unsigned cpf,
                  const uint32_t* grp_off, uint32_t* rec_list, const uint32_t* vox_end, const uint4* rec, float4* vox,
                  int* vox_frame, unsigned cap_f, int id_bits)
{
  __shared__ unsigned s_list[kListCap];
  // Workgroups are dealt to the eight XCDs in turn, and each XCD has its own L2.  The voxels of a frame share
  // their depth lines and records with the voxels a few blocks on (the next row of cells), so the blocks of a
  // frame belong on ONE XCD: XCD x takes the x-th eighth of the blocks, in order.
  const unsigned n_blocks = block_first[n_frames];
  const unsigned per_xcd = (n_blocks + kXcds - 1u) / kXcds;
  const unsigned blk = (blockIdx.x % kXcds) * per_xcd + blockIdx.x / kXcds;
  if (blk >= n_blocks) return;
  // frame of this block: the last one whose first block is <= blk (frames without voxels own no block)
  int lo = 0, hi = n_frames;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (block_first[mid] <= blk) lo = mid; else hi = mid;
  }
  const int frame = lo;
  const FrameInfo& F = fi[frame];
  const unsigned n_vox = (unsigned)F.n_vox;
  const unsigned v0 = (blk - block_first[frame]) * kBlock;
  if (v0 >= n_vox) return;
  const unsigned rec_base = DENSE ? (unsigned)frame * cap_f : grp_off[(size_t)frame * cpf];
  const unsigned v = v0 + threadIdx.x;
  const bool live = v < n_vox;
  const unsigned v_last = min(v0 + (unsigned)kBlock, n_vox) - 1u;
  const unsigned S = v0 > 0u ? vox_end[rec_base + v0 - 1u] : 0u;
  const unsigned E = vox_end[rec_base + v_last];
  const unsigned start = live ? (v > 0u ? vox_end[rec_base + v - 1u] : 0u) : 0u;
  const unsigned cnt = live ? vox_end[rec_base + v] - start : 0u;
  float aff[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) aff[k] = F.aff[k];
  float sx = 0.f, sy = 0.f, sz = 0.f;
  unsigned n = 0;
  if (E - S <= (unsigned)kListCap) {  // (block-uniform)
    for (unsigned i = threadIdx.x; i < E - S; i += kBlock) s_list[i] = rec_list[rec_base + S + i];
    __syncthreads();
    const unsigned id_mask = (1u << id_bits) - 1u;
    if (DENSE) sort_words(s_list + (live ? start - S : 0u), cnt, id_mask, rec_base);  // (every lane: wave votes inside)
    if (live) {
      unsigned* list = s_list + (start - S);
      if (!DENSE) sort_small(list, cnt);
      centroid_walk<FAST>(src, aff, frame, list, cnt, rec, sx, sy, sz, n);
    }
  } else if (live) {
    uint32_t* list = rec_list + rec_base + start;
    sort_small(list, cnt);
    if (DENSE)
      for (unsigned i = 0; i < cnt; ++i) list[i] = rec_base + (list[i] & ((1u << id_bits) - 1u));
    centroid_walk<FAST>(src, aff, frame, list, cnt, rec, sx, sy, sz, n);
  }
  if (live) {
    const float c = (float)n;
    vox[(unsigned)F.vox_start + v] = make_float4(sx / c, sy / c, sz / c, 0.f);  // every voxel has a record: n > 0
    vox_frame[(unsigned)F.vox_start + v] = frame;
  }
}

}  // namespace hope
