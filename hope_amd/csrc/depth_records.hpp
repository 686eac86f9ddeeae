// depth_records.hpp -- A1..A5 for batches of depth frames by (chunk, voxel) group records: the kernels that read
// the depth images (k_bbox_records, the two-pass fallback k_bbox_groups + k_group_records, k_group_keys and the
// record walk of the centroids).  A header of its own so that the neighbour-search kernels (neighbours.hip, three
// minutes of compile time) do not depend on it.
#pragma once
#include "kernels.hpp"

namespace hope {

// ---- depth-frame voxel stage by (chunk, voxel) groups --------------------------------------
// A depth image is spatially coherent: the 64 pixels of a small image rectangle (a chunk: kChunkRows
// rows of kChunkCols pixels, lane = row * kChunkCols + column) fall into a handful of voxels.  Instead
// of sorting one (key, index) pair per pixel, the depth path sorts one record per (chunk, voxel)
// GROUP -- about 10x fewer -- carrying the 64-bit mask of the group's pixels.  Records are emitted
// in chunk order (band of rows, then left to right) and the radix sort is stable, so a voxel's
// records arrive band by band; within a band its pixels interleave row by row across the chunks,
// which k_centroids_groups walks as (row, record, bit): ascending pixel index, the same sequential
// float sum as VoxelGrid's (voxel_grid.hpp).  A 2-D chunk meets fewer voxels than 64 pixels of
// one row: 1.95x fewer records at 4x16 than at 1x64 on the synthetic frames.
constexpr int kChunk = 64;
constexpr unsigned kXcds = 8u;  // accelerator dies of an MI355X: workgroup i of a launch runs on XCD i % 8, each with its own L2
constexpr int kChunkRows = 4;
constexpr int kChunkCols = kChunk / kChunkRows;

__device__ __forceinline__ float readlane_f(float v, int lane)
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// v_writelane_b32: put a wave-uniform value into one lane of a VGPR (hipcc has no builtin for
// it).  VOP3 may read one SGPR only, so the lane select goes through M0 (s_nop: M0 write ->
// use); the 4-wait-state rule for a VALU-written lane select does not apply to an SALU write.
__device__ __forceinline__ void writelane_u64(unsigned long long uniform_val, int uniform_lane, int& lo, int& hi)
{
  int saved_m0;  // M0 is reserved by the compiler: saved and restored inside the statement
  asm("s_mov_b32 %2, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tv_writelane_b32 %0, %3, m0\n\t"
      "v_writelane_b32 %1, %4, m0\n\ts_mov_b32 m0, %2"
      : "+v"(lo), "+v"(hi), "=&s"(saved_m0)
      : "s"((int)(unsigned)uniform_val), "s"((int)(unsigned)(uniform_val >> 32)), "s"(uniform_lane));
}

// a / b for a divisor that is constant over the call.  FAST: one Markstein correction step,
// q0 = a*(1/b), r = fma(-b, q0, a), q = fma(r, 1/b, q0) -- 3 instructions instead of the ~11 of
// the IEEE division sequence.  It is only used after k_verify_fast_div has compared it with the
// IEEE quotient for EVERY dividend the call can produce (65536 depth codes x image columns /
// rows), so the bits are those of the reference's division by construction.
template <bool FAST>
__device__ __forceinline__ float div_const(float a, float b, float rcp_b)
{
  if (!FAST) return a / b;
  const float q0 = a * rcp_b;
  const float r = __fmaf_rn(-b, q0, a);
  return __fmaf_rn(r, rcp_b, q0);
}

// every dividend of A1 for this camera model: fails (sets *fail) if div_const<true> ever differs
// from the IEEE quotient.  grid = (65536 / kBlock, max(w, h)).
static __global__ void __launch_bounds__(kBlock) k_verify_fast_div(DepthSrc s, unsigned* fail)
{
  const unsigned d = blockIdx.x * kBlock + threadIdx.x;
  const int i = blockIdx.y;
  if (d > 65535u) return;
  const float zf = (float)d;
  const float z = zf / s.scale;
  bool bad = __float_as_uint(z) != __float_as_uint(div_const<true>(zf, s.scale, s.rcp_scale));
  if (z < s.max_d && z > s.min_d) {
    if (i < s.w) {
      const float a = ((float)i - s.cx) * z;
      bad |= __float_as_uint(a / s.fx) != __float_as_uint(div_const<true>(a, s.fx, s.rcp_fx));
    }
    if (i < s.h) {
      const float a = ((float)i - s.cy) * z;
      bad |= __float_as_uint(a / s.fy) != __float_as_uint(div_const<true>(a, s.fy, s.rcp_fy));
    }
  }
  if (bad) *fail = 1u;
}

// A1..A3 with the pose held in registers (same arithmetic as load_point(DepthSrc))
template <bool FAST>
__device__ __forceinline__ bool depth_point(const DepthSrc& s, const float* aff, int frame, unsigned pix, float& ox,
                                            float& oy, float& oz)
{
  const uint16_t d = s.depth[(size_t)frame * s.ppf + pix];
  const int row = (int)(pix / (unsigned)s.w);
  const int col = (int)(pix - (unsigned)row * (unsigned)s.w);
  const float z = div_const<FAST>((float)d, s.scale, s.rcp_scale);
  if (!(z < s.max_d && z > s.min_d)) return false;
  const float x = div_const<FAST>(((float)col - s.cx) * z, s.fx, s.rcp_fx);
  const float y = div_const<FAST>(((float)row - s.cy) * z, s.fy, s.rcp_fy);
  apply_affine(aff, x, y, z, ox, oy, oz);
  return true;
}

// Groups of one chunk: lanes with equal cell triple (gx, gy, gz).  Returns the number of groups;
// with MASKS also, per group, the pixel mask in the lane of the group's first pixel (0 elsewhere)
// and the mask of those first lanes.  Fast path: the triple is packed relative to the chunk's
// first surviving pixel into 11+11+10 bits, one compare per group; a chunk whose offsets do not
// fit (a depth jump of more than 51 m / 10 m) compares the three floats.  Both passes call this
// same function, so their group counts agree by construction.
template <bool MASKS>
__device__ __forceinline__ unsigned chunk_groups(bool ok, float gx, float gy, float gz, unsigned long long& my_grp)
{
  unsigned long long todo = __builtin_amdgcn_ballot_w64(ok);
  unsigned ng = 0;
  int g_lo = 0, g_hi = 0;  // this lane's group mask, written by v_writelane from the scalar ballot
  if (todo) {
    const int l0 = __ffsll((long long)todo) - 1;
    const float dx = gx - readlane_f(gx, l0), dy = gy - readlane_f(gy, l0), dz = gz - readlane_f(gz, l0);
    const bool fit = !ok || (fabsf(dx) < 1023.f && fabsf(dy) < 1023.f && fabsf(dz) < 511.f);
    if (__all(fit)) {
      const unsigned code = ok ? ((unsigned)((int)dx + 1024) | ((unsigned)((int)dy + 1024) << 11) |
                                  ((unsigned)((int)dz + 512) << 22))
                               : 0xffffffffu;
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned k = (unsigned)__builtin_amdgcn_readlane((int)code, leader);
        const unsigned long long grp = __builtin_amdgcn_ballot_w64(code == k);
        if (MASKS) {
          writelane_u64(grp, leader, g_lo, g_hi);
        }
        todo &= ~grp;
        ++ng;
      }
    } else {
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const float lx = readlane_f(gx, leader), ly = readlane_f(gy, leader), lz = readlane_f(gz, leader);
        const unsigned long long grp = __builtin_amdgcn_ballot_w64(ok && gx == lx && gy == ly && gz == lz) | (1ull << leader);
        if (MASKS) {
          writelane_u64(grp, leader, g_lo, g_hi);
        }
        todo &= ~grp;
        ++ng;
      }
    }
  }
  my_grp = (unsigned long long)(unsigned)g_lo | ((unsigned long long)(unsigned)g_hi << 32);
  return ng;
}

// The same groups from a cell code that every surviving lane already holds (k_bbox_records: the code relative to
// the frame's camera cell): no fit test and no second compare path.  Lanes whose cell is outside the code range hold
// garbage -- the caller raises Globals::code_overflow for them and the call is repeated in two passes.
__device__ __forceinline__ unsigned chunk_groups_code(unsigned long long okm, unsigned code, unsigned long long& my_grp)
{
  unsigned long long todo = okm;
  unsigned ng = 0;
  int g_lo = 0, g_hi = 0;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const unsigned k = (unsigned)__builtin_amdgcn_readlane((int)code, leader);
    const unsigned long long grp = __builtin_amdgcn_ballot_w64(code == k) & okm;
    writelane_u64(grp, leader, g_lo, g_hi);
    todo &= ~grp;
    ++ng;
  }
  my_grp = (unsigned long long)(unsigned)g_lo | ((unsigned long long)(unsigned)g_hi << 32);
  return ng;
}

// fminf / fmaxf of values that are never NaN, without the canonicalising v_max(x, x) the compiler puts in front of
// every operand it cannot prove quiet (the kernels here are bound by VALU issue)
__device__ __forceinline__ float min_nn(float a, float b)
{
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float max_nn(float a, float b)
{
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// row / column of this lane's pixel of chunk c (cpb = chunks per band of rows); the divisions are
// wave-uniform
__device__ __forceinline__ void chunk_rowcol(unsigned c, int lane, unsigned cpb, int& row, int& col)
{
  const unsigned band = c / cpb;
  const unsigned cib = c - band * cpb;
  row = (int)(band * kChunkRows) + lane / kChunkCols;
  col = (int)(cib * kChunkCols) + lane % kChunkCols;
}

// A1..A3 for the pixel at (row, col), pose in registers
template <bool FAST>
__device__ __forceinline__ bool depth_point_rc(const DepthSrc& s, const float* aff, int frame, unsigned pix, int row,
                                               int col, float& ox, float& oy, float& oz)
{
  const uint16_t d = s.depth[(size_t)frame * s.ppf + pix];
  const float z = div_const<FAST>((float)d, s.scale, s.rcp_scale);
  if (!(z < s.max_d && z > s.min_d)) return false;
  const float x = div_const<FAST>(((float)col - s.cx) * z, s.fx, s.rcp_fx);
  const float y = div_const<FAST>(((float)row - s.cy) * z, s.fy, s.rcp_fy);
  apply_affine(aff, x, y, z, ox, oy, oz);
  return true;
}

// the same from a depth sample that has been fetched already (k_bbox_records reads one chunk ahead)
template <bool FAST>
__device__ __forceinline__ bool depth_point_d(const DepthSrc& s, const float* aff, unsigned d, int row, int col,
                                              float& ox, float& oy, float& oz)
{
  const float z = div_const<FAST>((float)d, s.scale, s.rcp_scale);
  if (!(z < s.max_d && z > s.min_d)) return false;
  const float x = div_const<FAST>(((float)col - s.cx) * z, s.fx, s.rcp_fx);
  const float y = div_const<FAST>(((float)row - s.cy) * z, s.fy, s.rcp_fy);
  apply_affine(aff, x, y, z, ox, oy, oz);
  return true;
}

// Pass 1: bbox + survivor count per frame (getMinMax3D) and the number of groups per chunk.
// Groups are found on the un-shifted cell triple floor(p * inv_leaf), which needs no bbox.
// BOX false: the group counts only (the call fell back from k_bbox_records<.., DENSE>, which has already left the
// bounding boxes and the survivor counts)
template <bool FAST, bool BOX>
__global__ void __launch_bounds__(kBlock)
k_bbox_groups(DepthSrc src, FrameInfo* fi, float inv_xy, float inv_z, unsigned cpf, unsigned cpb, uint32_t* grp_cnt)
{
  const int frame = blockIdx.y;
  const int lane = threadIdx.x & (kWave - 1);
  const unsigned wv = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  const unsigned nwv = gridDim.x * (kBlock / kWave);
  float aff[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) aff[k] = fi[frame].aff[k];
  float mn[3] = {INFINITY, INFINITY, INFINITY};
  float mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  int cnt = 0;
  for (unsigned c = wv; c < cpf; c += nwv) {
    int row, col;
    chunk_rowcol(c, lane, cpb, row, col);
    const unsigned pix = (unsigned)row * (unsigned)src.w + (unsigned)col;
    float x = 0.f, y = 0.f, z = 0.f;
    const bool ok = row < src.h && col < src.w && depth_point_rc<FAST>(src, aff, frame, pix, row, col, x, y, z);
    const float gx = floorf(x * inv_xy), gy = floorf(y * inv_xy), gz = floorf(z * inv_z);
    if (ok) {
      mn[0] = fminf(mn[0], x);
      mn[1] = fminf(mn[1], y);
      mn[2] = fminf(mn[2], z);
      mx[0] = fmaxf(mx[0], x);
      mx[1] = fmaxf(mx[1], y);
      mx[2] = fmaxf(mx[2], z);
      ++cnt;
    }
    unsigned long long mg;
    const unsigned ng = chunk_groups<false>(ok, gx, gy, gz, mg);
    if (lane == 0) grp_cnt[(size_t)frame * cpf + c] = ng;
  }
  if (!BOX) return;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      mn[d] = fminf(mn[d], __shfl_down(mn[d], off));
      mx[d] = fmaxf(mx[d], __shfl_down(mx[d], off));
    }
    cnt += __shfl_down(cnt, off);
  }
  // one set of atomics per block (hundreds of same-address atomics per frame serialise in L2)
  __shared__ float s_mn[kBlock / kWave][3], s_mx[kBlock / kWave][3];
  __shared__ int s_cnt[kBlock / kWave];
  const int w_in_b = threadIdx.x / kWave;
  if (lane == 0) {
    for (int d = 0; d < 3; ++d) {
      s_mn[w_in_b][d] = mn[d];
      s_mx[w_in_b][d] = mx[d];
    }
    s_cnt[w_in_b] = cnt;
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

// Pass 2: one record per group: sort key = frame << cell_bits | PCL voxel index, payload =
// (global chunk id, 64-bit pixel mask, row and column of the chunk's first pixel).  grp_off = exclusive scan of pass 1's counts.
template <class KeyT, bool FAST>
__global__ void __launch_bounds__(kBlock)
k_group_records(DepthSrc src, const FrameInfo* fi, float inv_xy, float inv_z, int cell_bits, unsigned cpf,
                unsigned cpb, const uint32_t* grp_off, KeyT* rkey, uint32_t* rval, uint4* rec)
{
  const int frame = blockIdx.y;
  const int lane = threadIdx.x & (kWave - 1);
  const unsigned wv = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  const unsigned nwv = gridDim.x * (kBlock / kWave);
  const FrameInfo& F = fi[frame];
  float aff[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) aff[k] = F.aff[k];
  const float mb0 = (float)F.min_b[0], mb1 = (float)F.min_b[1], mb2 = (float)F.min_b[2];
  const int d0 = F.div_b[0], d01 = F.div_b[0] * F.div_b[1];
  const KeyT fbits = (KeyT)frame << cell_bits;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (unsigned c = wv; c < cpf; c += nwv) {
    int row, col;
    chunk_rowcol(c, lane, cpb, row, col);
    const unsigned pix = (unsigned)row * (unsigned)src.w + (unsigned)col;
    float x = 0.f, y = 0.f, z = 0.f;
    const bool ok = row < src.h && col < src.w && depth_point_rc<FAST>(src, aff, frame, pix, row, col, x, y, z);
    const float gx = floorf(x * inv_xy), gy = floorf(y * inv_xy), gz = floorf(z * inv_z);
    unsigned long long my_grp;
    chunk_groups<true>(ok, gx, gy, gz, my_grp);
    const unsigned long long leaders = __builtin_amdgcn_ballot_w64(my_grp != 0ull);
    if (my_grp) {
      const size_t cg = (size_t)frame * cpf + c;
      const unsigned o = grp_off[cg] + (unsigned)__popcll(leaders & lt);
      const int i0 = (int)(gx - mb0), i1 = (int)(gy - mb1), i2 = (int)(gz - mb2);
      rkey[o] = fbits | (KeyT)(unsigned)(i0 + i1 * d0 + i2 * d01);
      rval[o] = o;
      // (row, column) of the chunk's first pixel: 16 bits each (the launcher checks the image size)
      const unsigned rc0 = (((unsigned)row - (unsigned)(lane / kChunkCols)) << 16) | ((unsigned)col - (unsigned)(lane % kChunkCols));
      rec[o] = make_uint4((unsigned)cg, (unsigned)my_grp, (unsigned)(my_grp >> 32), rc0);
    }
  }
}

// ---- one pass over the depth batch instead of two ----------------------------------------------
// k_bbox_groups + k_group_records read the batch and derive every point twice, because the sort
// key needs the bounding box.  k_bbox_records does the per-pixel work once: it writes each group's
// record -- with the cell as a packed code relative to the frame's camera cell, which needs no
// bounding box -- into a fixed slot of kChunk records per chunk (a sparse array: only the slots
// used are ever touched), next to the group count and the bounding box.  k_group_keys then turns
// (count scan, codes, bounding box) into the dense (key, value) pairs of the sort; value = slot of
// the record, which is what k_centroids_groups gathers.  Cells outside the code range (+-1023 cells
// in x and y, +-511 in z around the camera) raise Globals::code_overflow and the call falls back to
// the two passes.
__device__ __forceinline__ void frame_ref_cell(const float* aff, float inv_xy, float inv_z, float& rx, float& ry, float& rz)
{
  rx = floorf(aff[9] * inv_xy);
  ry = floorf(aff[10] * inv_xy);
  rz = floorf(aff[11] * inv_z);
}

constexpr unsigned kCountStride = 64u;  // words between the record counters of two frames: a cache line (an L2 channel) each
// the n records a wave has staged in LDS go to the next free places of its frame's range; false: the range is full
__device__ __forceinline__ bool dense_flush(const uint4* staged, unsigned n, int lane, int frame, unsigned cap_f,
                                            uint32_t* rec_count, uint4* rec, uint32_t* codes, unsigned cpb)
{
  // (the records were written by other lanes of this wave: LDS runs a wave's instructions in order; the fences keep
  // the compiler from moving the reads above the writes)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  unsigned base = 0u;
  if (lane == 0) base = atomicAdd(&rec_count[(size_t)frame * kCountStride], n);
  base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
  const bool fits = base + n <= cap_f;
  if (fits) {
    const size_t at = (size_t)frame * cap_f + base;
    for (unsigned i = (unsigned)lane; i < n; i += (unsigned)kWave) {
      const uint4 r = staged[i];
      rec[at + i] = r;
      codes[at + i] = r.x;
      // the record's chunk number within the frame (its rank in pixel order), for k_voxel_rank's sort words
      codes[(size_t)gridDim.y * cap_f + at + i] = (r.w >> 16) / (unsigned)kChunkRows * cpb + (r.w & 0xffffu) / (unsigned)kChunkCols;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();  // (the next chunk's records overwrite the staging area)
  return fits;
}

__device__ __forceinline__ int floor_to_int(float v)  // (int)floorf(v) for |v| < 2^31
{
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}

// FULL: the image is a whole number of chunks (no pixel of a chunk lies outside it).
// DENSE: the records do not go to the sparse slots (and no group counts are written); a wave collects them in LDS
// and claims room for kDenseCap of them at a time in its frame's range of `slots` (frame * cap_f onwards, cell codes
// also on their own in `codes`) with one atomic on rec_count[frame].  The records of a frame then lie densely, in no
// particular order -- which k_voxel_rank does not need, and k_centroids_lists restores per voxel from the records'
// chunk positions -- and the scan of the counts and k_group_keys' pass over the sparse slots fall away.  A frame with
// more than cap_f records, like a cell outside the code range, raises Globals::code_overflow (two passes then).
constexpr int kDenseCap = 256;
template <bool FAST, bool FULL, bool DENSE>
__global__ void __launch_bounds__(kBlock)
k_bbox_records(DepthSrc src, FrameInfo* fi, float inv_xy, float inv_z, unsigned cpf, unsigned cpb, uint32_t* grp_cnt,
               uint4* slots, Globals* g, unsigned cap_f, uint32_t* rec_count, uint32_t* codes)
{
  __shared__ uint4 s_rec[DENSE ? kBlock / kWave : 1][DENSE ? kDenseCap : 1];
  unsigned n_staged = 0u;  // (wave-uniform)
  bool full_frame = false;
  const int frame = blockIdx.y;
  const int lane = threadIdx.x & (kWave - 1);
  // (the wave's number as a scalar: the chunk's row and column then come out of the scalar unit)
  const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave));
  const unsigned nwv = gridDim.x * (kBlock / kWave);
  float aff[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) aff[k] = fi[frame].aff[k];
  float rx, ry, rz;
  frame_ref_cell(aff, inv_xy, inv_z, rx, ry, rz);
  const unsigned bx = 1023u - (unsigned)(int)rx, by = 1023u - (unsigned)(int)ry, bz = 511u - (unsigned)(int)rz;
  const unsigned long long lt = (1ull << lane) - 1ull;
  float mn[3] = {INFINITY, INFINITY, INFINITY};
  float mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  int cnt = 0;  // (wave-uniform: survivors of the wave's chunks)
  bool over = false;
  // (a wave takes a run of consecutive chunks: its dense records then come from neighbouring voxels, which
  // k_voxel_rank's LDS look-ups need: 0.57 -> 0.44 ms; a run per block with the waves taking turns: 0.49 ms)
  const unsigned per_wave = (cpf + nwv - 1u) / nwv;
  // the depth sample of the next chunk is fetched before this chunk's arithmetic (the kernel is bound by VALU issue:
  // a wave that waits for its own load at the top of every chunk leaves issue slots empty)
  const uint16_t* dp = src.depth + (size_t)frame * src.ppf;
  const auto fetch = [&](unsigned c, int& row, int& col, bool& in) -> unsigned {
    chunk_rowcol(c, lane, cpb, row, col);
    in = FULL || (row < src.h && col < src.w);
    return in ? (unsigned)dp[(unsigned)row * (unsigned)src.w + (unsigned)col] : 0u;
  };
  const unsigned c_begin = wv * per_wave, c_end = min(c_begin + per_wave, cpf);
  int row_n = 0, col_n = 0;
  bool in_n = false;
  unsigned d_n = c_begin < c_end ? fetch(c_begin, row_n, col_n, in_n) : 0u;
  for (unsigned c = c_begin; c < c_end; ++c) {
    const int row = row_n, col = col_n;
    const bool in = in_n;
    const unsigned d = d_n;
    if (c + 1u < c_end) d_n = fetch(c + 1u, row_n, col_n, in_n);
    float x = 0.f, y = 0.f, z = 0.f;
    const bool ok = in && depth_point_d<FAST>(src, aff, d, row, col, x, y, z);
    const unsigned long long okm = __builtin_amdgcn_ballot_w64(ok);
    // cell relative to the camera's, + 1023: in range when it fits 0 .. 2046 (0 .. 1022 in z); the packed code holds
    // the same offsets + 1, as before (floor and conversion in one instruction; integer from there on)
    const unsigned ux = (unsigned)floor_to_int(x * inv_xy) + bx, uy = (unsigned)floor_to_int(y * inv_xy) + by,
                   uz = (unsigned)floor_to_int(z * inv_z) + bz;
    const unsigned code = (ux | (uy << 11) | (uz << 22)) + (1u | (1u << 11) | (1u << 22));
    if (ok) {
      mn[0] = min_nn(mn[0], x);
      mn[1] = min_nn(mn[1], y);
      mn[2] = min_nn(mn[2], z);
      mx[0] = max_nn(mx[0], x);
      mx[1] = max_nn(mx[1], y);
      mx[2] = max_nn(mx[2], z);
    }
    cnt += __popcll(okm);
    // (every surviving pixel is tested, not the leaders only: the garbage code of a cell outside the range can equal
    // the code of a cell inside it)
    const bool out = ok && !(ux <= 2046u && uy <= 2046u && uz <= 1022u);
    unsigned long long my_grp;
    unsigned ng;
    if (__builtin_amdgcn_ballot_w64(out)) {
      // the call will be repeated in two passes, which keep these counts: count the groups the way k_group_records
      // finds them (the records of this chunk are not used; DENSE: the two passes count for themselves)
      over = true;
      ng = DENSE ? 0u : chunk_groups<false>(ok, floorf(x * inv_xy), floorf(y * inv_xy), floorf(z * inv_z), my_grp);
      my_grp = 0ull;
    } else {
      ng = chunk_groups_code(okm, code, my_grp);
    }
    const size_t cg = (size_t)frame * cpf + c;
    if (!DENSE && lane == 0) grp_cnt[cg] = ng;
    const unsigned long long leaders = __builtin_amdgcn_ballot_w64(my_grp != 0ull);
    const unsigned rc0 = (((unsigned)row - (unsigned)(lane / kChunkCols)) << 16) | ((unsigned)col - (unsigned)(lane % kChunkCols));
    if (DENSE) {
      if (n_staged + ng > (unsigned)kDenseCap) {
        full_frame |= !dense_flush(s_rec[threadIdx.x / kWave], n_staged, lane, frame, cap_f, rec_count, slots, codes, cpb);
        n_staged = 0u;
      }
      if (my_grp)
        s_rec[threadIdx.x / kWave][n_staged + (unsigned)__popcll(leaders & lt)] =
            make_uint4(code, (unsigned)my_grp, (unsigned)(my_grp >> 32), rc0);
      n_staged += ng;
    } else if (my_grp) {
      slots[cg * kChunk + (unsigned)__popcll(leaders & lt)] = make_uint4(code, (unsigned)my_grp, (unsigned)(my_grp >> 32), rc0);
    }
  }
  if (DENSE && n_staged > 0u)
    full_frame |= !dense_flush(s_rec[threadIdx.x / kWave], n_staged, lane, frame, cap_f, rec_count, slots, codes, cpb);
  if ((__any(over) || full_frame) && lane == 0) g->code_overflow = 1u;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      mn[d] = fminf(mn[d], __shfl_down(mn[d], off));
      mx[d] = fmaxf(mx[d], __shfl_down(mx[d], off));
    }
  }
  __shared__ float s_mn[kBlock / kWave][3], s_mx[kBlock / kWave][3];
  __shared__ int s_cnt[kBlock / kWave];
  const int w_in_b = threadIdx.x / kWave;
  if (lane == 0) {
    for (int d = 0; d < 3; ++d) {
      s_mn[w_in_b][d] = mn[d];
      s_mx[w_in_b][d] = mx[d];
    }
    s_cnt[w_in_b] = cnt;
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

// dense (key, value) pairs in chunk order: key = frame << cell_bits | PCL voxel index, value = slot
// of the record.  Eight lanes per chunk (a chunk has about six groups).
template <class KeyT>
__global__ void __launch_bounds__(kBlock)
k_group_keys(const FrameInfo* fi, float inv_xy, float inv_z, int cell_bits, unsigned cpf, unsigned n_chunks,
             const uint32_t* grp_off, const uint4* slots, KeyT* rkey, uint32_t* rval)
{
  const unsigned t = blockIdx.x * kBlock + threadIdx.x;
  const unsigned cg = t >> 3;
  if (cg >= n_chunks) return;
  const unsigned o0 = grp_off[cg], ng = grp_off[cg + 1] - o0;
  if ((t & 7u) >= ng) return;
  const int frame = (int)(cg / cpf);
  const FrameInfo& F = fi[frame];
  float rx, ry, rz;
  frame_ref_cell(F.aff, inv_xy, inv_z, rx, ry, rz);
  const int b0 = (int)rx - 1024 - F.min_b[0], b1 = (int)ry - 1024 - F.min_b[1], b2 = (int)rz - 512 - F.min_b[2];
  const int d0 = F.div_b[0], d01 = F.div_b[0] * F.div_b[1];
  const KeyT fbits = (KeyT)frame << cell_bits;
  for (unsigned sl = t & 7u; sl < ng; sl += 8u) {
    const unsigned slot = cg * kChunk + sl;
    const unsigned code = slots[slot].x;
    const int i0 = (int)(code & 2047u) + b0, i1 = (int)((code >> 11) & 2047u) + b1, i2 = (int)(code >> 22) + b2;
    rkey[o0 + sl] = fbits | (KeyT)(unsigned)(i0 + i1 * d0 + i2 * d01);
    rval[o0 + sl] = slot;
  }
}

// range of each frame's records in emission order (grp_off = scan of the per-chunk group counts)
static __global__ void k_frame_record_ranges(const uint32_t* grp_off, unsigned cpf, int n_frames, uint32_t* seg_off)
{
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f <= n_frames) seg_off[f] = grp_off[(size_t)f * cpf];
}

// Centroids from the sorted group records: one lane per voxel.  A voxel's records arrive band by
// band (chunk order); the records of one band are walked row by row -- (row, record, bit) is
// ascending pixel index.  Most voxels have one record per band, which is then walked from bit 0 up.
// A1..A3 from an already fetched depth code (same arithmetic as depth_point_rc; the masks hold
// surviving pixels only)
template <bool FAST>
__device__ __forceinline__ void depth_code_point(const DepthSrc& s, const float (&aff)[12], unsigned d, unsigned row,
                                                 unsigned col, float& ox, float& oy, float& oz)
{
  const float z = div_const<FAST>((float)d, s.scale, s.rcp_scale);
  const float x = div_const<FAST>(((float)(int)col - s.cx) * z, s.fx, s.rcp_fx);
  const float y = div_const<FAST>(((float)(int)row - s.cy) * z, s.fy, s.rcp_fy);
  apply_affine(aff, x, y, z, ox, oy, oz);
}

// the points of mask m (chunk at rc0) in ascending bit order, four depth reads in flight per step
template <bool FAST>
__device__ __forceinline__ void centroid_add_bits(const DepthSrc& src, const float (&aff)[12], int frame, unsigned rc0,
                                                  unsigned long long m, float& sx, float& sy, float& sz)
{
  const unsigned row0 = rc0 >> 16, col0 = rc0 & 0xffffu;
  const uint16_t* dp = src.depth + (size_t)frame * src.ppf;
  while (m) {
    unsigned rr[4], cc[4], dd[4];
    bool on[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      on[i] = m != 0ull;
      const int b = on[i] ? __ffsll((long long)m) - 1 : 0;
      m &= m - 1ull;  // (0 stays 0)
      rr[i] = row0 + (unsigned)(b / kChunkCols);
      cc[i] = col0 + (unsigned)(b % kChunkCols);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) dd[i] = dp[rr[i] * (unsigned)src.w + cc[i]];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (on[i]) {
        float x, y, z;
        depth_code_point<FAST>(src, aff, dd[i], rr[i], cc[i], x, y, z);
        sx += x;
        sy += y;
        sz += z;
      }
    }
  }
}

template <class KeyT, bool FAST>
__global__ void __launch_bounds__(kBlock)
k_centroids_groups(DepthSrc src, const FrameInfo* fi, const KeyT* unique, const uint32_t* run_start,
                   const uint32_t* run_count, const uint32_t* sval, const uint4* rec, unsigned n_vox, int cell_bits,
                   float4* vox, int* vox_frame)
{
  const unsigned v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= n_vox) return;
  const int frame = (int)(unique[v] >> cell_bits);
  float aff[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) aff[k] = fi[frame].aff[k];
  const unsigned start = run_start[v], cnt = run_count[v];
  constexpr unsigned long long kRowMask = (kChunkCols == 64) ? ~0ull : ((1ull << (kChunkCols & 63)) - 1ull);
  float sx = 0.f, sy = 0.f, sz = 0.f;
  unsigned n = 0;
  // (records as scalars: mask halves and the packed row/column of the chunk's first pixel)
  unsigned r_lo, r_hi, r_rc;
  {
    const uint4 t = rec[sval[start]];  // every voxel has at least one record
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
      const uint4 t = rec[sval[start + e1]];
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
    // one record: its bits in ascending order; several: row by row across the records
    const int n_rows = (e1 == e + 1u) ? 1 : kChunkRows;
    for (int row = 0; row < n_rows; ++row) {
      for (unsigned ee = e; ee < e1; ++ee) {
        unsigned m_lo, m_hi, m_rc;
        if (ee == e) {
          m_lo = r_lo; m_hi = r_hi; m_rc = r_rc;
        } else if (ee == e + 1u) {
          m_lo = b_lo; m_hi = b_hi; m_rc = b_rc;
        } else if (ee == e + 2u) {
          m_lo = c_lo; m_hi = c_hi; m_rc = c_rc;
        } else {
          const uint4 t = rec[sval[start + ee]];
          m_lo = t.y; m_hi = t.z; m_rc = t.w;
        }
        unsigned long long m = (unsigned long long)m_lo | ((unsigned long long)m_hi << 32);
        if (n_rows > 1) m &= kRowMask << (row * kChunkCols);
        n += (unsigned)__popcll(m);
        centroid_add_bits<FAST>(src, aff, frame, m_rc, m, sx, sy, sz);
      }
    }
    e = e1;
    r_lo = n_lo;
    r_hi = n_hi;
    r_rc = n_rc;
  }
  const float c = (float)n;
  vox[v] = make_float4(sx / c, sy / c, sz / c, 0.f);
  vox_frame[v] = frame;
}

// first voxel of each frame = lower_bound over the unique keys
template <class KeyT>
__global__ void k_frame_vox_ranges(FrameInfo* fi, int n_frames, const KeyT* unique, const unsigned* n_runs_p,
                                   int cell_bits, Globals* g)
{
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f > n_frames) return;
  const unsigned n_runs = *n_runs_p;
  const KeyT target = (KeyT)f << cell_bits;
  unsigned lo = 0, hi = n_runs;
  while (lo < hi) {
    const unsigned mid = (lo + hi) >> 1;
    if (unique[mid] < target) lo = mid + 1; else hi = mid;
  }
  if (f == n_frames) {
    g->n_vox_total = lo;
  } else {
    fi[f].vox_start = (int)lo;
  }
}

static __global__ void k_frame_vox_counts(FrameInfo* fi, int n_frames, const Globals* g)
{
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_frames) return;
  const int next = (f + 1 < n_frames) ? fi[f + 1].vox_start : (int)g->n_vox_total;
  fi[f].n_vox = next - fi[f].vox_start;
}

}  // namespace hope
