// voxel_part.hpp -- A5 (pcl::VoxelGrid, utilities.cpp:455-468) for ONE large cloud without a device-wide
// sort of (key, index) pairs and without a random gather of the points afterwards.
//
// VoxelGrid needs the points of every voxel together, voxels in ascending cell index, and -- because the
// centroid is a sequential float sum -- the points of a voxel in ascending input index.  The sort-based stage
// (k_voxel_keys, radix sort of 10^8 pairs, run-length encode, k_centroids_w) moves 8-byte pairs four times
// and then fetches every point again through a 64-byte sector for its 12 bytes: 7.2 ms for 10^8 points.
// Here the POINT travels, and only as far as it has to:
//   * the cell index is cut into  [ high digits | low 9 bits ].  One stable partition pass per high digit
//     (least significant first: k_part_count, a scan of the chunk x digit counts, k_part_scatter) moves
//     (cell, x, y, z) records so that afterwards every BUCKET -- the 512 cells that share their high digits
//     -- is one contiguous range, its records still in input order.  Two passes of 9 bits for the 2^27 cells
//     of the 10^8-point scene.  A pass ranks the records of a 2048-record tile by wave-wide ballot matching
//     (stable by construction), reorders the tile in LDS and writes runs, so stores stay coalesced.
//   * k_part_finish gives each bucket to ONE wave, which walks its records in order and adds each to the
//     accumulator of its cell in LDS.  Records of a batch that share a cell are added in lane order (the
//     lowest pending lane of a cell wins an LDS min and goes first), so every cell's sum is the reference's
//     chain, and lanes with different cells go at the same time.  Occupied cells leave in ascending order.
// No step needs the input index itself: stability carries the order.
#pragma once
#include "kernels.hpp"

namespace hope {

constexpr int kPartLow = 9;                        // low cell bits: cells of a bucket
constexpr int kPartCells = 1 << kPartLow;          // 512
constexpr int kPartMaxDigit = 9;                   // bits per partition pass
constexpr int kPartDigits = 1 << kPartMaxDigit;    // 512
constexpr int kPartIPT = 8;                        // records per thread and tile
constexpr int kPartTile = kBlock * kPartIPT;       // 2048
constexpr int kPartChunk = 8 * kPartTile;          // 16384 records per workgroup
constexpr unsigned kPartNone = 0xffffffffu;

// cell index of input point i (A2/A3 applied), false if the point is dropped
template <class Src>
__device__ __forceinline__ bool part_cell(const Src& src, const FrameInfo* fi, float inv_xy, float inv_z, unsigned i,
                                          float& x, float& y, float& z, unsigned& cell)
{
  if (!load_point(src, fi, 0, i, x, y, z)) return false;
  const FrameInfo& F = fi[0];
  const int i0 = (int)(floorf(x * inv_xy) - (float)F.min_b[0]);
  const int i1 = (int)(floorf(y * inv_xy) - (float)F.min_b[1]);
  const int i2 = (int)(floorf(z * inv_z) - (float)F.min_b[2]);
  cell = (unsigned)(i0 + i1 * F.div_b[0] + i2 * (F.div_b[0] * F.div_b[1]));
  return true;
}

// ---- counts[digit][chunk] of one pass ------------------------------------------------------------
// FROM_SRC: the first pass reads the input; later passes read the cell array of the previous pass
template <bool FROM_SRC, class Src>
__global__ void __launch_bounds__(kBlock)
k_part_count(Src src, const FrameInfo* fi, float inv_xy, float inv_z, const uint32_t* cell_in, unsigned n, int shift,
             int bits, uint32_t* counts, unsigned n_chunks)
{
  __shared__ unsigned s_hist[kPartDigits];
  const unsigned chunk = blockIdx.x;
  for (int d = threadIdx.x; d < kPartDigits; d += kBlock) s_hist[d] = 0u;
  __syncthreads();
  if (!FROM_SRC) n = (unsigned)fi[0].n_raw;  // the records that survived A2 (counted by k_bbox), compacted by pass 0
  const unsigned lo = min(n, chunk * (unsigned)kPartChunk);
  const unsigned hi = min(n, lo + (unsigned)kPartChunk);
  const unsigned mask = (1u << bits) - 1u;
  for (unsigned i = lo + threadIdx.x; i < hi; i += kBlock) {
    unsigned cell;
    bool ok;
    if (FROM_SRC) {
      float x, y, z;
      ok = part_cell(src, fi, inv_xy, inv_z, i, x, y, z, cell);
    } else {
      cell = cell_in[i];
      ok = true;
    }
    if (ok) atomicAdd(&s_hist[(cell >> shift) & mask], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < (1 << bits); d += kBlock) counts[(size_t)d * n_chunks + chunk] = s_hist[d];
}

// ---- the stable scatter of one pass ---------------------------------------------------------------
// offsets = exclusive scan of counts (digit-major): where the records of (digit, chunk) start.
// Record order inside a tile: wave w owns records [512 w, 512 w + 512) of the tile, item k of lane l is
// record 512 w + 64 k + l (coalesced loads); ranks follow that order.
template <bool FROM_SRC, class Src>
__global__ void __launch_bounds__(kBlock)
k_part_scatter(Src src, const FrameInfo* fi, float inv_xy, float inv_z, const uint32_t* cell_in, const float* xyz_in,
               unsigned n, int shift, int bits, const uint32_t* offsets, unsigned n_chunks, uint32_t* cell_out,
               float* xyz_out)
{
  constexpr int kWaves = kBlock / kWave;
  __shared__ unsigned s_cnt[kWaves][kPartDigits];  // records of (wave, digit) so far; then the waves' prefix
  __shared__ unsigned s_tot[kPartDigits];          // records of a digit in this tile
  __shared__ unsigned s_dstart[kPartDigits];       // where the digit starts in the reordered tile
  __shared__ unsigned s_base[kPartDigits];         // where the chunk's next record of a digit goes
  __shared__ unsigned s_scan[kWaves];
  __shared__ unsigned s_cell[kPartTile];
  __shared__ float s_xyz[3 * kPartTile];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), w = tid / kWave;
  const unsigned chunk = blockIdx.x;
  const int nd = 1 << bits;
  const unsigned dmask = (unsigned)nd - 1u;
  for (int d = tid; d < kPartDigits; d += kBlock) s_base[d] = d < nd ? offsets[(size_t)d * n_chunks + chunk] : 0u;
  if (!FROM_SRC) n = (unsigned)fi[0].n_raw;
  const unsigned c_lo = min(n, chunk * (unsigned)kPartChunk);
  const unsigned c_hi = min(n, c_lo + (unsigned)kPartChunk);
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (unsigned t0 = c_lo; t0 < c_hi; t0 += kPartTile) {
    for (int i = tid; i < kWaves * kPartDigits; i += kBlock) (&s_cnt[0][0])[i] = 0u;
    __syncthreads();
    unsigned cell[kPartIPT], rank[kPartIPT];
    float px[kPartIPT], py[kPartIPT], pz[kPartIPT];
    bool ok[kPartIPT];
    // loads first (all in flight together), ranking afterwards
#pragma unroll
    for (int k = 0; k < kPartIPT; ++k) {
      const unsigned i = t0 + (unsigned)(w * (kWave * kPartIPT) + k * kWave + lane);
      ok[k] = i < c_hi;
      cell[k] = 0u;
      px[k] = py[k] = pz[k] = 0.f;
      if (ok[k]) {
        if (FROM_SRC) {
          ok[k] = part_cell(src, fi, inv_xy, inv_z, i, px[k], py[k], pz[k], cell[k]);
        } else {
          cell[k] = cell_in[i];
          px[k] = xyz_in[3 * (size_t)i];
          py[k] = xyz_in[3 * (size_t)i + 1];
          pz[k] = xyz_in[3 * (size_t)i + 2];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < kPartIPT; ++k) {
      const unsigned d = (cell[k] >> shift) & dmask;
      // lanes of this batch with the same digit
      unsigned long long same = __ballot(ok[k]);
      for (int b = 0; b < bits; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long v = __ballot(bit);
        same &= bit ? v : ~v;
      }
      const unsigned before = s_cnt[w][d];  // (the wave's batches run in order: only this wave touches its row)
      rank[k] = before + (unsigned)__popcll(same & lt);
      if (ok[k] && (same & lt) == 0ull) s_cnt[w][d] = before + (unsigned)__popcll(same);
    }
    __syncthreads();
    // per digit: prefix over the waves, total; then the digits' starts in the reordered tile
    unsigned my_tot[kPartDigits / kBlock];
#pragma unroll
    for (int j = 0; j < kPartDigits / kBlock; ++j) {
      const int d = tid * (kPartDigits / kBlock) + j;  // consecutive digits per thread: the scan below is in digit order
      unsigned run = 0u;
#pragma unroll
      for (int ww = 0; ww < kWaves; ++ww) {
        const unsigned c = s_cnt[ww][d];
        s_cnt[ww][d] = run;
        run += c;
      }
      s_tot[d] = run;
      my_tot[j] = run;
    }
    unsigned mine = 0u;
#pragma unroll
    for (int j = 0; j < kPartDigits / kBlock; ++j) mine += my_tot[j];
    unsigned inc = mine;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const unsigned t = (unsigned)__shfl_up((int)inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == kWave - 1) s_scan[w] = inc;
    __syncthreads();
    unsigned pre = inc - mine;
    for (int ww = 0; ww < w; ++ww) pre += s_scan[ww];
    unsigned tile_n = 0u;
    for (int ww = 0; ww < kWaves; ++ww) tile_n += s_scan[ww];
#pragma unroll
    for (int j = 0; j < kPartDigits / kBlock; ++j) {
      s_dstart[tid * (kPartDigits / kBlock) + j] = pre;
      pre += my_tot[j];
    }
    __syncthreads();
    // reorder the tile in LDS
#pragma unroll
    for (int k = 0; k < kPartIPT; ++k) {
      if (!ok[k]) continue;
      const unsigned d = (cell[k] >> shift) & dmask;
      const unsigned p = s_dstart[d] + s_cnt[w][d] + rank[k];
      s_cell[p] = cell[k];
      s_xyz[3 * p] = px[k];
      s_xyz[3 * p + 1] = py[k];
      s_xyz[3 * p + 2] = pz[k];
    }
    __syncthreads();
    // write the runs out
    for (unsigned p = tid; p < tile_n; p += kBlock) {
      const unsigned cl = s_cell[p];
      const unsigned d = (cl >> shift) & dmask;
      const size_t g = (size_t)s_base[d] + (p - s_dstart[d]);
      cell_out[g] = cl;
      xyz_out[3 * g] = s_xyz[3 * p];
      xyz_out[3 * g + 1] = s_xyz[3 * p + 1];
      xyz_out[3 * g + 2] = s_xyz[3 * p + 2];
    }
    __syncthreads();
    for (int d = tid; d < nd; d += kBlock) s_base[d] += s_tot[d];
    // (the next tile's first barrier orders this against its readers)
  }
}

// ---- bucket ranges of the partitioned records -----------------------------------------------------
// bstart[h] = first record whose (cell >> kPartLow) >= h, for h = 0 .. n_buckets (a binary search per bucket)
__global__ void __launch_bounds__(kBlock)
k_part_bounds(const uint32_t* cell, const FrameInfo* fi, unsigned n_buckets, uint32_t* bstart)
{
  const unsigned m = (unsigned)fi[0].n_raw;  // records that survived A2 (counted by k_bbox)
  const unsigned h = blockIdx.x * kBlock + threadIdx.x;
  if (h > n_buckets) return;
  unsigned lo = 0u, hi = m;
  while (lo < hi) {
    const unsigned mid = lo + ((hi - lo) >> 1);
    if ((cell[mid] >> kPartLow) < h) lo = mid + 1u; else hi = mid;
  }
  bstart[h] = lo;
}

// ---- centroids: one wave per bucket -----------------------------------------------------------------
// out: the bucket's voxels in ascending cell order at out[bstart[b] ...] (a bucket has at most as many voxels as
// records), vox_cnt[b] their number.
__global__ void __launch_bounds__(kBlock)
k_part_finish(const uint32_t* cell, const float* xyz, const uint32_t* bstart, unsigned n_buckets, float4* out,
              uint32_t* vox_cnt)
{
  constexpr int kWaves = kBlock / kWave;
  constexpr int kPer = kPartCells / kWave;  // cells per lane at the end
  __shared__ float s_x[kWaves][kPartCells], s_y[kWaves][kPartCells], s_z[kWaves][kPartCells];
  __shared__ unsigned s_n[kWaves][kPartCells];
  __shared__ unsigned s_slot[kWaves][kPartCells];
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  const unsigned b = blockIdx.x * kWaves + w;
  if (b >= n_buckets) return;
  const unsigned r_lo = bstart[b], r_hi = bstart[b + 1];
  if (r_lo == r_hi) {
    if (lane == 0) vox_cnt[b] = 0u;
    return;
  }
  float* ax = s_x[w];
  float* ay = s_y[w];
  float* az = s_z[w];
  unsigned* an = s_n[w];
  volatile unsigned* slot = s_slot[w];
  for (int c = lane; c < kPartCells; c += kWave) {
    ax[c] = 0.f;
    ay[c] = 0.f;
    az[c] = 0.f;
    an[c] = 0u;
    slot[c] = kPartNone;
  }
  __builtin_amdgcn_wave_barrier();
  for (unsigned r0 = r_lo; r0 < r_hi; r0 += 4 * kWave) {
    // four batches in flight
    unsigned cl[4];
    float px[4], py[4], pz[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned i = r0 + (unsigned)(q * kWave + lane);
      const unsigned ii = min(i, r_hi - 1u);
      cl[q] = i < r_hi ? (cell[ii] & (unsigned)(kPartCells - 1)) : kPartNone;
      px[q] = xyz[3 * (size_t)ii];
      py[q] = xyz[3 * (size_t)ii + 1];
      pz[q] = xyz[3 * (size_t)ii + 2];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bool pending = cl[q] != kPartNone;
      const unsigned c = pending ? cl[q] : 0u;
      while (__any(pending)) {
        // the lowest pending lane of every cell goes now: ascending lane = ascending input index
        if (pending) atomicMin(const_cast<unsigned*>(&slot[c]), (unsigned)lane);
        __builtin_amdgcn_wave_barrier();
        const bool win = pending && slot[c] == (unsigned)lane;
        __builtin_amdgcn_wave_barrier();
        if (win) {
          ax[c] = ax[c] + px[q];
          ay[c] = ay[c] + py[q];
          az[c] = az[c] + pz[q];
          an[c] = an[c] + 1u;
          slot[c] = kPartNone;
          pending = false;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  // occupied cells in ascending order: lane l owns cells [kPer l, kPer l + kPer)
  unsigned occ = 0u;
#pragma unroll
  for (int j = 0; j < kPer; ++j) occ += an[lane * kPer + j] ? 1u : 0u;
  unsigned inc = occ;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const unsigned t = (unsigned)__shfl_up((int)inc, o);
    if (lane >= o) inc += t;
  }
  unsigned at = r_lo + inc - occ;
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    const int c = lane * kPer + j;
    const unsigned n = an[c];
    if (n) {
      const float fn = (float)n;
      out[at++] = make_float4(ax[c] / fn, ay[c] / fn, az[c] / fn, 0.f);
    }
  }
  if (lane == kWave - 1) vox_cnt[b] = inc;
}

// n_vox of the call and of frame 0 from the scanned bucket counts (vox_off[n_buckets] = total)
__global__ void k_part_vox_total(FrameInfo* fi, const uint32_t* vox_off, unsigned n_buckets, Globals* g)
{
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const unsigned n = vox_off[n_buckets];
  g->n_vox_total = n;
  fi[0].vox_start = 0;
  fi[0].n_vox = (int)n;
}

// the buckets' voxels to their places in the VoxelGrid output: one wave per bucket
__global__ void __launch_bounds__(kBlock)
k_part_compact(const float4* tmp, const uint32_t* bstart, const uint32_t* vox_off, unsigned n_buckets, float4* vox,
               int* vox_frame)
{
  const int lane = threadIdx.x & (kWave - 1);
  const unsigned b = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  if (b >= n_buckets) return;
  const unsigned o = vox_off[b], n = vox_off[b + 1] - o, src = bstart[b];
  for (unsigned i = lane; i < n; i += kWave) {
    vox[o + i] = tmp[src + i];
    vox_frame[o + i] = 0;
  }
}

}  // namespace hope
