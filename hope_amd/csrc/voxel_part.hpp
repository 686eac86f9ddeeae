// voxel_part.hpp -- A5 (pcl::VoxelGrid, utilities.cpp:455-468) for ONE large cloud without a device-wide
// sort of (key, index) pairs and without a random gather of the points afterwards.
//
// VoxelGrid needs the points of every voxel together, voxels in ascending cell index, and -- because the
// centroid is a sequential float sum -- the points of a voxel in ascending input index.  The sort-based stage
// (k_voxel_keys, radix sort of 10^8 pairs, run-length encode, k_centroids_w) moves 8-byte pairs four times
// and then fetches every point again through a 64-byte sector for its 12 bytes: 7.2 ms for 10^8 points.
// Here the POINT travels, as one 16-byte record (x, y, z, cell), and only as far as it has to:
//   * the cell index is cut into  [ high digits | low 9 bits ].  One stable partition pass per high digit
//     (least significant first: k_part_count, a scan of the digit x chunk counts, k_part_scatter) moves the
//     records so that afterwards every BUCKET -- the 512 cells that share their high digits -- is one
//     contiguous range, its records still in input order.  Two passes of 9 bits for the 2^27 cells of the
//     10^8-point scene.  A pass ranks the records of a 4096-record tile by wave-wide ballot matching (stable
//     by construction), reorders the tile in LDS and writes runs of whole records.
//   * k_part_finish turns a bucket into its voxels.  A bucket of up to 64 records (clutter) is one wave's
//     work: lanes with the same cell are found by ballots and the first of them adds the group's points in
//     lane order.  Any other bucket is a workgroup's: tiles of 2048 records are ranked by cell the same way,
//     reordered in LDS, and one thread per cell adds its run to the cell's running sums in registers -- a
//     shelf row of 10^5 records in 30 cells costs 30 chains in parallel, not one wave's walk over all of them.
//     Every cell's sum is the reference's chain (ascending input index); occupied cells leave in ascending order.
// No step needs the input index itself: stability carries the order.
#pragma once
#include "kernels.hpp"
#include "scan.hpp"

namespace hope {

constexpr int kPartLow = 9;                        // low cell bits: cells of a bucket
constexpr int kPartCells = 1 << kPartLow;          // 512
constexpr int kPartMaxDigit = 9;                   // bits per partition pass
constexpr int kPartDigits = 1 << kPartMaxDigit;    // 512
constexpr int kPartThreads = 512;                  // workgroup of a scatter pass (one digit per thread)
constexpr int kPartIPT = 8;                        // records per thread and tile
constexpr int kPartTile = kPartThreads * kPartIPT; // 4096
constexpr int kPartChunk = 4 * kPartTile;          // 16384 records per workgroup
constexpr int kFinThreads = 256;                   // workgroup of k_part_finish
constexpr int kFinWaves = kFinThreads / kWave;
constexpr int kFinTile = kFinThreads * 8;          // 2048 records
constexpr int kFinLong = 16384;                    // records from which a bucket is started early
constexpr int kFinGrid = 1024;                     // workgroups of k_part_finish_big (4 per CU)
static_assert(kPartThreads == kPartDigits, "k_part_scatter scans one digit per thread");
static_assert(kFinThreads * 2 == kPartCells, "k_part_finish keeps two cells per thread");

// cell index of an (already filtered and transformed) point
__device__ __forceinline__ unsigned part_cell_of(const FrameInfo& F, float inv_xy, float inv_z, float x, float y, float z)
{
  const int i0 = (int)(floorf(x * inv_xy) - (float)F.min_b[0]);
  const int i1 = (int)(floorf(y * inv_xy) - (float)F.min_b[1]);
  const int i2 = (int)(floorf(z * inv_z) - (float)F.min_b[2]);
  return (unsigned)(i0 + i1 * F.div_b[0] + i2 * (F.div_b[0] * F.div_b[1]));
}

// cell index of input point i (A2/A3 applied), false if the point is dropped
template <class Src>
__device__ __forceinline__ bool part_cell(const Src& src, const FrameInfo* fi, float inv_xy, float inv_z, unsigned i,
                                          float& x, float& y, float& z, unsigned& cell)
{
  if (!load_point(src, fi, 0, i, x, y, z)) return false;
  cell = part_cell_of(fi[0], inv_xy, inv_z, x, y, z);
  return true;
}

// ---- counts[digit][chunk] of one pass ------------------------------------------------------------
// FROM_SRC: the first pass reads the input; later passes read the records of the previous pass
template <bool FROM_SRC, class Src>
__global__ void __launch_bounds__(kBlock)
k_part_count(Src src, const FrameInfo* fi, float inv_xy, float inv_z, const float4* rec_in, unsigned n, int shift,
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
      cell = __float_as_uint(rec_in[i].w);
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
__global__ void __launch_bounds__(kPartThreads)
k_part_scatter(Src src, const FrameInfo* fi, float inv_xy, float inv_z, const float4* rec_in, unsigned n, int shift,
               int bits, const uint32_t* offsets, unsigned n_chunks, float4* rec_out)
{
  constexpr int kWaves = kPartThreads / kWave;
  __shared__ unsigned short s_cnt[kWaves][kPartDigits];  // records of (wave, digit) so far; then the waves' prefix
  __shared__ unsigned short s_dstart[kPartDigits];       // where the digit starts in the reordered tile
  __shared__ unsigned s_base[kPartDigits];               // where the chunk's next record of a digit goes
  __shared__ unsigned s_scan[kWaves];
  __shared__ float4 s_rec[kPartTile];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), w = tid / kWave;
  const unsigned chunk = blockIdx.x;
  const int nd = 1 << bits;
  const unsigned dmask = (unsigned)nd - 1u;
  s_base[tid] = tid < nd ? offsets[(size_t)tid * n_chunks + chunk] : 0u;
  if (!FROM_SRC) n = (unsigned)fi[0].n_raw;
  const unsigned c_lo = min(n, chunk * (unsigned)kPartChunk);
  const unsigned c_hi = min(n, c_lo + (unsigned)kPartChunk);
  const unsigned long long lt = (1ull << lane) - 1ull;
  unsigned my_tot = 0u;  // records of digit `tid` in the previous tile
  for (unsigned t0 = c_lo; t0 < c_hi; t0 += kPartTile) {
    {
      unsigned* z = reinterpret_cast<unsigned*>(&s_cnt[0][0]);
      for (int i = tid; i < kWaves * kPartDigits / 2; i += kPartThreads) z[i] = 0u;
    }
    __syncthreads();  // (every reader of the previous tile's s_rec, s_dstart and s_base is done)
    s_base[tid] += my_tot;  // (read again only after the barriers below)
    float4 r[kPartIPT];
    unsigned short rank[kPartIPT];
    bool ok[kPartIPT];
    // loads first (all in flight together), ranking afterwards
#pragma unroll
    for (int k = 0; k < kPartIPT; ++k) {
      const unsigned i = t0 + (unsigned)(w * (kWave * kPartIPT) + k * kWave + lane);
      ok[k] = i < c_hi;
      r[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok[k]) {
        if (FROM_SRC) {
          unsigned cell = 0u;
          ok[k] = part_cell(src, fi, inv_xy, inv_z, i, r[k].x, r[k].y, r[k].z, cell);
          r[k].w = __uint_as_float(cell);
        } else {
          r[k] = rec_in[i];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < kPartIPT; ++k) {
      const unsigned d = (__float_as_uint(r[k].w) >> shift) & dmask;
      const unsigned long long same = match_lanes(d, bits, ok[k]);
      const unsigned before = s_cnt[w][d];  // (the wave's batches run in order: only this wave touches its row)
      rank[k] = (unsigned short)(before + (unsigned)__popcll(same & lt));
      __builtin_amdgcn_wave_barrier();
      if (ok[k] && (same & lt) == 0ull) s_cnt[w][d] = (unsigned short)(before + (unsigned)__popcll(same));
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // digit `tid`: prefix over the waves, total; then the digits' starts in the reordered tile
    unsigned run = 0u;
#pragma unroll
    for (int ww = 0; ww < kWaves; ++ww) {
      const unsigned c = s_cnt[ww][tid];
      s_cnt[ww][tid] = (unsigned short)run;
      run += c;
    }
    my_tot = run;
    const unsigned inc = wave_incl_scan(run, lane);
    if (lane == kWave - 1) s_scan[w] = inc;
    __syncthreads();
    unsigned pre = inc - run, tile_n = 0u;
#pragma unroll
    for (int ww = 0; ww < kWaves; ++ww) {
      const unsigned t = s_scan[ww];
      if (ww < w) pre += t;
      tile_n += t;
    }
    s_dstart[tid] = (unsigned short)pre;
    __syncthreads();
    // reorder the tile in LDS
#pragma unroll
    for (int k = 0; k < kPartIPT; ++k) {
      if (!ok[k]) continue;
      const unsigned d = (__float_as_uint(r[k].w) >> shift) & dmask;
      s_rec[(unsigned)s_dstart[d] + (unsigned)s_cnt[w][d] + (unsigned)rank[k]] = r[k];
    }
    __syncthreads();
    // write the runs out
    for (unsigned p = tid; p < tile_n; p += kPartThreads) {
      const float4 q = s_rec[p];
      const unsigned d = (__float_as_uint(q.w) >> shift) & dmask;
      rec_out[(size_t)s_base[d] + (p - (unsigned)s_dstart[d])] = q;
    }
    // (the next tile's first barrier orders its LDS writes against these reads)
  }
}

// ---- bucket ranges of the partitioned records -----------------------------------------------------
// bstart[h] = first record whose (cell >> kPartLow) >= h, for h = 0 .. n_buckets (a binary search per bucket)
__global__ void __launch_bounds__(kBlock)
k_part_bounds(const float4* rec, const FrameInfo* fi, unsigned n_buckets, uint32_t* bstart)
{
  const unsigned m = (unsigned)fi[0].n_raw;  // records that survived A2 (counted by k_bbox)
  const unsigned h = blockIdx.x * kBlock + threadIdx.x;
  if (h > n_buckets) return;
  unsigned lo = 0u, hi = m;
  while (lo < hi) {
    const unsigned mid = lo + ((hi - lo) >> 1);
    if ((__float_as_uint(rec[mid].w) >> kPartLow) < h) lo = mid + 1u; else hi = mid;
  }
  bstart[h] = lo;
}

// ---- centroids ------------------------------------------------------------------------------------
// out: the bucket's voxels in ascending cell order at out[bstart[b] ...] (a bucket has at most as many voxels as
// records), vox_cnt[b] their number.
//
// k_part_finish_small: one wave per bucket.  Empty buckets and buckets of up to 64 records are finished here;
// the others are listed in `big` (n_buckets entries; big_n[0] / big_n[2] count the front / back entries; any order).
__global__ void __launch_bounds__(kFinThreads)
k_part_finish_small(const float4* rec, const uint32_t* bstart, unsigned n_buckets, float4* out, uint32_t* vox_cnt,
                    uint32_t* big, uint32_t* big_n)
{
  __shared__ float4 s_rec[kFinWaves][kWave];
  __shared__ unsigned s_bits[kFinWaves][kPartCells / 32];
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const unsigned cmask = (unsigned)kPartCells - 1u;
  const unsigned b = blockIdx.x * kFinWaves + w;
  if (b >= n_buckets) return;
  const unsigned lo = bstart[b], hi = bstart[b + 1];
  if (hi == lo) {
    if (lane == 0) vox_cnt[b] = 0u;
    return;
  }
  if (hi - lo > (unsigned)kWave) {
    // the longest buckets (shelf rows: 10^5 records) from the front of the list, the others from its back:
    // k_part_finish_big starts at the front, so the long ones are under way while the short ones fill the gaps
    if (lane == 0) {
      if (hi - lo > (unsigned)kFinLong) big[atomicAdd(&big_n[0], 1u)] = b;
      else big[n_buckets - 1u - atomicAdd(&big_n[2], 1u)] = b;
    }
    return;
  }
  const bool ok = lo + (unsigned)lane < hi;
  const float4 r = ok ? rec[lo + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned key = __float_as_uint(r.w) & cmask;
  const unsigned long long same = match_lanes(key, kPartLow, ok);
  const bool first = ok && (same & lt) == 0ull;
  s_rec[w][lane] = r;
  if (lane < kPartCells / 32) s_bits[w][lane] = 0u;
  __builtin_amdgcn_wave_barrier();
  if (first) atomicOr(&s_bits[w][key >> 5], 1u << (key & 31u));
  __builtin_amdgcn_wave_barrier();
  if (first) {
    // the group's points in lane order = ascending input index
    float ax = 0.f, ay = 0.f, az = 0.f;
    unsigned cnt = 0u;
    unsigned long long m = same;
    while (m) {
      const int j = __ffsll((long long)m) - 1;
      m &= m - 1ull;
      const float4 q = s_rec[w][j];
      ax = ax + q.x;
      ay = ay + q.y;
      az = az + q.z;
      ++cnt;
    }
    unsigned pos = 0u;
    const unsigned kw = key >> 5;
#pragma unroll
    for (int i = 0; i < kPartCells / 32; ++i) {
      const unsigned bw = s_bits[w][i];
      if ((unsigned)i < kw) pos += (unsigned)__popc(bw);
      if ((unsigned)i == kw) pos += (unsigned)__popc(bw & ((1u << (key & 31u)) - 1u));
    }
    const float fn = (float)cnt;
    out[lo + pos] = make_float4(ax / fn, ay / fn, az / fn, 0.f);
  }
  const unsigned long long firsts = __builtin_amdgcn_ballot_w64(first);
  if (lane == 0) vox_cnt[b] = (unsigned)__popcll(firsts);
}

// k_part_finish_big: the listed buckets, each by a whole workgroup; the workgroups of a fixed-size grid take the
// next bucket of the list until it is used up (big_n[0] / big_n[2] = entries at the front / at the back of big[0 ..
// n_buckets), big_n[1] = next entry, zero at launch).
__global__ void __launch_bounds__(kFinThreads)
k_part_finish_big(const float4* rec, const uint32_t* bstart, unsigned n_buckets, const uint32_t* big, uint32_t* big_n,
                  float4* out, uint32_t* vox_cnt)
{
  __shared__ float4 s_rec[kFinTile];
  __shared__ unsigned short s_cnt[kFinWaves][kPartCells];
  __shared__ unsigned short s_cstart[kPartCells];
  __shared__ unsigned short s_ctot[kPartCells];
  __shared__ unsigned s_scan[kFinWaves];
  __shared__ unsigned s_next;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), w = tid / kWave;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const unsigned cmask = (unsigned)kPartCells - 1u;
  const unsigned n_long = big_n[0], n_big = n_long + big_n[2];
  for (;;) {
    __syncthreads();  // (s_next, s_scan: the readers of the previous round are done)
    if (tid == 0) s_next = atomicAdd(&big_n[1], 1u);
    __syncthreads();
    const unsigned at = s_next;
    if (at >= n_big) break;  // (the same for every thread; every workgroup gets here: the list is finite)
    const unsigned b = at < n_long ? big[at] : big[n_buckets - 1u - (at - n_long)];
    const unsigned lo = bstart[b], hi = bstart[b + 1];
    // this thread's cells: tid and tid + 256 (a shelf row is a run of adjacent cells: one chain per lane)
    float ax[2] = {0.f, 0.f}, ay[2] = {0.f, 0.f}, az[2] = {0.f, 0.f};
    unsigned an[2] = {0u, 0u};
    float4 r[8];
    bool ok[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned i = lo + (unsigned)(w * (kWave * 8) + k * kWave + lane);
      ok[k] = i < hi;
      r[k] = ok[k] ? rec[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (unsigned t0 = lo; t0 < hi; t0 += kFinTile) {
      {
        unsigned* z = reinterpret_cast<unsigned*>(&s_cnt[0][0]);
        for (int i = tid; i < kFinWaves * kPartCells / 2; i += kFinThreads) z[i] = 0u;
      }
      __syncthreads();  // (also: every reader of the previous tile's s_rec, s_cstart and s_ctot is done)
      unsigned short rank[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        rank[k] = 0;
        if (__builtin_amdgcn_ballot_w64(ok[k]) == 0ull) continue;
        const unsigned d = __float_as_uint(r[k].w) & cmask;
        const unsigned long long same = match_lanes(d, kPartLow, ok[k]);
        const unsigned before = s_cnt[w][d];
        rank[k] = (unsigned short)(before + (unsigned)__popcll(same & lt));
        __builtin_amdgcn_wave_barrier();
        if (ok[k] && (same & lt) == 0ull) s_cnt[w][d] = (unsigned short)(before + (unsigned)__popcll(same));
        __builtin_amdgcn_wave_barrier();
      }
      __syncthreads();
      // cells 2 tid and 2 tid + 1: prefix over the waves, then over the cells
      unsigned tot[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = 2 * tid + j;
        unsigned run = 0u;
#pragma unroll
        for (int ww = 0; ww < kFinWaves; ++ww) {
          const unsigned v = s_cnt[ww][c];
          s_cnt[ww][c] = (unsigned short)run;
          run += v;
        }
        tot[j] = run;
      }
      const unsigned both = tot[0] + tot[1];
      const unsigned inc = wave_incl_scan(both, lane);
      if (lane == kWave - 1) s_scan[w] = inc;
      __syncthreads();
      unsigned pre = inc - both;
#pragma unroll
      for (int ww = 0; ww < kFinWaves; ++ww)
        if (ww < w) pre += s_scan[ww];
      s_cstart[2 * tid] = (unsigned short)pre;
      s_cstart[2 * tid + 1] = (unsigned short)(pre + tot[0]);
      s_ctot[2 * tid] = (unsigned short)tot[0];
      s_ctot[2 * tid + 1] = (unsigned short)tot[1];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (!ok[k]) continue;
        const unsigned d = __float_as_uint(r[k].w) & cmask;
        s_rec[(unsigned)s_cstart[d] + (unsigned)s_cnt[w][d] + (unsigned)rank[k]] = r[k];
      }
      // the next tile's records: on their way while the chains run
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const unsigned i = t0 + (unsigned)kFinTile + (unsigned)(w * (kWave * 8) + k * kWave + lane);
        ok[k] = i < hi;
        r[k] = ok[k] ? rec[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      __syncthreads();
      // one chain per cell
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = tid + j * kFinThreads;
        const unsigned s = s_cstart[c], e = s + (unsigned)s_ctot[c];
        for (unsigned p = s; p < e; p += 4u) {
          const float4 q0 = s_rec[p];
          const float4 q1 = s_rec[min(p + 1u, e - 1u)];
          const float4 q2 = s_rec[min(p + 2u, e - 1u)];
          const float4 q3 = s_rec[min(p + 3u, e - 1u)];
          ax[j] = ax[j] + q0.x;
          ay[j] = ay[j] + q0.y;
          az[j] = az[j] + q0.z;
          if (p + 1u < e) {
            ax[j] = ax[j] + q1.x;
            ay[j] = ay[j] + q1.y;
            az[j] = az[j] + q1.z;
          }
          if (p + 2u < e) {
            ax[j] = ax[j] + q2.x;
            ay[j] = ay[j] + q2.y;
            az[j] = az[j] + q2.z;
          }
          if (p + 3u < e) {
            ax[j] = ax[j] + q3.x;
            ay[j] = ay[j] + q3.y;
            az[j] = az[j] + q3.z;
          }
        }
        an[j] += e - s;
      }
      // (the next tile's first barrier keeps its LDS writes behind these reads)
    }
    // occupied cells in ascending order: cells [0, 256) by tid, then cells [256, 512)
    unsigned base = lo;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool occ = an[j] != 0u;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(occ);
      __syncthreads();  // (s_scan: the readers of the previous use are done)
      if (lane == 0) s_scan[w] = (unsigned)__popcll(m);
      __syncthreads();
      unsigned pos = base + (unsigned)__popcll(m & lt);
      unsigned all = 0u;
#pragma unroll
      for (int ww = 0; ww < kFinWaves; ++ww) {
        const unsigned t = s_scan[ww];
        if (ww < w) pos += t;
        all += t;
      }
      if (occ) {
        const float fn = (float)an[j];
        out[pos] = make_float4(ax[j] / fn, ay[j] / fn, az[j] / fn, 0.f);
      }
      base += all;
    }
    if (tid == 0) vox_cnt[b] = base - lo;
  }
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
