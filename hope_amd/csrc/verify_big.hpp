// verify_big.hpp -- A10..A12 for the LARGE clusters (a floor, a wall, a table top): the same sums as
// k_cluster_verify_b (kernels.hpp), member by member in ascending index as the reference adds them
// (getCloudZInfo utilities.cpp:751-769, calNormalMean :257-282, normalAnalysis :151-255), but evaluated by
// exact_sum.hpp instead of a chain of dependent adds: every member's terms are computed in parallel by as
// many workgroups as the cluster has 1024-member super blocks, and what is left of the chain is one scalar
// compare-and-add per super block.  The 570 296-member floor of the 10^8-point scene: 6.5 ms -> see DESIGN.md.
//
// Launch order per call (all clusters of `big_from` members and more at once):
//   k_big_list                              the clusters, their super-block spans, initial state
//   pass 1: k_big_terms<1>  k_xs_scan  k_xs_records  k_xs_chain  k_big_gate<1>   (z, nx, ny, |nz|; bbox; count)
//   pass 2: k_big_terms<2>  ...                      k_big_gate<2>               (covariance sums)
//   pass 3: k_big_terms<3>  ...                      k_big_gate<3>               (split sums -> ClusterRec)
#pragma once
#include "exact_sum.hpp"
#include "kernels.hpp"

namespace hope {

struct BigInfo {
  unsigned cluster;
  int size;
};
struct BigState {
  unsigned ct;     // members with a finite z
  unsigned bb[4];  // ordered encodings of min x, min y, max x, max y
  unsigned na;     // members on the positive side of the split
  int alive;       // still a candidate after the last gate
  float m0, m1;    // mean normal (x, y)
  float ax, ay;    // principal axis
};
constexpr int kBigSums = 16;  // chain sums per cluster: [0..3] pass 1, [4..6] pass 2, [8..13] pass 3

// Before the cluster count comes back to the host: how many clusters take this path, and their super blocks
// (cnt / flags are per seed rank: k_segment_sizes, k_segment_flags).  g->n_big, g->n_big_sb zeroed by the host.
__global__ void __launch_bounds__(kBlock)
k_big_count(const uint32_t* cnt, const U3* flags, unsigned n, int big_from, Globals* g)
{
  const unsigned r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= n || !flags[r].b || cnt[r] < (unsigned)big_from) return;
  atomicAdd(&g->n_big, 1u);
  atomicAdd(&g->n_big_sb, (cnt[r] + xs::kSuper - 1) / xs::kSuper);
}

// cursor[0] = clusters listed, cursor[1] = super blocks handed out (both zeroed by the host)
__global__ void __launch_bounds__(kBlock)
k_big_list(const ClusterRec* rec, unsigned n_cl, int big_from, unsigned* cursor, xs::Span* spans, BigInfo* info,
           unsigned* sb_owner, BigState* state)
{
  const unsigned c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= n_cl) return;
  const int sz = rec[c].size;
  if (sz < big_from) return;
  const unsigned n_sb = ((unsigned)sz + xs::kSuper - 1) / xs::kSuper;
  const unsigned j = atomicAdd(&cursor[0], 1u);
  const unsigned off = atomicAdd(&cursor[1], n_sb);
  spans[j].sb_off = off;
  spans[j].n_sb = n_sb;
  info[j].cluster = c;
  info[j].size = sz;
  for (unsigned i = 0; i < n_sb; ++i) sb_owner[off + i] = j;
  BigState st;
  st.ct = 0u;
  st.bb[0] = st.bb[1] = 0xffffffffu;
  st.bb[2] = st.bb[3] = 0u;
  st.na = 0u;
  st.alive = 1;
  st.m0 = st.m1 = st.ax = st.ay = 0.f;
  state[j] = st;
}

__device__ __forceinline__ double wave_sum_double(float a)
{
  double v = (double)a;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += (unsigned)__shfl_xor((int)v, off);
  return v;
}
__device__ __forceinline__ float wave_minmax(float v, bool is_min)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(v, off);
    v = is_min ? fminf(v, o) : fmaxf(v, o);
  }
  return v;
}

// One workgroup per super block: thread t owns members k * 256 + t (k = 0..3) of it, so wave w holds block
// 4k + w of the super block in lane order.  Writes the terms of this pass (chain-major, 0 beyond the
// cluster's end and for clusters that have already failed) and the double sums per block that predict the
// exponent of the running sum.
template <int PASS>
__global__ void __launch_bounds__(kBlock)
k_big_terms(const xs::Span* spans, const BigInfo* info, const unsigned* sb_owner, BigState* state,
            const uint32_t* members, const uint32_t* cl_first, const float4* fit_pt, const float4* fit_nrm,
            float* terms, size_t chain_stride, double* bsum, size_t bsum_stride, unsigned* part)
{
  __shared__ unsigned s_part[kBlock / kWave][6];
  constexpr int kChains = PASS == 1 ? 4 : (PASS == 2 ? 3 : 6);
  const unsigned sb = blockIdx.x;
  const unsigned j = sb_owner[sb];
  const BigInfo bi = info[j];
  const unsigned sbi = sb - spans[j].sb_off;
  const uint32_t* m = members + cl_first[bi.cluster];
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  const BigState st = state[j];
  unsigned ct = 0, na = 0;
  float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = (int)(sbi * xs::kSuper) + k * kBlock + (int)threadIdx.x;
    const bool in = e < bi.size && (PASS == 1 || st.alive);
    float v[kChains];
#pragma unroll
    for (int q = 0; q < kChains; ++q) v[q] = 0.f;
    if (in) {
      const unsigned idx = m[e];
      const float4 nn = fit_nrm[idx];
      if (PASS == 1) {
        const float4 p = fit_pt[idx];
        const bool fin = isfinite(p.z);
        v[0] = fin ? p.z : 0.f;
        v[1] = nn.x;
        v[2] = nn.y;
        v[3] = fabsf(nn.z);
        ct += fin ? 1u : 0u;
        mnx = fminf(mnx, p.x);
        mny = fminf(mny, p.y);
        mxx = fmaxf(mxx, p.x);
        mxy = fmaxf(mxy, p.y);
      } else if (PASS == 2) {
        const float tx = nn.x - st.m0, ty = nn.y - st.m1;
        v[0] = tx * tx;
        v[1] = tx * ty;
        v[2] = ty * ty;
      } else {
        const float tx = nn.x - st.m0, ty = nn.y - st.m1;
        const bool side = (st.ax * tx + st.ay * ty > 0.f);
        const float nz = fabsf(nn.z);
        v[0] = side ? nn.x : 0.f;
        v[1] = side ? nn.y : 0.f;
        v[2] = side ? nz : 0.f;
        v[3] = side ? 0.f : nn.x;
        v[4] = side ? 0.f : nn.y;
        v[5] = side ? 0.f : nz;
        na += side ? 1u : 0u;
      }
    }
    const size_t at = (size_t)sb * xs::kSuper + k * kBlock + threadIdx.x;
    const size_t blk = (size_t)sb * xs::kSub + k * 4 + w;
#pragma unroll
    for (int q = 0; q < kChains; ++q) {
      terms[q * chain_stride + at] = v[q];
      const double d = wave_sum_double(v[q]);
      if (lane == 0) bsum[q * bsum_stride + blk] = d;
    }
  }
  // count, bounding box and split count of this super block (k_big_gate adds the super blocks up: thousands
  // of atomics on one cluster's record would queue up behind each other)
  if (PASS == 1) {
    ct = wave_sum_u32(ct);
    mnx = wave_minmax(mnx, true);
    mny = wave_minmax(mny, true);
    mxx = wave_minmax(mxx, false);
    mxy = wave_minmax(mxy, false);
    if (lane == 0) {
      s_part[w][0] = ct;
      s_part[w][1] = f2ord(mnx);
      s_part[w][2] = f2ord(mny);
      s_part[w][3] = f2ord(mxx);
      s_part[w][4] = f2ord(mxy);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned r[5] = {s_part[0][0], s_part[0][1], s_part[0][2], s_part[0][3], s_part[0][4]};
      for (int i = 1; i < kBlock / kWave; ++i) {
        r[0] += s_part[i][0];
        r[1] = min(r[1], s_part[i][1]);
        r[2] = min(r[2], s_part[i][2]);
        r[3] = max(r[3], s_part[i][3]);
        r[4] = max(r[4], s_part[i][4]);
      }
      for (int i = 0; i < 5; ++i) part[(size_t)sb * 6 + i] = r[i];
    }
  } else if (PASS == 3) {
    na = wave_sum_u32(na);
    if (lane == 0) s_part[w][5] = na;
    __syncthreads();
    if (threadIdx.x == 0) part[(size_t)sb * 6 + 5] = s_part[0][5] + s_part[1][5] + s_part[2][5] + s_part[3][5];
  }
}

// One wave per large cluster: the scalar step between two passes (the same functions as k_cluster_verify_b),
// after adding up what the super blocks counted.
template <int PASS>
__global__ void __launch_bounds__(kWave)
k_big_gate(const xs::Span* spans, const BigInfo* info, BigState* state, const unsigned* part, const float* sums,
           float th_angle, ClusterRec* rec)
{
  const unsigned j = blockIdx.x;
  const int lane = threadIdx.x;
  const xs::Span sp = spans[j];
  BigState& st = state[j];
  if (PASS == 1) {
    unsigned ct = 0, b0 = 0xffffffffu, b1 = 0xffffffffu, b2 = 0u, b3 = 0u;
    for (unsigned i = lane; i < sp.n_sb; i += kWave) {
      const unsigned* p = part + (size_t)(sp.sb_off + i) * 6;
      ct += p[0];
      b0 = min(b0, p[1]);
      b1 = min(b1, p[2]);
      b2 = max(b2, p[3]);
      b3 = max(b3, p[4]);
    }
    ct = wave_sum_u32(ct);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      b0 = min(b0, (unsigned)__shfl_xor((int)b0, off));
      b1 = min(b1, (unsigned)__shfl_xor((int)b1, off));
      b2 = max(b2, (unsigned)__shfl_xor((int)b2, off));
      b3 = max(b3, (unsigned)__shfl_xor((int)b3, off));
    }
    if (lane == 0) {
      st.ct = ct;
      st.bb[0] = b0;
      st.bb[1] = b1;
      st.bb[2] = b2;
      st.bb[3] = b3;
    }
  } else if (PASS == 3) {
    unsigned na = 0;
    for (unsigned i = lane; i < sp.n_sb; i += kWave) na += part[(size_t)(sp.sb_off + i) * 6 + 5];
    na = wave_sum_u32(na);
    if (lane == 0) st.na = na;
  }
  if (lane != 0) return;
  const float* s = sums + (size_t)j * kBigSums;
  const int sz = info[j].size;
  if (PASS == 1) {
    float m0 = 0.f, m1 = 0.f;
    st.alive = verify_mean_gate(sz, s[1], s[2], s[3], th_angle, m0, m1) ? 1 : 0;
    st.m0 = m0;
    st.m1 = m1;
  } else if (PASS == 2) {
    if (st.alive) {
      float ax, ay;
      verify_axis(sz, s[4], s[5], s[6], ax, ay);
      st.ax = ax;
      st.ay = ay;
    }
  } else {
    int pass = 0;
    if (st.alive) pass = verify_split(s + 8, st.na, (unsigned)sz - st.na, th_angle);
    ClusterRec& r = rec[info[j].cluster];
    r.z_mean = s[0] / (float)st.ct;
    r.min_x = ord2f(st.bb[0]);
    r.min_y = ord2f(st.bb[1]);
    r.max_x = ord2f(st.bb[2]);
    r.max_y = ord2f(st.bb[3]);
    r.pass = pass;
  }
}

}  // namespace hope
