// Microbenchmark + self-check of voxel_part.hpp on a scene laid out like config 4 (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
//         -I hope_amd/csrc tools/micro/part_bench.hip -o tools/micro/part_bench
//   tools/micro/part_bench [points] [check 0|1]
// 60 % of the points on 40 shelves + the floor (1/41 each), 30 % on vertical panels, 10 % clutter.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>
#include "voxel_part.hpp"
#include "grid_index.hpp"
using namespace hope;

struct Shelf { float x0, x1, y0, y1, z; };
__constant__ Shelf c_shelf[40];

__device__ __forceinline__ unsigned mix(unsigned a, unsigned b)
{
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}
__device__ __forceinline__ float uni(unsigned i, unsigned s) { return ((mix(i, s) >> 8) + 0.5f) * (1.0f / 16777216.0f); }

__global__ void k_gen(float* xyz, unsigned n)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float kind = uni(i, 0), u = uni(i, 2), v = uni(i, 3), wz = uni(i, 4);
  const int pick = min(40, (int)(uni(i, 1) * 41.f));
  const bool fl = pick == 40;
  const Shelf s = c_shelf[min(pick, 39)];
  float x, y, z;
  if (kind < 0.6f) {
    x = fl ? u * 30.f : s.x0 + u * (s.x1 - s.x0);
    y = fl ? v * 30.f : s.y0 + v * (s.y1 - s.y0);
    z = fl ? 0.f : s.z;
  } else if (kind < 0.9f) {
    x = fl ? 0.f : s.x0 + u * (s.x1 - s.x0);
    y = fl ? v * 30.f : s.y0;
    z = fl ? wz * 6.f : wz * s.z;
  } else {
    x = u * 30.f; y = v * 30.f; z = wz * 6.f;
  }
  // noise: sum of four uniforms, sigma about 2 mm, bounded
  const float nx = (uni(i, 5) + uni(i, 6) + uni(i, 7) + uni(i, 8) - 2.f) * 0.0035f;
  const float ny = (uni(i, 9) + uni(i, 10) + uni(i, 11) + uni(i, 12) - 2.f) * 0.0035f;
  const float nz = (uni(i, 13) + uni(i, 14) + uni(i, 15) + uni(i, 16) - 2.f) * 0.0035f;
  xyz[3 * (size_t)i] = x + nx; xyz[3 * (size_t)i + 1] = y + ny; xyz[3 * (size_t)i + 2] = z + nz;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv)
{
  const unsigned N = argc > 1 ? (unsigned)atoll(argv[1]) : 100000000u;
  const bool check = argc > 2 ? atoi(argv[2]) != 0 : N <= 20000000u;
  {
    Shelf sh[40];
    unsigned h = 12345u;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return (h >> 8) * (1.0f / 16777216.0f); };
    for (int i = 0; i < 40; ++i) {
      const float w = 1.f + 3.f * rnd(), d = 0.6f + 1.9f * rnd();
      const float x0 = 0.5f + rnd() * (29.f - w), y0 = 0.5f + rnd() * (29.f - d);
      sh[i] = {x0, x0 + w, y0, y0 + d, 0.4f + (5.0f / 11.f) * (i % 12)};
    }
    CK(hipMemcpyToSymbol(HIP_SYMBOL(c_shelf), sh, sizeof(sh)));
  }
  float* xyz; CK(hipMalloc(&xyz, 12ull * N));
  hipLaunchKernelGGL(k_gen, dim3((N + 255) / 256), dim3(256), 0, 0, xyz, N);
  // grid: coordinates lie in (-0.01, 30.01) x (-0.01, 30.01) x (-0.01, 6.01)
  const float inv_xy = 20.f, inv_z = 50.f;
  FrameInfo hfi{};
  hfi.min_b[0] = hfi.min_b[1] = hfi.min_b[2] = -1;
  hfi.div_b[0] = 602; hfi.div_b[1] = 602; hfi.div_b[2] = 302; hfi.n_raw = (int)N;
  const unsigned long long max_cells = 602ull * 602ull * 302ull;
  int cell_bits = 0; while ((1ull << cell_bits) < max_cells) ++cell_bits;
  FrameInfo* fi; CK(hipMalloc(&fi, sizeof(hfi))); CK(hipMemcpy(fi, &hfi, sizeof(hfi), hipMemcpyHostToDevice));
  Globals* glob; CK(hipMalloc(&glob, sizeof(Globals)));
  CloudSrc src{(const char*)xyz, 12ull, N, 0, 0.f, 0.f};
  const unsigned n_chunks = (N + kPartChunk - 1) / kPartChunk;
  const unsigned n_buckets = (unsigned)(((max_cells - 1ull) >> kPartLow) + 1ull);
  const int high = cell_bits - kPartLow;
  const int n_pass = (high + kPartMaxDigit - 1) / kPartMaxDigit;
  printf("points %u, cell bits %d, passes %d, chunks %u, buckets %u\n", N, cell_bits, n_pass, n_chunks, n_buckets);
  float4* buf[2]; uint32_t *counts, *sums, *bstart, *vox_cnt, *big, *big_n; float4* vox; int* vox_frame;
  const size_t n_cnt_max = (size_t)kPartDigits * n_chunks;
  CK(hipMalloc(&buf[0], 16ull * N)); CK(hipMalloc(&buf[1], 16ull * N));
  CK(hipMalloc(&counts, 4 * (n_cnt_max + 4))); CK(hipMalloc(&sums, 4 * scan_sums_words(std::max(n_cnt_max, (size_t)n_buckets + 1))));
  CK(hipMalloc(&bstart, 4ull * (n_buckets + 4))); CK(hipMalloc(&vox_cnt, 4ull * (n_buckets + 4)));
  CK(hipMalloc(&big, 4ull * (n_buckets + 4))); CK(hipMalloc(&big_n, 16));
  CK(hipMalloc(&vox, 16ull * N)); CK(hipMalloc(&vox_frame, 4ull * N));
  hipEvent_t ev[16]; for (auto& e : ev) hipEventCreate(&e);
  const char* names[16]; int ne = 0;
  int cur = 0;
  auto run = [&](bool timed) {
    ne = 0;
    auto mark = [&](const char* nm) { if (timed) { names[ne] = nm; hipEventRecord(ev[ne++], 0); } };
    int shift = kPartLow;
    mark("start");
    for (int p = 0; p < n_pass; ++p) {
      const int bits = high / n_pass + (p < high % n_pass ? 1 : 0);
      const size_t n_cnt = ((size_t)1 << bits) * n_chunks;
      if (p == 0) {
        hipLaunchKernelGGL((k_part_count<true, CloudSrc>), dim3(n_chunks), dim3(kBlock), 0, 0, src, fi, inv_xy, inv_z, (const float4*)nullptr, N, shift, bits, counts, n_chunks);
        mark("count<src>");
        scan_exclusive_u32(counts, counts, n_cnt, sums, 0);
        mark("scan");
        hipLaunchKernelGGL((k_part_scatter<true, CloudSrc>), dim3(n_chunks), dim3(kPartThreads), 0, 0, src, fi, inv_xy, inv_z, (const float4*)nullptr, N, shift, bits, (const uint32_t*)counts, n_chunks, buf[0]);
        mark("scatter<src>");
        cur = 0;
      } else {
        hipLaunchKernelGGL((k_part_count<false, CloudSrc>), dim3(n_chunks), dim3(kBlock), 0, 0, src, fi, inv_xy, inv_z, (const float4*)buf[cur], N, shift, bits, counts, n_chunks);
        mark("count<rec>");
        scan_exclusive_u32(counts, counts, n_cnt, sums, 0);
        mark("scan");
        hipLaunchKernelGGL((k_part_scatter<false, CloudSrc>), dim3(n_chunks), dim3(kPartThreads), 0, 0, src, fi, inv_xy, inv_z, (const float4*)buf[cur], N, shift, bits, (const uint32_t*)counts, n_chunks, buf[cur ^ 1]);
        mark("scatter<rec>");
        cur ^= 1;
      }
      shift += bits;
    }
    hipLaunchKernelGGL(k_part_bounds, dim3((n_buckets + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, (const float4*)buf[cur], fi, n_buckets, bstart);
    mark("bounds");
    hipMemsetAsync(vox_cnt + n_buckets, 0, 4, 0);
    hipMemsetAsync(big_n, 0, 16, 0);
    hipLaunchKernelGGL(k_part_finish_small, dim3((n_buckets + kFinWaves - 1) / kFinWaves), dim3(kFinThreads), 0, 0, (const float4*)buf[cur], (const uint32_t*)bstart, n_buckets, buf[cur ^ 1], vox_cnt, big, big_n);
    mark("finish small");
    hipLaunchKernelGGL(k_part_finish_big, dim3(kFinGrid), dim3(kFinThreads), 0, 0, (const float4*)buf[cur], (const uint32_t*)bstart, n_buckets, (const uint32_t*)big, big_n, buf[cur ^ 1], vox_cnt);
    mark("finish big");
    scan_exclusive_u32(vox_cnt, vox_cnt, (size_t)n_buckets + 1, sums, 0);
    hipLaunchKernelGGL(k_part_vox_total, dim3(1), dim3(1), 0, 0, fi, (const uint32_t*)vox_cnt, n_buckets, glob);
    hipLaunchKernelGGL(k_part_compact, dim3((n_buckets + 3) / 4), dim3(kBlock), 0, 0, (const float4*)buf[cur ^ 1], (const uint32_t*)bstart, (const uint32_t*)vox_cnt, n_buckets, vox, vox_frame);
    mark("scan+compact");
  };
  run(false);
  CK(hipDeviceSynchronize());
  Globals hg; CK(hipMemcpy(&hg, glob, sizeof(hg), hipMemcpyDeviceToHost));
  printf("voxels %u\n", hg.n_vox_total);
  if (check) {
    std::vector<float> h(3ull * N); CK(hipMemcpy(h.data(), xyz, 12ull * N, hipMemcpyDeviceToHost));
    std::vector<uint32_t> cell(N);
    for (unsigned i = 0; i < N; ++i) {
      const int i0 = (int)(floorf(h[3ull * i] * inv_xy) - (float)hfi.min_b[0]);
      const int i1 = (int)(floorf(h[3ull * i + 1] * inv_xy) - (float)hfi.min_b[1]);
      const int i2 = (int)(floorf(h[3ull * i + 2] * inv_z) - (float)hfi.min_b[2]);
      cell[i] = (unsigned)(i0 + i1 * 602 + i2 * (602 * 602));
      if (i0 < 0 || i0 >= 602 || i1 < 0 || i1 >= 602 || i2 < 0 || i2 >= 302) { printf("point %u outside the grid\n", i); return 1; }
    }
    std::vector<uint32_t> order(N); std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cell[a] < cell[b]; });
    std::vector<float> ref;
    size_t maxrun = 0;
    for (size_t a = 0; a < N;) {
      size_t b = a; float sx = 0.f, sy = 0.f, sz = 0.f;
      while (b < N && cell[order[b]] == cell[order[a]]) { const float* p = &h[3ull * order[b]]; sx = sx + p[0]; sy = sy + p[1]; sz = sz + p[2]; ++b; }
      const float fn = (float)(b - a);
      ref.push_back(sx / fn); ref.push_back(sy / fn); ref.push_back(sz / fn);
      maxrun = std::max(maxrun, b - a);
      a = b;
    }
    printf("reference voxels %zu, largest voxel %zu points\n", ref.size() / 3, maxrun);
    if (ref.size() / 3 != hg.n_vox_total) { printf("MISMATCH: voxel count\n"); return 1; }
    std::vector<float> got(4ull * hg.n_vox_total); CK(hipMemcpy(got.data(), vox, 16ull * hg.n_vox_total, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t v = 0; v < hg.n_vox_total; ++v)
      for (int k = 0; k < 3; ++k) {
        uint32_t a, b; memcpy(&a, &got[4 * v + k], 4); memcpy(&b, &ref[3 * v + k], 4);
        if (a != b && bad++ < 5) printf("  voxel %zu comp %d: %.9g vs %.9g\n", v, k, got[4 * v + k], ref[3 * v + k]);
      }
    printf(bad ? "MISMATCH: %zu components differ\n" : "check: bit-exact (%zu)\n", bad);
    if (bad) return 1;
  }
  // ---- A6 over the occupancy index (grid_index.hpp) on the voxels just made
  {
    hfi.slack = 30.f * 20.f * 2.4e-7f + 1e-6f; hfi.n_raw = (int)N;
    CK(hipMemcpy(fi, &hfi, sizeof(hfi), hipMemcpyHostToDevice));
    const unsigned n_vox = hg.n_vox_total, n_words = (unsigned)(max_cells / 32ull) + 2u;
    uint2* tab; uint32_t *gcnt, *gsums, *flags; float4* normals;
    CK(hipMalloc(&tab, 8ull * n_words)); CK(hipMalloc(&gcnt, 4ull * (n_words + 4))); CK(hipMalloc(&gsums, 4 * scan_sums_words(n_words)));
    CK(hipMalloc(&flags, 4ull * (n_vox + 1))); CK(hipMalloc(&normals, 16ull * n_vox));
    const float radius = (float)(1.01 * 0.05), r2 = (float)((double)radius * (double)radius);
    hipEvent_t g0, g1, g2;
    hipEventCreate(&g0); hipEventCreate(&g1); hipEventCreate(&g2);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(g0, 0);
      hipMemsetAsync(tab, 0, 8ull * n_words, 0);
      hipLaunchKernelGGL(k_grid_mark, dim3((n_vox + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, (const float4*)vox, n_vox, fi, inv_xy, inv_z, tab, glob);
      hipLaunchKernelGGL(k_grid_popc, dim3((n_words + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, (const uint2*)tab, n_words, gcnt);
      scan_exclusive_u32(gcnt, gcnt, n_words, gsums, 0);
      hipLaunchKernelGGL(k_grid_pref, dim3((n_words + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, tab, n_words, (const uint32_t*)gcnt);
      hipEventRecord(g1, 0);
      hipLaunchKernelGGL((k_normals_grid<kNbCap>), dim3((n_vox + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, (const float4*)vox, (const uint2*)tab, fi, inv_xy, inv_z, radius, r2, 2, 0u, 0xffffffffu, n_vox, normals, 0.87039f, flags);
      hipEventRecord(g2, 0);
      CK(hipDeviceSynchronize());
      float a, b; hipEventElapsedTime(&a, g0, g1); hipEventElapsedTime(&b, g1, g2);
      printf("grid build %.3f ms, k_normals_grid %.3f ms", a, b);
      hipEventRecord(g1, 0);
      hipLaunchKernelGGL((k_normals_grid<kNbCap, 1>), dim3((n_vox + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, (const float4*)vox, (const uint2*)tab, fi, inv_xy, inv_z, radius, r2, 2, 0u, 0xffffffffu, n_vox, normals, 0.87039f, (uint32_t*)nullptr);
      hipEventRecord(g2, 0);
      CK(hipDeviceSynchronize());
      hipEventElapsedTime(&b, g1, g2);
      printf(", scan only %.3f ms", b);
      if (rep == 0) {
        std::vector<float> hn(4ull * n_vox); CK(hipMemcpy(hn.data(), normals, 16ull * n_vox, hipMemcpyDeviceToHost));
        size_t hist[7] = {0, 0, 0, 0, 0, 0, 0}, lanes[7] = {0, 0, 0, 0, 0, 0, 0};
        for (size_t w0 = 0; w0 < n_vox; w0 += 64) {
          int mx = 0, act = 0;
          for (size_t i = w0; i < std::min<size_t>(w0 + 64, n_vox); ++i) { const int t = (int)hn[4 * i]; mx = std::max(mx, t); act += t >= 3; }
          const int b = mx < 3 ? 0 : mx <= 8 ? 1 : mx <= 12 ? 2 : mx <= 16 ? 3 : mx <= 32 ? 4 : 5;
          ++hist[b]; lanes[b] += act;
        }
        printf("\n  waves by longest list: <3: %zu, <=8: %zu (%zu lanes), <=12: %zu (%zu), <=16: %zu (%zu), <=32: %zu (%zu), more: %zu (%zu)\n  ",
               hist[0], hist[1], lanes[1], hist[2], lanes[2], hist[3], lanes[3], hist[4], lanes[4], hist[5], lanes[5]);
      }
      hipEventRecord(g1, 0);
      hipLaunchKernelGGL((k_normals_grid<kNbCap, 2>), dim3((n_vox + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, (const float4*)vox, (const uint2*)tab, fi, inv_xy, inv_z, radius, r2, 2, 0u, 0xffffffffu, n_vox, normals, 0.87039f, (uint32_t*)nullptr);
      hipEventRecord(g2, 0);
      CK(hipDeviceSynchronize());
      hipEventElapsedTime(&b, g1, g2);
      printf(", without the eigen solver %.3f ms\n", b);
#define TRY(NAME, ...)                                                                                                  \
      hipEventRecord(g1, 0);                                                                                            \
      hipLaunchKernelGGL((k_normals_grid<__VA_ARGS__>), dim3((n_vox + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, (const float4*)vox, (const uint2*)tab, fi, inv_xy, inv_z, radius, r2, 2, 0u, 0xffffffffu, n_vox, normals, 0.87039f, (uint32_t*)nullptr); \
      hipEventRecord(g2, 0);                                                                                            \
      CK(hipDeviceSynchronize());                                                                                       \
      hipEventElapsedTime(&b, g1, g2);                                                                                  \
      printf("   %s %.3f ms\n", NAME, b);
    }
    CK(hipMemcpy(&hg, glob, sizeof(hg), hipMemcpyDeviceToHost));
    std::vector<uint32_t> hf(n_vox); CK(hipMemcpy(hf.data(), flags, 4ull * n_vox, hipMemcpyDeviceToHost));
    size_t nf = 0; for (auto f : hf) nf += f;
    printf("grid_unsorted %u, fit %zu of %u\n", hg.grid_unsorted, nf, n_vox);
  }
  for (int rep = 0; rep < 2; ++rep) {
    run(true);
    CK(hipDeviceSynchronize());
    float total; hipEventElapsedTime(&total, ev[0], ev[ne - 1]);
    printf("-- run %d: %.3f ms\n", rep, total);
    for (int i = 1; i < ne; ++i) { float ms; hipEventElapsedTime(&ms, ev[i - 1], ev[i]); printf("   %-14s %.3f ms\n", names[i], ms); }
  }
  return 0;
}
