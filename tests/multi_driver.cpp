// Test driver: segment_cloud_multi (hope_amd/host/multi_gpu.cpp) on N threads over ThreadComm, every rank with a ctx of
// the library this file is linked against (tests/Makefile: the CPU oracle).  usage: multi_driver <cloud.f32> <ranks>
// <apply 0|1> <allow_bands 0|1>; prints one JSON line with rank 0's merged result.
#include <cstdio>
#include <fstream>
#include <thread>
#include <vector>

#include "multi_gpu.hpp"

int main(int argc, char** argv)
{
  if (argc != 5) { std::fprintf(stderr, "usage: multi_driver <cloud.f32> <ranks> <apply> <allow_bands>\n"); return 2; }
  std::ifstream in(argv[1], std::ios::binary);
  std::vector<char> raw((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  const size_t n = raw.size() / 12;
  const float* xyz = reinterpret_cast<const float*>(raw.data());
  const int world = std::atoi(argv[2]), apply = std::atoi(argv[3]);
  const bool allow = std::atoi(argv[4]) != 0;
  hope_config cfg;
  hope_config_default(&cfg);
  hope::ThreadGroup group(world);
  std::vector<hope::MultiResult> res((size_t)world);
  std::vector<int> status((size_t)world, 0);
  std::vector<std::string> errs((size_t)world);
  std::vector<std::thread> th;
  for (int r = 0; r < world; ++r)
    th.emplace_back([&, r] {
      hope_ctx* ctx = nullptr;
      if (hope_create(&cfg, &ctx) != HOPE_OK) { status[r] = -100; return; }
      hope::ThreadComm comm(group, r);
      const size_t lo = n * (size_t)r / (size_t)world, hi = n * (size_t)(r + 1) / (size_t)world;
      status[r] = hope::segment_cloud_multi(ctx, comm, xyz + 3 * lo, hi - lo, apply, cfg, res[r], errs[r], allow);
      hope_destroy(ctx);
    });
  for (std::thread& t : th) t.join();
  for (int r = 0; r < world; ++r)
    if (status[r] < 0) { std::fprintf(stderr, "rank %d: %d %s\n", r, status[r], errs[r].c_str()); return 1; }
  const hope::MultiResult& m = res[0];
  std::printf("{\"path\": \"%s\", \"status\": %d, \"n_raw\": %lld, \"n_vox\": %lld, \"n_fit\": %lld, \"n_segments\": %d, \"n_planes\": %d, "
              "\"received\": [", m.path.c_str(), status[0], (long long)m.n_raw_total, (long long)m.n_vox, (long long)m.n_fit,
              m.n_segments, m.n_planes);
  for (size_t i = 0; i < m.received.size(); ++i) std::printf("%s%lld", i ? ", " : "", (long long)m.received[i]);
  std::printf("], \"clusters\": [");
  for (size_t i = 0; i < m.clusters.size(); ++i) {
    const hope_cluster& c = m.clusters[i];
    std::printf("%s[%d, %d, %.9g, %.9g, %.9g, %.9g, %.9g, %d]", i ? ", " : "", c.size, c.pass, c.z_mean, c.min_x, c.min_y, c.max_x,
                c.max_y, c.seed_rank);
  }
  unsigned long long h = 1469598103934665603ull;   // FNV-1a of the labels
  for (int32_t l : m.labels) { h ^= (unsigned)l; h *= 1099511628211ull; }
  std::printf("], \"n_labels\": %zu, \"labels_fnv\": \"%llx\"}\n", m.labels.size(), h);
  return 0;
}
