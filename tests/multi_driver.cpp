// Test driver: segment_cloud_multi (hope_amd/host/multi_gpu.cpp) on N threads over ThreadComm, every rank with a ctx of
// the library this file is linked against (tests/Makefile: the CPU oracle).  usage: multi_driver <cloud.f32> <ranks>
// <apply 0|1> <allow_bands 0|1> [fail_rank fail_step [throw 0|1]]; prints one JSON line with rank 0's merged result.
// With a fault injected (a library step of one rank fails, or that rank throws): every rank must come back, the exit
// code is 3 and stderr names each rank's status.
#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <thread>
#include <vector>

#include "multi_gpu.hpp"

int main(int argc, char** argv)
{
  if (argc != 5 && argc != 7 && argc != 8) { std::fprintf(stderr, "usage: multi_driver <cloud.f32> <ranks> <apply> <allow_bands> [fail_rank fail_step [throw]]\n"); return 2; }
  const int fail_rank = argc >= 7 ? std::atoi(argv[5]) : -1, fail_step = argc >= 7 ? std::atoi(argv[6]) : -1;
  const bool do_throw = argc == 8 && std::atoi(argv[7]) != 0;
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
      try {
        if (do_throw && r == fail_rank) throw std::runtime_error("injected exception");
        status[r] = hope::segment_cloud_multi(ctx, comm, xyz + 3 * lo, hi - lo, apply, cfg, res[r], errs[r], allow,
                                              do_throw ? -1 : fail_rank, fail_step);
      } catch (const std::exception& e) {
        group.abort();  // nobody waits for this rank any more
        status[r] = -101;
        errs[r] = e.what();
      }
      hope_destroy(ctx);
    });
  for (std::thread& t : th) t.join();
  bool failed = false;
  for (int r = 0; r < world; ++r)
    if (status[r] < 0) { std::fprintf(stderr, "rank %d: %d %s\n", r, status[r], errs[r].c_str()); failed = true; }
  if (failed) return fail_rank >= 0 ? 3 : 1;
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
