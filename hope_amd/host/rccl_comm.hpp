// The collectives of multi_gpu.hpp over RCCL: one process, one thread per GPU, one communicator per GPU from
// ncclCommInitAll; all buffers in device memory.  run_cloud_on_gpus is what hope_node calls.
#pragma once
#include <string>

#include "multi_gpu.hpp"

namespace hope {

// Splits the host cloud (n points, 12-byte stride) into n_gpus contiguous ranges, runs segment_cloud_multi with one
// thread, one ctx (cfg with device = rank) and one RCCL communicator per GPU, and returns rank 0's result.
// Status: that of rank 0's call, or the first error (< 0, text in err).
int run_cloud_on_gpus(int n_gpus, const hope_config& cfg, const float* xyz_host, size_t n, int apply_filter_transform,
                      const float* t_xyz, const float* q_xyzw, MultiResult& out, std::string& err);

}  // namespace hope
