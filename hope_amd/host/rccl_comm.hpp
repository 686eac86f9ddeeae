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

// Test harness for boxes with ONE GPU: n_ranks threads, every rank with its own ctx on device 0 and DEVICE buffers, the
// collectives done with device-to-device copies between the threads (no RCCL).  Exercises the device-buffer paths of
// segment_cloud_multi and of the library for more than one rank; `hope_node` runs it when HOPE_NODE_RANKS_ON_GPU0=N.
int run_cloud_as_ranks_on_gpu0(int n_ranks, const hope_config& cfg, const float* xyz_host, size_t n,
                               int apply_filter_transform, MultiResult& out, std::string& err);

}  // namespace hope
