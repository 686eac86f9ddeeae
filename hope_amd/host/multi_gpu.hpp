// One large cloud on several GPUs from C++ (no counterpart in the reference, which is one
// process and one thread).  The same orchestration as hope_amd/sharding.py::segment_cloud_sharded, over a small
// collective interface with two backends:
//   * RcclComm (rccl_comm.cpp): one thread per GPU in one process, ncclCommInitAll, buffers in device memory --
//     what `hope_node <file> ply` uses when HOPE_NODE_GPUS=N;
//   * ThreadComm (below): the same collectives between threads on host memory, for ctxs whose library takes
//     host buffers (the CPU tests run the orchestration on the oracle with 2 and 3 ranks).
// Ranks hold contiguous ranges of the cloud in rank order.  Steps and exchanges: DESIGN.md section 7.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "hope_gpu.h"

namespace hope {

// Collectives and buffer handling of one rank.  Buffers handed to the collectives live where alloc() puts them
// (device memory for RCCL, host memory for threads); every call returns with its result complete.
class Comm {
 public:
  virtual ~Comm() {}
  virtual int rank() const = 0;
  virtual int world() const = 0;
  virtual int on_device() const = 0;                     // the on_device flag of buffers from alloc()
  virtual void* alloc(size_t bytes) = 0;
  virtual void release(void* p) = 0;
  virtual void put(void* dst, const void* host_src, size_t bytes) = 0;   // host -> buffer
  virtual void get(void* host_dst, const void* src, size_t bytes) = 0;   // buffer -> host
  virtual void copy(void* dst, const void* src, size_t bytes) = 0;       // buffer -> buffer
  virtual void fill_ff(void* dst, size_t bytes) = 0;                     // every byte 0xff (int32 -1)
  virtual void all_reduce_max_f32(float* buf, size_t n) = 0;             // in place
  virtual void all_reduce_sum_i64(int64_t* buf, size_t n) = 0;           // in place
  // recv = the ranks' pieces in rank order; bytes[r] = size of rank r's piece (known to every rank)
  virtual void all_gather_v(const void* send, void* recv, const std::vector<size_t>& bytes) = 0;
  // float elements: send_counts[d] go to rank d (pieces in rank order in `send`), recv_counts[s] arrive from s
  virtual void all_to_all_v(const float* send, const std::vector<int64_t>& send_counts, float* recv,
                            const std::vector<int64_t>& recv_counts) = 0;
};

struct MultiResult {                    // filled on every rank (the records are small)
  std::string path;                     // "z-bands" | "replicated" | "one rank" | "empty"
  int64_t n_raw_total = 0, n_vox = 0, n_fit = 0;
  int n_segments = 0, n_planes = 0;
  std::vector<hope_cluster> clusters;   // the whole cloud's clusters in the reference's order (ascending seed rank)
  std::vector<int32_t> labels;          // cluster of every fit point, or -1
  std::vector<int64_t> sent, received;  // points this rank routed to / got from every rank
  std::vector<int32_t> layer_owner;     // rank of every z layer
};

// Owner rank of every z layer (same rule as sharding.py::band_partition): cuts at empty layers nearest to equal
// population when there are any, else at equal population.
std::vector<int32_t> band_partition(const std::vector<int64_t>& hist, int world);

// Collective: every rank calls it with its own ctx (created on its own device) and its share of the cloud in HOST
// memory (n points, stride 12).  Returns HOPE_OK / HOPE_EMPTY / HOPE_NO_NORMAL_FIT or < 0 (text in err).
// A library step that fails on one rank does not leave the others waiting: the statuses travel with every exchange of
// sizes, and all ranks return at the next one (the failing rank its own status, the others HOPE_E_PEER).  A backend
// that throws (HIP / RCCL errors) is the caller's to handle: abort the group (ThreadGroup::abort, ncclCommAbort).
// fail_rank / fail_step: fault injection for the tests -- library step number fail_step of rank fail_rank reports
// HOPE_E_NOMEM instead of running.
int segment_cloud_multi(hope_ctx* ctx, Comm& comm, const float* xyz_host, size_t n, int apply_filter_transform,
                        const hope_config& cfg, MultiResult& out, std::string& err, bool allow_bands = true,
                        int fail_rank = -1, int fail_step = -1);

// ---- threads on host memory -------------------------------------------------------------------------------------
class ThreadGroup {
 public:
  explicit ThreadGroup(int world) : world_(world), ptr_(world), cnt_(world) {}
  int world() const { return world_; }
  // throws when the group has been aborted (a rank left the collective call with an exception): nobody waits for
  // a rank that will not come
  void barrier()
  {
    std::unique_lock<std::mutex> lk(m_);
    if (aborted_) throw std::runtime_error("the rank group was aborted");
    const unsigned gen = gen_;
    if (++waiting_ == world_) { waiting_ = 0; ++gen_; cv_.notify_all(); }
    else cv_.wait(lk, [&] { return gen != gen_ || aborted_; });
    if (aborted_) throw std::runtime_error("the rank group was aborted");
  }
  void abort()
  {
    std::lock_guard<std::mutex> lk(m_);
    aborted_ = true;
    cv_.notify_all();
  }
  std::vector<const void*>& ptr() { return ptr_; }
  std::vector<const int64_t*>& cnt() { return cnt_; }

 private:
  int world_, waiting_ = 0;
  unsigned gen_ = 0;
  bool aborted_ = false;
  std::mutex m_;
  std::condition_variable cv_;
  std::vector<const void*> ptr_;
  std::vector<const int64_t*> cnt_;
};

class ThreadComm : public Comm {
 public:
  ThreadComm(ThreadGroup& g, int rank) : g_(g), rank_(rank) {}
  int rank() const override { return rank_; }
  int world() const override { return g_.world(); }
  int on_device() const override { return 0; }
  void* alloc(size_t bytes) override { return std::malloc(bytes ? bytes : 1); }
  void release(void* p) override { std::free(p); }
  void put(void* d, const void* s, size_t b) override { if (b) std::memcpy(d, s, b); }
  void get(void* d, const void* s, size_t b) override { if (b) std::memcpy(d, s, b); }
  void copy(void* d, const void* s, size_t b) override { if (b) std::memcpy(d, s, b); }
  void fill_ff(void* d, size_t b) override { if (b) std::memset(d, 0xff, b); }
  void all_reduce_max_f32(float* buf, size_t n) override
  {
    std::vector<float> r(buf, buf + n);
    g_.ptr()[rank_] = buf;
    g_.barrier();
    for (int k = 0; k < world(); ++k) {
      const float* o = static_cast<const float*>(g_.ptr()[k]);
      for (size_t i = 0; i < n; ++i) if (o[i] > r[i]) r[i] = o[i];
    }
    g_.barrier();
    std::copy(r.begin(), r.end(), buf);
    g_.barrier();
  }
  void all_reduce_sum_i64(int64_t* buf, size_t n) override
  {
    std::vector<int64_t> r(n, 0);
    g_.ptr()[rank_] = buf;
    g_.barrier();
    for (int k = 0; k < world(); ++k) {
      const int64_t* o = static_cast<const int64_t*>(g_.ptr()[k]);
      for (size_t i = 0; i < n; ++i) r[i] += o[i];
    }
    g_.barrier();
    std::copy(r.begin(), r.end(), buf);
    g_.barrier();
  }
  void all_gather_v(const void* send, void* recv, const std::vector<size_t>& bytes) override
  {
    g_.ptr()[rank_] = send;
    g_.barrier();
    size_t at = 0;
    for (int k = 0; k < world(); ++k) {
      if (bytes[k]) std::memcpy(static_cast<char*>(recv) + at, g_.ptr()[k], bytes[k]);
      at += bytes[k];
    }
    g_.barrier();
  }
  void all_to_all_v(const float* send, const std::vector<int64_t>& send_counts, float* recv,
                    const std::vector<int64_t>& recv_counts) override
  {
    g_.ptr()[rank_] = send;
    g_.cnt()[rank_] = send_counts.data();
    g_.barrier();
    size_t at = 0;
    for (int s = 0; s < world(); ++s) {
      const int64_t* sc = g_.cnt()[s];
      size_t off = 0;
      for (int d = 0; d < rank_; ++d) off += (size_t)sc[d];
      if (recv_counts[s]) std::memcpy(recv + at, static_cast<const float*>(g_.ptr()[s]) + off, sizeof(float) * (size_t)recv_counts[s]);
      at += (size_t)recv_counts[s];
    }
    g_.barrier();
  }

 private:
  ThreadGroup& g_;
  int rank_;
};

}  // namespace hope
