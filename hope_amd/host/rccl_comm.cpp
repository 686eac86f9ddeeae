// RCCL backend of multi_gpu.hpp (see rccl_comm.hpp).  Host C++ against the HIP runtime API and RCCL; no kernels here.
#include "rccl_comm.hpp"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <thread>

namespace hope {
namespace {

#define HIP_OK(x)                                                                              \
  do {                                                                                         \
    const hipError_t e__ = (x);                                                                \
    if (e__ != hipSuccess) throw std::runtime_error(std::string(#x ": ") + hipGetErrorString(e__)); \
  } while (0)
#define NCCL_OK(x)                                                                               \
  do {                                                                                           \
    const ncclResult_t r__ = (x);                                                                \
    if (r__ != ncclSuccess) throw std::runtime_error(std::string(#x ": ") + ncclGetErrorString(r__)); \
  } while (0)

// Device buffers of a backend, kept for re-use until the backend goes away: an exchange sequence asks for some
// twenty buffers, and hipMalloc / hipFree are device-wide synchronisations that N rank threads of one process
// would take turns in.
class DevicePool {
 public:
  explicit DevicePool(int device) : device_(device) {}
  ~DevicePool()
  {
    (void)hipSetDevice(device_);
    for (auto& b : all_) (void)hipFree(b.second);
  }
  void* take(size_t bytes)
  {
    bytes = bytes ? bytes : 1;
    size_t best = free_.size();
    for (size_t i = 0; i < free_.size(); ++i)
      if (free_[i].first >= bytes && (best == free_.size() || free_[i].first < free_[best].first)) best = i;
    if (best != free_.size() && free_[best].first <= 2 * bytes + 4096) {
      void* p = free_[best].second;
      free_.erase(free_.begin() + (long)best);
      return p;
    }
    void* p = nullptr;
    HIP_OK(hipSetDevice(device_));
    HIP_OK(hipMalloc(&p, bytes));
    all_.emplace_back(bytes, p);
    return p;
  }
  void give(void* p)
  {
    for (const auto& b : all_)
      if (b.second == p) {
        free_.push_back(b);
        return;
      }
  }

 private:
  int device_;
  std::vector<std::pair<size_t, void*>> all_, free_;
};

class RcclComm : public Comm {
 public:
  RcclComm(ncclComm_t comm, int rank, int world, int device)
      : comm_(comm), rank_(rank), world_(world), device_(device), pool_(device)
  {
    HIP_OK(hipSetDevice(device_));
    HIP_OK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  }
  ~RcclComm() override { (void)hipSetDevice(device_); (void)hipStreamDestroy(stream_); }
  int rank() const override { return rank_; }
  int world() const override { return world_; }
  int on_device() const override { return 1; }
  void* alloc(size_t bytes) override { return pool_.take(bytes); }
  void release(void* p) override { pool_.give(p); }
  void put(void* d, const void* s, size_t b) override { if (b) HIP_OK(hipMemcpy(d, s, b, hipMemcpyHostToDevice)); }
  void get(void* d, const void* s, size_t b) override { if (b) HIP_OK(hipMemcpy(d, s, b, hipMemcpyDeviceToHost)); }
  // device to device, and a fill: on the backend's own stream and complete on return (the library's stream and this
  // one are non-blocking streams, which the null stream does not order)
  void copy(void* d, const void* s, size_t b) override
  {
    if (!b) return;
    HIP_OK(hipMemcpyAsync(d, s, b, hipMemcpyDeviceToDevice, stream_));
    HIP_OK(hipStreamSynchronize(stream_));
  }
  void fill_ff(void* d, size_t b) override
  {
    if (!b) return;
    HIP_OK(hipMemsetAsync(d, 0xff, b, stream_));
    HIP_OK(hipStreamSynchronize(stream_));
  }
  void all_reduce_max_f32(float* buf, size_t n) override
  {
    NCCL_OK(ncclAllReduce(buf, buf, n, ncclFloat, ncclMax, comm_, stream_));
    HIP_OK(hipStreamSynchronize(stream_));
  }
  void all_reduce_sum_i64(int64_t* buf, size_t n) override
  {
    NCCL_OK(ncclAllReduce(buf, buf, n, ncclInt64, ncclSum, comm_, stream_));
    HIP_OK(hipStreamSynchronize(stream_));
  }
  void all_gather_v(const void* send, void* recv, const std::vector<size_t>& bytes) override
  {
    // pieces of different sizes: one broadcast per piece, grouped (xGMI is point to point: each piece travels
    // from its owner to the seven others, which is what a ring all-gather moves as well)
    NCCL_OK(ncclGroupStart());
    size_t at = 0;
    for (int r = 0; r < world_; ++r) {
      if (bytes[r]) NCCL_OK(ncclBroadcast(send, static_cast<char*>(recv) + at, bytes[r], ncclChar, r, comm_, stream_));
      at += bytes[r];
    }
    NCCL_OK(ncclGroupEnd());
    HIP_OK(hipStreamSynchronize(stream_));
  }
  void all_to_all_v(const float* send, const std::vector<int64_t>& sc, float* recv, const std::vector<int64_t>& rc) override
  {
    NCCL_OK(ncclGroupStart());
    size_t so = 0, ro = 0;
    for (int r = 0; r < world_; ++r) {
      if (sc[r]) NCCL_OK(ncclSend(send + so, (size_t)sc[r], ncclFloat, r, comm_, stream_));
      if (rc[r]) NCCL_OK(ncclRecv(recv + ro, (size_t)rc[r], ncclFloat, r, comm_, stream_));
      so += (size_t)sc[r];
      ro += (size_t)rc[r];
    }
    NCCL_OK(ncclGroupEnd());
    HIP_OK(hipStreamSynchronize(stream_));
  }

 private:
  ncclComm_t comm_;
  int rank_, world_, device_;
  hipStream_t stream_ = nullptr;
  DevicePool pool_;
};

// threads + device buffers on one GPU: ThreadComm's rendezvous, hipMemcpy as the transport
class DeviceThreadComm : public Comm {
 public:
  DeviceThreadComm(ThreadGroup& g, int rank) : g_(g), rank_(rank), pool_(0) {}
  int rank() const override { return rank_; }
  int world() const override { return g_.world(); }
  int on_device() const override { return 1; }
  void* alloc(size_t bytes) override { return pool_.take(bytes); }
  void release(void* p) override { pool_.give(p); }
  void put(void* d, const void* s, size_t b) override { if (b) HIP_OK(hipMemcpy(d, s, b, hipMemcpyHostToDevice)); }
  void get(void* d, const void* s, size_t b) override { if (b) HIP_OK(hipMemcpy(d, s, b, hipMemcpyDeviceToHost)); }
  void copy(void* d, const void* s, size_t b) override
  {
    if (!b) return;
    HIP_OK(hipMemcpy(d, s, b, hipMemcpyDeviceToDevice));
    HIP_OK(hipDeviceSynchronize());  // (complete on return, whichever stream reads it next)
  }
  void fill_ff(void* d, size_t b) override { if (b) { HIP_OK(hipMemset(d, 0xff, b)); HIP_OK(hipDeviceSynchronize()); } }
  void all_reduce_max_f32(float* buf, size_t n) override
  {
    std::vector<float> mine(n), all(n * (size_t)world());
    get(mine.data(), buf, n * sizeof(float));
    exchange(mine.data(), all.data(), n * sizeof(float));
    for (size_t i = 0; i < n; ++i)
      for (int k = 0; k < world(); ++k) mine[i] = std::max(mine[i], all[(size_t)k * n + i]);
    put(buf, mine.data(), n * sizeof(float));
  }
  void all_reduce_sum_i64(int64_t* buf, size_t n) override
  {
    std::vector<int64_t> mine(n), all(n * (size_t)world());
    get(mine.data(), buf, n * sizeof(int64_t));
    exchange(mine.data(), all.data(), n * sizeof(int64_t));
    for (size_t i = 0; i < n; ++i) {
      int64_t t = 0;
      for (int k = 0; k < world(); ++k) t += all[(size_t)k * n + i];
      mine[i] = t;
    }
    put(buf, mine.data(), n * sizeof(int64_t));
  }
  void all_gather_v(const void* send, void* recv, const std::vector<size_t>& bytes) override
  {
    g_.ptr()[rank_] = send;
    g_.barrier();
    size_t at = 0;
    for (int k = 0; k < world(); ++k) {
      copy(static_cast<char*>(recv) + at, g_.ptr()[k], bytes[k]);
      at += bytes[k];
    }
    g_.barrier();
  }
  void all_to_all_v(const float* send, const std::vector<int64_t>& sc, float* recv, const std::vector<int64_t>& rc) override
  {
    g_.ptr()[rank_] = send;
    g_.cnt()[rank_] = sc.data();
    g_.barrier();
    size_t at = 0;
    for (int s = 0; s < world(); ++s) {
      const int64_t* c = g_.cnt()[s];
      size_t off = 0;
      for (int d = 0; d < rank_; ++d) off += (size_t)c[d];
      copy(recv + at, static_cast<const float*>(g_.ptr()[s]) + off, sizeof(float) * (size_t)rc[s]);
      at += (size_t)rc[s];
    }
    g_.barrier();
  }

 private:
  void exchange(const void* mine, void* all, size_t bytes)  // host buffers: every rank's `mine`, in rank order
  {
    g_.ptr()[rank_] = mine;
    g_.barrier();
    for (int k = 0; k < world(); ++k) std::memcpy(static_cast<char*>(all) + (size_t)k * bytes, g_.ptr()[k], bytes);
    g_.barrier();
  }
  ThreadGroup& g_;
  int rank_;
  DevicePool pool_;
};

}  // namespace

int run_cloud_as_ranks_on_gpu0(int n_ranks, const hope_config& cfg, const float* xyz_host, size_t n, int apply,
                               MultiResult& out, std::string& err)
{
  if (n_ranks < 1 || n_ranks > 8) {
    err = "run_cloud_as_ranks_on_gpu0: 1 to 8 ranks";
    return HOPE_E_INVALID;
  }
  ThreadGroup group(n_ranks);
  std::vector<int> status((size_t)n_ranks, 0);
  std::vector<std::string> errs((size_t)n_ranks);
  std::vector<MultiResult> results((size_t)n_ranks);
  std::vector<hope_ctx*> ctxs((size_t)n_ranks, nullptr);
  for (int r = 0; r < n_ranks; ++r) {  // (created up front: a rank that cannot start would leave the others waiting)
    hope_config c = cfg;
    c.device = 0;
    if (hope_create(&c, &ctxs[r]) != HOPE_OK) {
      for (hope_ctx* x : ctxs) if (x) hope_destroy(x);
      err = "hope_create failed";
      return HOPE_E_HIP;
    }
  }
  std::vector<std::thread> threads;
  for (int r = 0; r < n_ranks; ++r)
    threads.emplace_back([&, r] {
      try {
        HIP_OK(hipSetDevice(0));
        DeviceThreadComm comm(group, r);
        const size_t lo = n * (size_t)r / (size_t)n_ranks, hi = n * (size_t)(r + 1) / (size_t)n_ranks;
        status[r] = segment_cloud_multi(ctxs[r], comm, xyz_host + 3 * lo, hi - lo, apply, cfg, results[r], errs[r]);
      } catch (const std::exception& e) {
        group.abort();  // (the other ranks' next rendezvous throws instead of waiting for this one)
        status[r] = HOPE_E_HIP;
        errs[r] = e.what();
      }
    });
  for (std::thread& t : threads) t.join();
  for (hope_ctx* x : ctxs) hope_destroy(x);
  for (int r = 0; r < n_ranks; ++r)
    if (status[r] < 0) {
      err = "rank " + std::to_string(r) + ": " + errs[r];
      return status[r];
    }
  out = std::move(results[0]);
  return status[0];
}

int run_cloud_on_gpus(int n_gpus, const hope_config& cfg, const float* xyz_host, size_t n, int apply,
                      const float* t_xyz, const float* q_xyzw, MultiResult& out, std::string& err)
{
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have < n_gpus || n_gpus < 1) {
    err = "run_cloud_on_gpus: " + std::to_string(n_gpus) + " GPUs asked for, " + std::to_string(have) + " visible";
    return HOPE_E_INVALID;
  }
  std::vector<ncclComm_t> comms((size_t)n_gpus);
  std::vector<int> devs((size_t)n_gpus);
  for (int r = 0; r < n_gpus; ++r) devs[r] = r;
  const ncclResult_t nr = ncclCommInitAll(comms.data(), n_gpus, devs.data());
  if (nr != ncclSuccess) {
    err = std::string("ncclCommInitAll: ") + ncclGetErrorString(nr);
    return HOPE_E_INVALID;
  }
  std::vector<int> status((size_t)n_gpus, 0);
  std::vector<std::string> errs((size_t)n_gpus);
  std::vector<MultiResult> results((size_t)n_gpus);
  std::vector<std::thread> threads;
  std::once_flag abort_once;
  std::atomic<bool> aborted(false);
  for (int r = 0; r < n_gpus; ++r)
    threads.emplace_back([&, r] {
      hope_ctx* ctx = nullptr;
      try {
        hope_config c = cfg;
        c.device = r;
        if (hope_create(&c, &ctx) != HOPE_OK) throw std::runtime_error("hope_create failed on device " + std::to_string(r));
        if (t_xyz && q_xyzw) hope_set_pose(ctx, t_xyz, q_xyzw);
        RcclComm comm(comms[r], r, n_gpus, r);
        const size_t lo = n * (size_t)r / (size_t)n_gpus, hi = n * (size_t)(r + 1) / (size_t)n_gpus;
        status[r] = segment_cloud_multi(ctx, comm, xyz_host + 3 * lo, hi - lo, apply, c, results[r], errs[r]);
      } catch (const std::exception& e) {
        // a HIP / RCCL error on this rank (a library step that fails is agreed on inside segment_cloud_multi): the
        // other ranks may be waiting for it inside a collective, and RCCL has no timeout -- abort every communicator
        // once, which makes their pending operations fail and their threads come back
        status[r] = HOPE_E_HIP;
        errs[r] = e.what();
        std::call_once(abort_once, [&] {
          aborted = true;
          for (int k = 0; k < n_gpus; ++k) ncclCommAbort(comms[k]);
        });
      }
      if (ctx) hope_destroy(ctx);
    });
  for (std::thread& t : threads) t.join();
  if (!aborted)
    for (int r = 0; r < n_gpus; ++r) ncclCommDestroy(comms[r]);
  for (int r = 0; r < n_gpus; ++r)
    if (status[r] < 0) {
      err = "rank " + std::to_string(r) + ": " + errs[r];
      return status[r];
    }
  out = std::move(results[0]);
  return status[0];
}

}  // namespace hope
