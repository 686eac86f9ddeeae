// hope_hip.hip — host side of the C ABI (include/hope_gpu.h): owns the HIP stream and the
// HBM workspace, launches the kernels of kernels.hpp in pipeline order and keeps the
// results resident until the next call.  No CPU fallback: every stage runs on the GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels.hpp"
#include "depth_records.hpp"
#include "neighbours.hpp"
#include "prims.hpp"
#include "upon_plane.hpp"
#include "voxel_rank.hpp"
#include "hull.hpp"
#include "verify_big.hpp"
#include "voxel_part.hpp"
#include "grid_index.hpp"
#include "neighbours_inst.hpp"

#ifndef HOPE_ABLATE  // (the ablation build changes the kernels' bodies: it instantiates its own)
namespace hope {
HOPE_NEIGHBOUR_KERNELS(HOPE_KERNEL_ELSEWHERE)
}  // namespace hope
#endif

using namespace hope;

static_assert(sizeof(ClusterRec) == sizeof(hope_cluster), "ClusterRec must mirror hope_cluster");

// ---- the no-allocation entry point's scratch (hope_shard_z_histogram_static) --------------------------
// Device memory beyond the caller's input and output that its kernels need: the frame record (pose,
// bounding box, VoxelGrid grid), the scalars k_frame_grid writes, and the 64-bit bins the workgroups
// merge their LDS-private votes into.  A file-scope __device__ array: each GPU holds its own copy at its
// own address, resolved once per device in hope_create (hipGetSymbolAddress) -- in host code the array's
// name is only a stand-in, never a device address.
constexpr int kStaticHistBins = kHistLds;  // 8192 z layers (164 m of th_z = 0.02)
struct StaticScratch {
  FrameInfo fi;
  Globals glob;
  unsigned long long hist[kStaticHistBins];
};
static __device__ __align__(256) unsigned char scratch[(sizeof(StaticScratch) + 255) / 256 * 256];
namespace {

constexpr int kMaxDevices = 64;
StaticScratch* g_scratch_ptr[kMaxDevices] = {};  // per device; written under g_scratch_mutex
std::mutex g_scratch_mutex;
std::mutex g_scratch_use[kMaxDevices];           // held across a call that uses a device's array

struct Pose {
  float t[3];
  float q[4];
};

// Grow-only HBM workspace: blocks are kept across calls, so after the first call of a
// given shape no hipMalloc happens on the hot path.
struct Arena {
  struct Block {
    char* p;
    size_t cap, used;
  };
  std::vector<Block> blocks;
  size_t min_block = (size_t)256 << 20;
  bool failed = false;

  void reset()
  {
    for (Block& b : blocks) b.used = 0;
    failed = false;
  }
  void* alloc(size_t bytes)
  {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    for (Block& b : blocks) {
      if (b.cap - b.used >= bytes) {
        void* r = b.p + b.used;
        b.used += bytes;
        return r;
      }
    }
    Block nb;
    nb.cap = std::max(bytes, min_block);
    nb.used = bytes;
    nb.p = nullptr;
    if (hipMalloc((void**)&nb.p, nb.cap) != hipSuccess || !nb.p) {
      (void)hipGetLastError();
      failed = true;
      return nullptr;
    }
    blocks.push_back(nb);
    return nb.p;
  }
  std::vector<size_t> mark() const
  {
    std::vector<size_t> m;
    for (const Block& b : blocks) m.push_back(b.used);
    return m;
  }
  void rollback(const std::vector<size_t>& m)  // blocks added since the mark become empty again
  {
    for (size_t i = 0; i < blocks.size(); ++i) blocks[i].used = i < m.size() ? m[i] : 0;
  }
  void release()
  {
    for (Block& b : blocks) (void)hipFree(b.p);
    blocks.clear();
  }
};

// Host vectors that HIP copies into and out of on every call live in pinned memory: a copy to or from pageable
// memory goes through the driver's bounce buffer and holds the host (and, at the end of a call, the GPU) for tens of
// microseconds per 100 KB.  Capacity is kept across calls, so the steady state allocates nothing.
template <class T>
struct PinnedAllocator {
  typedef T value_type;
  PinnedAllocator() = default;
  template <class U>
  PinnedAllocator(const PinnedAllocator<U>&) {}
  T* allocate(size_t n)
  {
    void* p = nullptr;
    if (hipHostMalloc(&p, n * sizeof(T), hipHostMallocPortable) != hipSuccess) {
      (void)hipGetLastError();
      throw std::bad_alloc();
    }
    return static_cast<T*>(p);
  }
  void deallocate(T* p, size_t) { (void)hipHostFree(p); }
  template <class U>
  bool operator==(const PinnedAllocator<U>&) const { return true; }
  template <class U>
  bool operator!=(const PinnedAllocator<U>&) const { return false; }
};

struct ArenaScope {  // everything allocated while it lives is given back
  Arena& a;
  std::vector<size_t> m;
  explicit ArenaScope(Arena& ar) : a(ar), m(ar.mark()) {}
  ~ArenaScope() { a.rollback(m); }
};

void* arena_alloc_cb(void* user, size_t bytes) { return static_cast<Arena*>(user)->alloc(bytes); }

inline int bits_for(unsigned long long v)  // bits needed to represent values 0..v
{
  int b = 1;
  while (b < 64 && (v >> b) != 0) ++b;
  return b;
}

inline unsigned nblk(size_t n, int per = kBlock) { return (unsigned)((n + per - 1) / per); }

// Optional per-stage timing with HIP events recorded on the ctx stream (bench.py's roofline
// figures come from here; off by default).
enum Stage {
  ST_BBOX, ST_GRID, ST_VKEYS, ST_VSORT, ST_VRLE, ST_VRANK, ST_VPCOUNT, ST_VPART, ST_VFINISH, ST_CENTROID, ST_COLKEYS, ST_COLSORT, ST_COLTABLE, ST_NORMALS,
  ST_FIT, ST_FITIDX, ST_KNN, ST_KNN_TODO, ST_LABELS, ST_SEGMENTS, ST_VERIFY, ST_COUNT
};
const char* const kStageNames[ST_COUNT] = {
  "k_bbox", "k_frame_grid", "k_voxel_keys", "voxel_sort(rocprim)", "voxel_rle+scan(rocprim)", "k_voxel_rank", "k_part_count", "k_part_scatter", "k_part_finish", "k_centroids",
  "k_column_keys", "column_sort(rocprim)", "column_index", "k_normals", "fit_gate+compact", "fit_column_index", "k_knn",
  "k_knn_todo",
  "labels(union+sweeps)", "segments(scan+sort)", "k_cluster_verify"};
static_assert(sizeof(kStageNames) / sizeof(kStageNames[0]) == ST_COUNT, "one name per stage");

struct Prof {
  int on = 0;
  std::vector<hipEvent_t> ev;
  size_t used = 0;
  struct Rec {
    int st;
    size_t a, b;
    double bytes;
  };
  std::vector<Rec> recs;
  double ms[ST_COUNT] = {};
  double bytes[ST_COUNT] = {};  // algorithmic bytes of the closed regions
  long long launches[ST_COUNT] = {};
};

}  // namespace

struct hope_ctx {
  hope_config cfg;
  float th_theta = 0, th_angle = 0, th_norm = 0;
  float inv_xy = 0, inv_z = 0;
  float radius = 0, r2 = 0;
  int ring = 2;
  hipStream_t stream = nullptr;
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  Arena arena;
  Globals* d_glob = nullptr;
  Globals* h_glob = nullptr;  // pinned, mapped, host-coherent: written by k_publish_globals (or a D2H copy)
  Globals* h_glob_dev = nullptr;  // the device's address of it
  unsigned* h_seq = nullptr;      // sequence number of the last publication, in the same allocation
  unsigned* h_seq_dev = nullptr;
  unsigned mail_seq = 0;
  int mail_poll = 1;          // HOPE_DEBUG_SYNC=memcpy: D2H copy + stream synchronisation instead of the polled mailbox
  StaticScratch* d_scratch = nullptr;  // this device's copy of `scratch` (no-allocation entry point)
  Pose pose;
  std::vector<Pose> frame_poses;
  std::string err;
  double whole_ms = 0, ref_ms = 0;
  int normals_cap_small = 0;  // test knob: force the neighbour-list overflow path
  // options of the staged (multi-GPU) entry points, reset after every call
  const float* forced_bbox = nullptr;  // 6 floats: use this bounding box instead of the input's own
  int stop_after_voxel = 0;            // A5 only
  // hope_segment_downsampled_staged: A6 / A8 for the down-sampled points [own_lo, own_hi) only, stages
  // given from outside instead of computed, early stop after A6 (1) or A8 (2)
  unsigned own_lo = 0u, own_hi = 0xffffffffu;
  const float* inj_normals = nullptr;
  const int32_t* inj_knn = nullptr;
  int inj_on_device = 0;
  int stop_stage = 0;
  int input_is_downsampled = 0;        // the input already is the VoxelGrid output (A6..A12 only)
  long long n_raw_override = -1;
  unsigned knn_fine_limit = 4096;  // more left-over k-NN queries than this build the coarse index (test knob)
  unsigned n_knn_todo = 0;
  int verify_big_from = 32768;     // clusters of this many members and more are verified by verify_big.hpp
                                   // (HOPE_DEBUG_VERIFY_BIG=n: test knob; 0 = never)
  xs::Stats* d_xs_stats = nullptr; // counters of the exact-sum chains of the last call (hope_debug_xs_stats)
  int force_ieee_div = 0;     // test knob HOPE_DEBUG_IEEE_DIV
  int fast_div = 0;           // div_const<true> verified for fast_div_key (the current camera model)
  float fast_div_key[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int ablate = 0;             // timing experiments only (-DHOPE_ABLATE builds read HOPE_DEBUG_ABLATE): wrong results
  int pixel_sort = 0;         // test knob: depth frames through the per-pixel sort (the cloud path's voxel stage)
  int col_sort_z = 0;         // test knob HOPE_DEBUG_COLSORT_Z=1: z cell bits in the column sort key even for VoxelGrid output
  int timeline = 0;           // HOPE_DEBUG_TIMELINE=1: host-side time marks of a call, printed to stderr when it returns
  std::vector<std::pair<const char*, double>> marks;
  std::chrono::steady_clock::time_point mark0;
  int seg_sort = 1;           // per-frame segmented sort of the group records (HOPE_DEBUG_SEGSORT=0: device-wide sort)
  int col_index_lds = 1;      // column index by a per-frame counting sort in LDS when the tables fit (HOPE_DEBUG_COL_INDEX=0: the radix sort)
  int todo_threads = 1;       // HOPE_DEBUG_TODO_THREADS=0: the k-NN leftovers over the coarse index by a wave each
  int knn_direct = 1;         // A6 and A8 read the column index straight from HBM through the caches, one thread per
  int normals_direct = 1;     // point (k_normals_direct, k_knn_direct); HOPE_DEBUG_NEIGHBOURS=tiles: the LDS tile kernels
  int tile_side = 0;          // neighbour tiles: 0 = chosen per call from the points per super tile; HOPE_DEBUG_TILE_SIDE=16|32 forces
  int voxel_part = 1;         // one cloud: A5 by stable partition passes + one wave per 512-cell bucket (voxel_part.hpp;
                              // HOPE_DEBUG_VOXEL_PART=0: keys, radix sort, run lengths, k_centroids_w)
  int grid_index = 1;         // one cloud: A6 over the 3-D occupancy index (grid_index.hpp; HOPE_DEBUG_GRID_INDEX=0: column index)
  int grid_off_once = 0;      // ... not for the call under way: its derived cells do not ascend (set by stage_rest)
  int voxel_rank = 1;         // depth batches: voxels ranked by an LDS bitmap per frame instead of sorting the group
                              // records (HOPE_DEBUG_VOXEL_RANK=0, or any knob that asks for a sort: the sorting path)
  uint32_t* d_grp_cnt = nullptr;  // depth path: groups per chunk, then its exclusive scan
  uint32_t* d_grp_off = nullptr;
  unsigned cpf = 0;               // chunks per frame
  unsigned cpb = 0;               // chunks per band of kChunkRows image rows
  uint4* d_slots = nullptr;       // one-pass depth path: kChunk record slots per chunk (sparse)
  int two_pass = 0;               // test knob HOPE_DEBUG_TWO_PASS: k_bbox_groups + k_group_records
  bool one_pass = false;          // this call's records are in d_slots
  bool dense = false;             // ... densely per frame (k_bbox_records<.., DENSE>), cap_f records at most; else kChunk slots per chunk
  int members_by_frame = 1;       // HOPE_DEBUG_MEMBERS_SORT=1: the member lists by radix sort always; 0: by frame always
  int union_small_tiles = 1;      // HOPE_DEBUG_UNION_TILES=0: a single cloud is united in tiles of 24 576 points too
  int frame_order = 1;            // HOPE_DEBUG_FRAME_ORDER=0: the per-frame kernels take the frames in frame order
  int dense_records = 1;          // HOPE_DEBUG_DENSE_RECORDS=0: the sparse slots + k_group_keys
  int cap_mul = 16;               // HOPE_DEBUG_CAP_MUL: records per chunk that a frame's dense range has room for
  unsigned cap_f = 0;
  uint32_t* d_rec_count = nullptr;
  uint32_t* d_codes = nullptr;
  uint32_t* d_frame_order = nullptr;  // frames by descending record count (dense records, 64..4096 frames), or null
  Prof prof;

  // ---- state of the last call --------------------------------------------------------
  int n_frames = 0;
  int src_kind = 0;  // 1 depth, 2 cloud
  DepthSrc dsrc{};
  CloudSrc csrc{};
  std::vector<FrameInfo, PinnedAllocator<FrameInfo>> h_fi;
  std::vector<ClusterRec, PinnedAllocator<ClusterRec>> h_rec;
  std::vector<hope_max_plane> h_max;
  std::vector<std::vector<int32_t>> h_max_members;  // RT winner: every member cluster, in max_plane_cloud_ order
  std::vector<int> h_nplanes;
  FrameInfo* d_fi = nullptr;
  float4* d_vox = nullptr;
  float4* d_normals = nullptr;
  uint32_t* d_fit_idx = nullptr;
  int* d_knn = nullptr;
  int* d_cid = nullptr;
  uint32_t* d_members = nullptr;
  const float4* d_fit_pt = nullptr;  // fit points in fit order (the members index it)
  uint32_t* d_cl_first = nullptr;
  unsigned n_vox_total = 0, n_fit_total = 0, n_cl_total = 0;
  // F1 (hope_clusters_upon_plane): clusters as ranges of d_upon_xyz / d_upon_src, in output order
  struct UponRange { unsigned first, count; };
  std::vector<UponRange> upon;
  float* d_upon_xyz = nullptr;
  int* d_upon_src = nullptr;
  std::vector<size_t> upon_mark;   // arena fill levels before the last F1 call (rolled back by the next one)
  bool upon_mark_valid = false;

  // ---- sub-batches of a large depth batch ------------------------------------------------
  // Frames are independent, so a batch of many frames is cut into n_sub contiguous ranges, each run
  // by a child context of its own (own stream, own workspace) on a host thread of its own: while one
  // range waits for a count to come back or runs a kernel that cannot fill the chip (a chain of
  // dependent adds, one workgroup per frame), the kernels of the other ranges use the idle CUs.
  int n_sub = 1;                  // hope_config::sub_batches (1: never split).  Measured on MI355X, 1024 frames: four ranges
                                  // 13.7 ms against 12.3 ms unsplit -- the kernels already fill the chip -- so off by default
  int sub_min_frames = 64;        // HOPE_DEBUG_SUB_MIN: frames per range below which a batch is not cut further
  std::vector<hope_ctx*> kids;
  std::vector<int> kid_first;     // last call: kid k holds frames [kid_first[k], kid_first[k+1]); empty: not split
  hope_ctx* upon_kid = nullptr;   // kid that ran the last hope_clusters_upon_plane
};

namespace {

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      char buf__[512];                                                                       \
      snprintf(buf__, sizeof(buf__), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
               __FILE__, __LINE__);                                                          \
      c->err = buf__;                                                                        \
      return HOPE_E_HIP;                                                                     \
    }                                                                                        \
  } while (0)

#define ALLOC(ptr, type, count)                                     \
  do {                                                              \
    ptr = static_cast<type*>(c->arena.alloc(sizeof(type) * (size_t)(count))); \
    if (!ptr) {                                                     \
      c->err = "out of device memory";                              \
      return HOPE_E_NOMEM;                                          \
    }                                                               \
  } while (0)

// exclusive scan of n words with the hand-written kernels of scan.hpp (scratch from the arena); in == out allowed
inline hipError_t arena_scan_u32(hope_ctx* c, const uint32_t* in, uint32_t* out, size_t n, hipStream_t s)
{
  if (n == 0) return hipSuccess;
  uint32_t* sums = static_cast<uint32_t*>(c->arena.alloc(sizeof(uint32_t) * scan_sums_words(n)));
  if (!sums) return hipErrorOutOfMemory;
  scan_exclusive_u32(in, out, n, sums, s);
  return hipGetLastError();
}

// thresholds exactly as the reference derives them (plane_segment.cpp:46-50; host scalars)
void derive_thresholds(hope_ctx* c)
{
  c->th_theta = c->cfg.th_z / c->cfg.th_xy;
  c->th_angle = std::atan(c->th_theta);
  c->th_norm = (float)std::sqrt(1 / (1 + 2 * std::pow((double)c->th_theta, 2)));
  c->inv_xy = 1.0f / c->cfg.th_xy;  // VoxelGrid inverse_leaf_size_
  c->inv_z = 1.0f / c->cfg.th_z;
  const float norm_r = (float)(1.01 * c->cfg.th_xy);  // plane_segment.cpp:695
  const double radius = (double)norm_r;
  c->radius = norm_r;
  c->r2 = static_cast<float>(radius * radius);  // KdTreeFLANN::radiusSearch
  c->ring = (int)std::floor(norm_r * c->inv_xy + 0.01f) + 1;  // = 2 = kHaloNormals for the reference's 1.01 * th_xy
}

// Eigen Quaternion::toRotationMatrix on the un-normalised quaternion (transform.cpp:168-171)
void make_affine(const Pose& p, float* a)
{
  const float x = p.q[0], y = p.q[1], z = p.q[2], w = p.q[3];
  const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  a[0] = 1.f - (tyy + tzz);
  a[1] = txy - twz;
  a[2] = txz + twy;
  a[3] = txy + twz;
  a[4] = 1.f - (txx + tzz);
  a[5] = tyz - twx;
  a[6] = txz - twy;
  a[7] = tyz + twx;
  a[8] = 1.f - (txx + tyy);
  a[9] = p.t[0];
  a[10] = p.t[1];
  a[11] = p.t[2];
}

// The counts that size the next launches come back to the host several times per call.  A D2H copy plus a stream
// synchronisation costs tens of microseconds of idle GPU each time; instead a one-wave kernel, ordered on the stream
// like the copy would be, writes the scalars straight into pinned, host-coherent memory and then a sequence number,
// which the host polls (falling back to a stream synchronisation, which also surfaces any error of an earlier kernel).
__global__ void k_publish_globals(const Globals* d, Globals* h, unsigned* h_seq, unsigned seq)
{
  const unsigned* src = reinterpret_cast<const unsigned*>(d);
  unsigned* dst = reinterpret_cast<unsigned*>(h);
  for (unsigned i = threadIdx.x; i < sizeof(Globals) / sizeof(unsigned); i += blockDim.x) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    __atomic_store_n(h_seq, seq, __ATOMIC_RELEASE);
    __threadfence_system();
  }
}

inline void tl_mark(hope_ctx* c, const char* what)
{
  if (!c->timeline) return;
  const auto now = std::chrono::steady_clock::now();
  if (c->marks.empty()) c->mark0 = now;
  c->marks.emplace_back(what, std::chrono::duration<double, std::micro>(now - c->mark0).count());
}
inline void tl_print(hope_ctx* c)
{
  if (!c->timeline) return;
  double prev = 0.0;
  for (const auto& m : c->marks) {
    fprintf(stderr, "[timeline] %9.1f us (+%8.1f) %s\n", m.second, m.second - prev, m.first);
    prev = m.second;
  }
  c->marks.clear();
}

int sync_globals(hope_ctx* c)
{
  tl_mark(c, "sync_globals: enter");
  struct Leave { hope_ctx* c; ~Leave() { tl_mark(c, "sync_globals: leave"); } } leave{c};
  if (!c->mail_poll) {
    HIPCHK(hipMemcpyAsync(c->h_glob, c->d_glob, sizeof(Globals), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return HOPE_OK;
  }
  const unsigned seq = ++c->mail_seq;
  hipLaunchKernelGGL(k_publish_globals, dim3(1), dim3(kWave), 0, c->stream, c->d_glob, c->h_glob_dev, c->h_seq_dev, seq);
  volatile unsigned* hs = c->h_seq;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (*hs != seq) {
    if ((++spins & 1023u) == 0u &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.05) {
      HIPCHK(hipStreamSynchronize(c->stream));  // (long kernels, or an error on the stream)
      if (*hs != seq) {
        c->err = "the count read-back kernel did not run";
        return HOPE_E_HIP;
      }
      break;
    }
    __builtin_ia32_pause();
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return HOPE_OK;
}

size_t prof_mark(hope_ctx* c)
{
  Prof& p = c->prof;
  if (!p.on) return 0;
  if (p.used == p.ev.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) {
      p.on = 0;
      return 0;
    }
    p.ev.push_back(e);
  }
  (void)hipEventRecord(p.ev[p.used], c->stream);
  return p.used++;
}

void prof_close(hope_ctx* c, int st, size_t a, double alg_bytes = 0.0)
{
  if (!c->prof.on) return;
  const size_t b = prof_mark(c);
  c->prof.recs.push_back(Prof::Rec{st, a, b, alg_bytes});
}

void prof_collect(hope_ctx* c)  // call after the stream has been synchronised
{
  Prof& p = c->prof;
  for (const Prof::Rec& r : p.recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.ev[r.a], p.ev[r.b]) == hipSuccess) {
      p.ms[r.st] += ms;
      p.bytes[r.st] += r.bytes;
      p.launches[r.st] += 1;
    }
  }
  p.recs.clear();
  p.used = 0;
}

// PlaneSegmentRT::getPlane's winner logic over the cluster records (plane_segment.cpp:930-962).
// max_plane_cloud_ is kept as the list of the clusters whose points it concatenates, without any
// limit (combineCloud appends the old max plane AFTER the new cluster, utilities.cpp:1717-1727).
void rt_post(hope_ctx* c)
{
  c->h_max.assign(c->n_frames, hope_max_plane{});
  c->h_max_members.assign(c->n_frames, std::vector<int32_t>());
  c->h_nplanes.assign(c->n_frames, 0);
  for (int f = 0; f < c->n_frames; ++f) {
    const FrameInfo& F = c->h_fi[f];
    hope_max_plane& mp = c->h_max[f];
    std::vector<int32_t>& mem = c->h_max_members[f];
    mp.z = -1000.0f;
    int n_planes = 0;
    for (int ci = 0; ci < F.n_clusters; ++ci) {
      const ClusterRec& r = c->h_rec[F.cl_start + ci];
      if (c->cfg.mode == HOPE_MODE_OFFLINE) {
        if (r.pass) ++n_planes;
        continue;
      }
      const float z_in = r.z_mean;
      if (!(z_in > c->cfg.min_height && z_in < c->cfg.max_height)) continue;
      if (!r.pass) continue;
      ++n_planes;
      int count = r.size;
      const bool merge = c->cfg.aggressive_merge && mp.valid && std::fabs(mp.z - z_in) <= c->cfg.th_z;
      if (merge) count += mp.n_points;
      if (count > mp.n_points) {
        if (merge) mem.insert(mem.begin(), ci); else mem.assign(1, ci);
        mp.valid = 1;
        mp.n_points = count;
        mp.z = z_in;
      }
    }
    mp.n_members = (int32_t)mem.size();
    for (size_t m = 0; m < mem.size() && m < 16; ++m) mp.members[m] = mem[m];
    c->h_nplanes[f] = n_planes;
  }
}

inline bool depth8_ok(const DepthSrc& s)
{
  return (s.w % 8) == 0 && (reinterpret_cast<uintptr_t>(s.depth) % 16) == 0;
}

template <class VK>
void launch_voxel_keys(hope_ctx* c, const DepthSrc& src, size_t n_per_frame, int cell_bits, VK* k_in, uint32_t* v_in)
{
  const int F = c->n_frames;
  if (depth8_ok(src)) {
    const unsigned bx = std::max(1u, std::min<unsigned>(nblk(n_per_frame / 8), 1024u));
    hipLaunchKernelGGL((k_voxel_keys_depth8<VK>), dim3(bx, F), dim3(kBlock), 0, c->stream, src, c->d_fi, c->inv_xy,
                       c->inv_z, cell_bits, F, k_in, v_in);
  } else {
    const unsigned bx = std::min<unsigned>(nblk(n_per_frame), 4096u);
    hipLaunchKernelGGL((k_voxel_keys<DepthSrc, VK>), dim3(bx, F), dim3(kBlock), 0, c->stream, src, c->d_fi,
                       c->inv_xy, c->inv_z, cell_bits, F, k_in, v_in);
  }
}

template <class VK>
void launch_voxel_keys(hope_ctx* c, const CloudSrc& src, size_t n_per_frame, int cell_bits, VK* k_in, uint32_t* v_in)
{
  const int F = c->n_frames;
  const unsigned bx = std::min<unsigned>(nblk(n_per_frame), 4096u);
  hipLaunchKernelGGL((k_voxel_keys<CloudSrc, VK>), dim3(bx, F), dim3(kBlock), 0, c->stream, src, c->d_fi, c->inv_xy,
                     c->inv_z, cell_bits, F, k_in, v_in);
}

// (the group records hold the row and column of a chunk in 16 bits each)
inline bool use_groups(const hope_ctx* c, const DepthSrc& s) { return !c->pixel_sort && s.w <= 65535 && s.h <= 65535; }
inline bool use_groups(const hope_ctx*, const CloudSrc&) { return false; }

inline int launch_bbox(hope_ctx* c, const DepthSrc& src, size_t n_per_frame)
{
  const int F = c->n_frames;
  if (use_groups(c, src)) {
    c->cpb = (unsigned)((src.w + kChunkCols - 1) / kChunkCols);
    c->cpf = c->cpb * (unsigned)((src.h + kChunkRows - 1) / kChunkRows);
    const size_t n_chunks = (size_t)c->cpf * F;
    ALLOC(c->d_grp_cnt, uint32_t, n_chunks + 1);
    ALLOC(c->d_grp_off, uint32_t, n_chunks + 1);
    HIPCHK(hipMemsetAsync(c->d_grp_cnt + n_chunks, 0, sizeof(uint32_t), c->stream));
    const unsigned bx = std::max(1u, std::min<unsigned>(nblk(c->cpf, (kBlock / kWave) * 16), 512u));
    // one pass (records into sparse slots) unless the knob or the size of the slot array says otherwise
    c->one_pass = !c->two_pass && n_chunks * kChunk * sizeof(uint4) <= ((size_t)16 << 30);
    // dense records (no scan of group counts, no k_group_keys) feed k_voxel_rank only: the sort path needs the
    // records in chunk order
    c->cap_f = std::max(c->cpf * (unsigned)c->cap_mul, std::min(c->cpf * (unsigned)kChunk, 4096u));
    c->dense = c->one_pass && c->voxel_rank && c->dense_records && (size_t)c->cap_f * F < ((size_t)1 << 31) &&
               bits_for(c->cap_f) + bits_for(c->cpf) <= 32;  // (k_voxel_rank's sort words: chunk number | record index)
    if (c->one_pass) {
      const bool full = src.w % kChunkCols == 0 && src.h % kChunkRows == 0;
      if (c->dense) {
        ALLOC(c->d_slots, uint4, (size_t)c->cap_f * F);
        ALLOC(c->d_codes, uint32_t, (size_t)c->cap_f * F * 2);  // (cell codes, then chunk numbers)
        ALLOC(c->d_rec_count, uint32_t, (size_t)F * kCountStride);
        HIPCHK(hipMemsetAsync(c->d_rec_count, 0, sizeof(uint32_t) * (size_t)F * kCountStride, c->stream));
      } else {
        ALLOC(c->d_slots, uint4, n_chunks * kChunk);
      }
#define HOPE_BBOX_RECORDS(FAST, FULL, DENSE)                                                                          \
  hipLaunchKernelGGL((k_bbox_records<FAST, FULL, DENSE>), dim3(bx, F), dim3(kBlock), 0, c->stream, src, c->d_fi,     \
                     c->inv_xy, c->inv_z, c->cpf, c->cpb, c->d_grp_cnt, c->d_slots, c->d_glob, c->cap_f,             \
                     c->d_rec_count, c->d_codes)
#define HOPE_BBOX_RECORDS2(FAST, FULL) \
  if (c->dense) HOPE_BBOX_RECORDS(FAST, FULL, true); else HOPE_BBOX_RECORDS(FAST, FULL, false)
      if (c->fast_div) {
        if (full) { HOPE_BBOX_RECORDS2(true, true); } else { HOPE_BBOX_RECORDS2(true, false); }
      } else {
        if (full) { HOPE_BBOX_RECORDS2(false, true); } else { HOPE_BBOX_RECORDS2(false, false); }
      }
#undef HOPE_BBOX_RECORDS2
#undef HOPE_BBOX_RECORDS
      if (c->dense && c->frame_order && F >= 64 && F <= 4096) {
        ALLOC(c->d_frame_order, uint32_t, (size_t)F);
        hipLaunchKernelGGL(k_frame_order, dim3(nblk(F, kBlock / kWave)), dim3(kBlock), 0, c->stream, (const uint32_t*)c->d_rec_count,
                           kCountStride, F, c->d_frame_order);
      }
    } else if (c->fast_div)
      hipLaunchKernelGGL((k_bbox_groups<true, true>), dim3(bx, F), dim3(kBlock), 0, c->stream, src, c->d_fi, c->inv_xy,
                         c->inv_z, c->cpf, c->cpb, c->d_grp_cnt);
    else
      hipLaunchKernelGGL((k_bbox_groups<false, true>), dim3(bx, F), dim3(kBlock), 0, c->stream, src, c->d_fi, c->inv_xy,
                         c->inv_z, c->cpf, c->cpb, c->d_grp_cnt);
  } else if (depth8_ok(src)) {
    const unsigned bx = std::max(1u, std::min<unsigned>(nblk(n_per_frame / 8), 256u));
    hipLaunchKernelGGL(k_bbox_depth8, dim3(bx, F), dim3(kBlock), 0, c->stream, src, c->d_fi);
  } else {
    const unsigned bx = std::max(1u, std::min<unsigned>(nblk(n_per_frame, kBlock * 8), 2048u));
    hipLaunchKernelGGL((k_bbox<DepthSrc>), dim3(bx, F), dim3(kBlock), 0, c->stream, src, c->d_fi);
  }
  return HOPE_OK;
}

// depth path: exclusive scan of the per-chunk group counts; the total lands in Globals
inline int scan_groups(hope_ctx* c, const DepthSrc& src, size_t n_per_frame)
{
  if (!use_groups(c, src) || n_per_frame == 0 || c->dense) return HOPE_OK;
  const size_t n_chunks = (size_t)c->cpf * c->n_frames;
  HIPCHK(arena_scan_u32(c, c->d_grp_cnt, c->d_grp_off, n_chunks + 1, c->stream));
  HIPCHK(hipMemcpyAsync(&c->d_glob->n_groups, c->d_grp_off + n_chunks, sizeof(uint32_t), hipMemcpyDeviceToDevice,
                        c->stream));
  return HOPE_OK;
}
inline int scan_groups(hope_ctx*, const CloudSrc&, size_t) { return HOPE_OK; }

// The dense records of this call cannot be used (a cell outside the code range, a frame with more records than its
// range holds, a grid that k_voxel_rank does not take): count the groups for the two-pass path, whose records come
// in chunk order.  The bounding boxes and survivor counts of k_bbox_records stay.
inline int dense_fallback(hope_ctx* c, const DepthSrc& src, size_t n_per_frame)
{
  const int F = c->n_frames;
  c->dense = false;
  c->one_pass = false;
  const unsigned bx = std::max(1u, std::min<unsigned>(nblk(c->cpf, (kBlock / kWave) * 16), 512u));
  if (c->fast_div)
    hipLaunchKernelGGL((k_bbox_groups<true, false>), dim3(bx, F), dim3(kBlock), 0, c->stream, src, c->d_fi, c->inv_xy,
                       c->inv_z, c->cpf, c->cpb, c->d_grp_cnt);
  else
    hipLaunchKernelGGL((k_bbox_groups<false, false>), dim3(bx, F), dim3(kBlock), 0, c->stream, src, c->d_fi, c->inv_xy,
                       c->inv_z, c->cpf, c->cpb, c->d_grp_cnt);
  int st = scan_groups(c, src, n_per_frame);
  if (st != HOPE_OK) return st;
  return sync_globals(c);
}
inline int dense_fallback(hope_ctx*, const CloudSrc&, size_t) { return HOPE_OK; }

inline int launch_bbox(hope_ctx* c, const CloudSrc& src, size_t n_per_frame)
{
  const unsigned bx = std::max(1u, std::min<unsigned>(nblk(n_per_frame, kBlock * 8), 2048u));
  hipLaunchKernelGGL((k_bbox<CloudSrc>), dim3(bx, c->n_frames), dim3(kBlock), 0, c->stream, src, c->d_fi);
  return HOPE_OK;
}

template <class VK>
void launch_group_records(hope_ctx* c, const DepthSrc& src, unsigned bx, int cell_bits, VK* rkey, uint32_t* rval,
                          uint4* rec)
{
  if (c->fast_div)
    hipLaunchKernelGGL((k_group_records<VK, true>), dim3(bx, c->n_frames), dim3(kBlock), 0, c->stream, src, c->d_fi,
                       c->inv_xy, c->inv_z, cell_bits, c->cpf, c->cpb, c->d_grp_off, rkey, rval, rec);
  else
    hipLaunchKernelGGL((k_group_records<VK, false>), dim3(bx, c->n_frames), dim3(kBlock), 0, c->stream, src, c->d_fi,
                       c->inv_xy, c->inv_z, cell_bits, c->cpf, c->cpb, c->d_grp_off, rkey, rval, rec);
}
template <class VK>
void launch_group_records(hope_ctx*, const CloudSrc&, unsigned, int, VK*, uint32_t*, uint4*) {}

template <class VK>
void launch_centroids_groups(hope_ctx* c, const DepthSrc& src, const VK* uniq, const uint32_t* run_start,
                             const uint32_t* run_cnt, const uint32_t* sval, const uint4* rec, unsigned n_vox,
                             int cell_bits, float4* d_vox, int* d_vox_frame)
{
  if (c->fast_div)
    hipLaunchKernelGGL((k_centroids_groups<VK, true>), dim3(nblk(n_vox)), dim3(kBlock), 0, c->stream, src, c->d_fi,
                       uniq, run_start, run_cnt, sval, rec, n_vox, cell_bits, d_vox, d_vox_frame);
  else
    hipLaunchKernelGGL((k_centroids_groups<VK, false>), dim3(nblk(n_vox)), dim3(kBlock), 0, c->stream, src, c->d_fi,
                       uniq, run_start, run_cnt, sval, rec, n_vox, cell_bits, d_vox, d_vox_frame);
}
template <class VK>
void launch_centroids_groups(hope_ctx*, const CloudSrc&, const VK*, const uint32_t*, const uint32_t*, const uint32_t*,
                             const uint4*, unsigned, int, float4*, int*) {}

inline double src_point_bytes(const DepthSrc&) { return 2.0; }
inline double src_point_bytes(const CloudSrc&) { return 12.0; }

// A5 for depth batches without a sort (voxel_rank.hpp): the voxels of every frame ranked by an LDS bitmap of
// its occupied cells, from k_group_keys' dense (cell, slot) pairs.
int stage_voxel_rank(hope_ctx* c, const DepthSrc& src, size_t n_per_frame, size_t M, int cell_bits, const uint32_t* keys,
                     const uint32_t* vals, float4*& d_vox, int*& d_vox_frame)
{
  const int F = c->n_frames;
  hipStream_t s = c->stream;
  const bool dense = c->dense;  // (keys, vals unused: the records' codes and their own indices; M = the ranges' capacity)
  if (dense) M = (size_t)c->cap_f * F;
  uint32_t *rec_vox, *rec_list, *vox_end, *vox_cnt, *block_first;
  ALLOC(block_first, uint32_t, (size_t)F + 1);
  ALLOC(rec_vox, uint32_t, M);
  ALLOC(rec_list, uint32_t, M);
  ALLOC(vox_end, uint32_t, M);
  ALLOC(vox_cnt, uint32_t, (size_t)F);
  size_t pm = prof_mark(c);
  const unsigned cell_mask = cell_bits >= 32 ? 0xffffffffu : ((1u << cell_bits) - 1u);
  if (dense)
    hipLaunchKernelGGL(k_voxel_rank<true>, dim3(F), dim3(kRankBlock), 0, s, c->d_fi, c->cpf, c->d_grp_off,
                       (const uint32_t*)c->d_codes, (const uint32_t*)nullptr, cell_mask, rec_vox, rec_list, vox_end,
                       vox_cnt, c->cap_f, (const uint32_t*)c->d_rec_count, c->inv_xy, c->inv_z, bits_for(c->cap_f),
                       (const uint32_t*)c->d_frame_order);
  else
    hipLaunchKernelGGL(k_voxel_rank<false>, dim3(F), dim3(kRankBlock), 0, s, c->d_fi, c->cpf, c->d_grp_off, keys, vals,
                       cell_mask, rec_vox, rec_list, vox_end, vox_cnt, 0u, (const uint32_t*)nullptr, c->inv_xy, c->inv_z, 0,
                       (const uint32_t*)nullptr);
  hipLaunchKernelGGL(k_frame_vox_scan, dim3(1), dim3(kRankBlock), 0, s, c->d_fi, F, vox_cnt, block_first, c->d_glob,
                     dense ? (const uint32_t*)c->d_rec_count : (const uint32_t*)nullptr);
  prof_close(c, ST_VRANK, pm, (8.0 + 4.0 + 4.0) * (double)M);
  int st = sync_globals(c);
  if (st != HOPE_OK) return st;
  const double n_rec = dense ? (double)c->h_glob->n_groups : (double)M;  // (dense: counted by k_frame_vox_scan)
  if (dense && c->prof.on && !c->prof.recs.empty()) c->prof.recs.back().bytes = (8.0 + 4.0 + 4.0) * n_rec;
  const unsigned n_vox = c->h_glob->n_vox_total;
  c->n_vox_total = n_vox;
  if (n_vox == 0) return HOPE_OK;
  ALLOC(d_vox, float4, n_vox);
  ALLOC(d_vox_frame, int, n_vox);
  pm = prof_mark(c);
  const dim3 grid((c->h_glob->max_vox + kXcds - 1u) / kXcds * kXcds);  // (blocks of kBlock voxels, frame by frame; dealt to the XCDs by the kernel)
#define HOPE_CENTROIDS_LISTS(FAST, DENSE)                                                                              \
  hipLaunchKernelGGL((k_centroids_lists<FAST, DENSE>), grid, dim3(kBlock), 0, s, src, c->d_fi, F, block_first, c->cpf, \
                     c->d_grp_off, rec_list, vox_end, c->d_slots, d_vox, d_vox_frame, c->cap_f, bits_for(c->cap_f))
  if (c->fast_div) {
    if (dense) HOPE_CENTROIDS_LISTS(true, true); else HOPE_CENTROIDS_LISTS(true, false);
  } else {
    if (dense) HOPE_CENTROIDS_LISTS(false, true); else HOPE_CENTROIDS_LISTS(false, false);
  }
#undef HOPE_CENTROIDS_LISTS
  prof_close(c, ST_CENTROID, pm, (4.0 + 16.0) * n_rec + 2.0 * (double)n_per_frame * F + (28.0 + 4.0) * n_vox);
  return HOPE_OK;
}
int stage_voxel_rank(hope_ctx*, const CloudSrc&, size_t, size_t, int, const uint32_t*, const uint32_t*, float4*&, int*&)
{
  return HOPE_E_INVALID;  // (never called: clouds do not take the group path)
}

// A5 for one cloud by stable partition passes (voxel_part.hpp).
template <class Src>
int stage_voxel_part(hope_ctx* c, const Src& src, size_t N, int cell_bits, float4*& d_vox, int*& d_vox_frame)
{
  hipStream_t s = c->stream;
  const int high = cell_bits - kPartLow;
  const int n_pass = (high + kPartMaxDigit - 1) / kPartMaxDigit;
  const unsigned n_chunks = nblk(N, kPartChunk);
  const unsigned n_buckets = (unsigned)(((c->h_glob->max_cells - 1ull) >> kPartLow) + 1ull);
  float4* buf[2];
  uint32_t *counts, *sums, *bstart, *vox_cnt, *big, *big_n;
  const size_t n_cnt_max = (size_t)kPartDigits * n_chunks;
  ALLOC(buf[0], float4, N);
  ALLOC(buf[1], float4, N);
  ALLOC(counts, uint32_t, n_cnt_max + 4);
  ALLOC(sums, uint32_t, scan_sums_words(std::max(n_cnt_max, (size_t)n_buckets + 1)));
  ALLOC(bstart, uint32_t, (size_t)n_buckets + 4);
  ALLOC(vox_cnt, uint32_t, (size_t)n_buckets + 4);
  ALLOC(big, uint32_t, (size_t)n_buckets + 4);
  ALLOC(big_n, uint32_t, 4);
  // algorithmic bytes of a kernel = its input read once + its output written once: the count of a pass needs the
  // point (first pass) or the 4-byte cell of a record; a scatter pass reads the point or the 16-byte record and
  // writes the 16-byte record
  size_t pm;
  int cur = 0, shift = kPartLow;
  for (int p = 0; p < n_pass; ++p) {
    const int bits = high / n_pass + (p < high % n_pass ? 1 : 0);
    const size_t n_cnt = ((size_t)1 << bits) * n_chunks;
    pm = prof_mark(c);
    if (p == 0) {
      hipLaunchKernelGGL((k_part_count<true, Src>), dim3(n_chunks), dim3(kBlock), 0, s, src, c->d_fi, c->inv_xy, c->inv_z,
                         (const float4*)nullptr, (unsigned)N, shift, bits, counts, n_chunks);
      scan_exclusive_u32(counts, counts, n_cnt, sums, s);
      prof_close(c, ST_VPCOUNT, pm, src_point_bytes(src) * (double)N + 8.0 * n_cnt);
      pm = prof_mark(c);
      hipLaunchKernelGGL((k_part_scatter<true, Src>), dim3(n_chunks), dim3(kPartThreads), 0, s, src, c->d_fi, c->inv_xy,
                         c->inv_z, (const float4*)nullptr, (unsigned)N, shift, bits, (const uint32_t*)counts, n_chunks,
                         buf[0]);
      prof_close(c, ST_VPART, pm, (src_point_bytes(src) + 16.0) * (double)N);
      cur = 0;
    } else {
      hipLaunchKernelGGL((k_part_count<false, Src>), dim3(n_chunks), dim3(kBlock), 0, s, src, c->d_fi, c->inv_xy,
                         c->inv_z, (const float4*)buf[cur], (unsigned)N, shift, bits, counts, n_chunks);
      scan_exclusive_u32(counts, counts, n_cnt, sums, s);
      prof_close(c, ST_VPCOUNT, pm, 4.0 * (double)N + 8.0 * n_cnt);
      pm = prof_mark(c);
      hipLaunchKernelGGL((k_part_scatter<false, Src>), dim3(n_chunks), dim3(kPartThreads), 0, s, src, c->d_fi, c->inv_xy,
                         c->inv_z, (const float4*)buf[cur], (unsigned)N, shift, bits, (const uint32_t*)counts, n_chunks,
                         buf[cur ^ 1]);
      prof_close(c, ST_VPART, pm, 32.0 * (double)N);
      cur ^= 1;
    }
    shift += bits;
  }
  pm = prof_mark(c);
  hipLaunchKernelGGL(k_part_bounds, dim3(nblk((size_t)n_buckets + 1)), dim3(kBlock), 0, s, (const float4*)buf[cur],
                     c->d_fi, n_buckets, bstart);
  float4* tmp = buf[cur ^ 1];
  HIPCHK(hipMemsetAsync(vox_cnt + n_buckets, 0, sizeof(uint32_t), s));
  HIPCHK(hipMemsetAsync(big_n, 0, 4 * sizeof(uint32_t), s));
  hipLaunchKernelGGL(k_part_finish_small, dim3(nblk(n_buckets, kFinWaves)), dim3(kFinThreads), 0, s,
                     (const float4*)buf[cur], (const uint32_t*)bstart, n_buckets, tmp, vox_cnt, big, big_n);
  hipLaunchKernelGGL(k_part_finish_big, dim3(kFinGrid), dim3(kFinThreads), 0, s, (const float4*)buf[cur],
                     (const uint32_t*)bstart, n_buckets, (const uint32_t*)big, big_n, tmp, vox_cnt);
  scan_exclusive_u32(vox_cnt, vox_cnt, (size_t)n_buckets + 1, sums, s);  // now: the buckets' first voxels
  hipLaunchKernelGGL(k_part_vox_total, dim3(1), dim3(1), 0, s, c->d_fi, (const uint32_t*)vox_cnt, n_buckets, c->d_glob);
  int st = sync_globals(c);
  if (st != HOPE_OK) return st;
  const unsigned n_vox = c->h_glob->n_vox_total;
  c->n_vox_total = n_vox;
  if (n_vox == 0) {
    prof_close(c, ST_VFINISH, pm);
    return HOPE_OK;
  }
  ALLOC(d_vox, float4, n_vox);
  ALLOC(d_vox_frame, int, n_vox);
  hipLaunchKernelGGL(k_part_compact, dim3(nblk(n_buckets, kBlock / kWave)), dim3(kBlock), 0, s, (const float4*)tmp,
                     (const uint32_t*)bstart, (const uint32_t*)vox_cnt, n_buckets, d_vox, d_vox_frame);
  prof_close(c, ST_VFINISH, pm, 16.0 * (double)N + (16.0 + 32.0) * n_vox);
  return HOPE_OK;
}

// A5.  Two ways to bring the points of a voxel together in ascending source index:
//  * depth frames: sort one record per (64-pixel chunk, voxel) group (k_bbox_groups has already
//    counted them, scan_groups placed them);
//  * point clouds (and depth frames under HOPE_DEBUG_PIXEL_SORT=1): sort one pair per point.
template <class Src, class VK>
int stage_voxel(hope_ctx* c, const Src& src, size_t n_per_frame, int cell_bits, int total_bits, float4*& d_vox,
                int*& d_vox_frame)
{
  const int F = c->n_frames;
  const size_t N = n_per_frame * (size_t)F;
  hipStream_t s = c->stream;
  PrimAlloc pa{arena_alloc_cb, &c->arena};
  const bool groups = use_groups(c, src);
  if (!groups && F == 1 && sizeof(VK) == 4 && c->voxel_part && !(c->h_glob->any_overflow & 1u) &&
      cell_bits > kPartLow && N > 0)
    return stage_voxel_part(c, src, N, cell_bits, d_vox, d_vox_frame);
  if (groups && c->dense)
    return stage_voxel_rank(c, src, n_per_frame, 0, cell_bits, nullptr, nullptr, d_vox, d_vox_frame);
  const size_t M = groups ? (size_t)c->h_glob->n_groups : N;  // records to sort
  if (M == 0) {
    c->n_vox_total = 0;
    return HOPE_OK;
  }
  VK *k_in, *k_out, *uniq;
  uint32_t *v_in, *v_out, *run_cnt, *run_start;
  uint4* rec = nullptr;
  ALLOC(k_in, VK, M);
  ALLOC(k_out, VK, M);
  ALLOC(v_in, uint32_t, M);
  ALLOC(v_out, uint32_t, M);
  size_t pm = prof_mark(c);
  if (groups && c->one_pass && !c->h_glob->code_overflow) {
    const size_t n_chunks = (size_t)c->cpf * F;
    rec = c->d_slots;
    hipLaunchKernelGGL((k_group_keys<VK>), dim3(nblk(n_chunks * 8)), dim3(kBlock), 0, s, c->d_fi, c->inv_xy, c->inv_z,
                       cell_bits, c->cpf, (unsigned)n_chunks, c->d_grp_off, c->d_slots, k_in, v_in);
    prof_close(c, ST_VKEYS, pm, 8.0 * ((double)c->cpf * F) + (4.0 + sizeof(VK) + 4.0) * M);
    // no sort: the voxels of a frame ranked by an LDS bitmap of its occupied cells (grids beyond 2^24 cells,
    // 16 cell ranges, and 64-bit keys take the sorting path)
    if (c->voxel_rank && sizeof(VK) == 4 && c->h_glob->max_cells <= (1ull << 24))
      return stage_voxel_rank(c, src, n_per_frame, M, cell_bits, (const uint32_t*)k_in, v_in, d_vox, d_vox_frame);
  } else if (groups) {
    // two passes: by choice, or because a cell fell outside the code range of k_bbox_records (its counts
    // and bounding box are valid all the same)
    ALLOC(rec, uint4, M);
    const unsigned bx = std::max(1u, std::min<unsigned>(nblk(c->cpf, (kBlock / kWave) * 16), 512u));
    launch_group_records<VK>(c, src, bx, cell_bits, k_in, v_in, rec);
    prof_close(c, ST_VKEYS, pm, 2.0 * N + 4.0 * ((double)c->cpf * F) + (sizeof(VK) + 4.0 + 16.0) * M);
  } else {
    launch_voxel_keys<VK>(c, src, n_per_frame, cell_bits, k_in, v_in);
    prof_close(c, ST_VKEYS, pm, (2.0 + sizeof(VK) + 4.0) * N);
  }
  pm = prof_mark(c);
  if (sizeof(VK) == 4 && groups && c->seg_sort && M / (size_t)F < ((size_t)1 << 18)) {
    // records are emitted frame by frame: sort the cell bits within each frame's range (many frames of
    // modest size; one workgroup sorts a segment, so huge frames go through the device-wide sort)
    uint32_t* seg_off;
    ALLOC(seg_off, uint32_t, (size_t)F + 1);
    hipLaunchKernelGGL(k_frame_record_ranges, dim3(nblk((size_t)F + 1)), dim3(kBlock), 0, s, c->d_grp_off, c->cpf, F,
                       seg_off);
    HIPCHK(segmented_sort_pairs_u32(pa, (const uint32_t*)k_in, (uint32_t*)k_out, v_in, v_out, M, (unsigned)F, seg_off,
                                    cell_bits, s));
  } else if (sizeof(VK) == 4) {
    HIPCHK(sort_pairs_u32(pa, (const uint32_t*)k_in, (uint32_t*)k_out, v_in, v_out, M, total_bits, s));
  } else {
    HIPCHK(sort_pairs_u64(pa, (const uint64_t*)k_in, (uint64_t*)k_out, v_in, v_out, M, total_bits, s));
  }
  prof_close(c, ST_VSORT, pm, 2.0 * (sizeof(VK) + 4.0) * M);
  pm = prof_mark(c);
  // k_in / v_in are dead now: reuse them for the run-length outputs
  uniq = k_in;
  run_cnt = v_in;
  ALLOC(run_start, uint32_t, M + 1);
  if (sizeof(VK) == 4) {
    HIPCHK(rle_u32(pa, (const uint32_t*)k_out, M, (uint32_t*)uniq, run_cnt, &c->d_glob->n_runs, s));
  } else {
    HIPCHK(rle_u64(pa, (const uint64_t*)k_out, M, (uint64_t*)uniq, run_cnt, &c->d_glob->n_runs, s));
  }
  hipLaunchKernelGGL((k_frame_vox_ranges<VK>), dim3(nblk(F + 1)), dim3(kBlock), 0, s, c->d_fi, F, uniq,
                     &c->d_glob->n_runs, cell_bits, c->d_glob);
  hipLaunchKernelGGL(k_frame_vox_counts, dim3(nblk(F)), dim3(kBlock), 0, s, c->d_fi, F, c->d_glob);
  prof_close(c, ST_VRLE, pm, (double)sizeof(VK) * M);
  int st = sync_globals(c);
  if (st != HOPE_OK) return st;
  const unsigned n_vox = c->h_glob->n_vox_total;
  c->n_vox_total = n_vox;
  if (n_vox == 0) return HOPE_OK;
  pm = prof_mark(c);
  HIPCHK(arena_scan_u32(c, run_cnt, run_start, n_vox, s));
  prof_close(c, ST_VRLE, pm, 8.0 * n_vox);
  ALLOC(d_vox, float4, n_vox);
  ALLOC(d_vox_frame, int, n_vox);
  pm = prof_mark(c);
  if (groups) {
    launch_centroids_groups<VK>(c, src, uniq, run_start, run_cnt, v_out, rec, n_vox, cell_bits, d_vox, d_vox_frame);
    prof_close(c, ST_CENTROID, pm, (4.0 + 16.0) * M + 2.0 * N + 28.0 * n_vox);
  } else {
    hipLaunchKernelGGL((k_centroids_w<Src, VK>), dim3(nblk(n_vox, kWave)), dim3(kWave), 0, s, src, c->d_fi, uniq,
                       run_start, run_cnt, v_out, n_vox, cell_bits, d_vox, d_vox_frame);
    prof_close(c, ST_CENTROID, pm, (4.0 + src_point_bytes(src)) * N + 28.0 * n_vox);
  }
  return HOPE_OK;
}

// A10..A12 of the clusters of big_from members and more (verify_big.hpp): exact_sum.hpp's integer blocks in
// place of the chain of dependent adds.
int verify_big(hope_ctx* c, unsigned n_cl, unsigned n_big, unsigned n_sb, int big_from, const uint32_t* members,
               const uint32_t* cl_first, const float4* fit_pt, const float4* fit_nrm, ClusterRec* rec)
{
  hipStream_t s = c->stream;
  const size_t n_el = (size_t)n_sb * xs::kSuper, n_blk = (size_t)n_sb * xs::kSub;
  unsigned* cursor;
  xs::Span* spans;
  BigInfo* info;
  unsigned *sb_owner, *part;
  BigState* state;
  float *terms, *spred, *sums;
  double* bsum;
  xs::RecA *recA, *topA;
  int *recB, *topB;
  ALLOC(cursor, unsigned, 2);
  ALLOC(spans, xs::Span, n_big);
  ALLOC(info, BigInfo, n_big);
  ALLOC(sb_owner, unsigned, n_sb);
  ALLOC(part, unsigned, 6 * (size_t)n_sb);
  ALLOC(state, BigState, n_big);
  ALLOC(terms, float, 6 * n_el);
  ALLOC(bsum, double, 6 * n_blk);
  ALLOC(spred, float, 6 * n_blk);
  ALLOC(recA, xs::RecA, 6 * n_blk);
  ALLOC(recB, int, 6 * n_blk);
  ALLOC(topA, xs::RecA, 6 * (size_t)n_sb);
  ALLOC(topB, int, 6 * (size_t)n_sb);
  ALLOC(sums, float, (size_t)n_big * kBigSums);
  if (!c->d_xs_stats) {
    if (hipMalloc((void**)&c->d_xs_stats, 24 * sizeof(xs::Stats)) != hipSuccess) {
      (void)hipGetLastError();
      c->d_xs_stats = nullptr;
    }
  }
  if (c->d_xs_stats) HIPCHK(hipMemsetAsync(c->d_xs_stats, 0, 24 * sizeof(xs::Stats), s));
  HIPCHK(hipMemsetAsync(cursor, 0, 2 * sizeof(unsigned), s));
  HIPCHK(hipMemsetAsync(sums, 0, sizeof(float) * (size_t)n_big * kBigSums, s));
  hipLaunchKernelGGL(k_big_list, dim3(nblk(n_cl)), dim3(kBlock), 0, s, rec, n_cl, big_from, cursor, spans, info, sb_owner,
                     state);
  auto sums_of = [&](int n_chains, int sum_off) {
    hipLaunchKernelGGL(xs::k_xs_scan, dim3(n_big, n_chains), dim3(256), 0, s, spans, bsum, n_blk, spred);
    hipLaunchKernelGGL(xs::k_xs_records, dim3(n_sb, n_chains), dim3(256), 0, s, terms, n_el, spred, n_blk, recA, recB,
                       topA, topB, (size_t)n_sb);
    hipLaunchKernelGGL(xs::k_xs_chain, dim3(n_big, n_chains), dim3(64), 0, s, spans, terms, n_el, recA, recB, n_blk, topA,
                       topB, (size_t)n_sb, sums + sum_off, kBigSums, &state[0].alive,
                       (int)(sizeof(BigState) / sizeof(int)), c->d_xs_stats ? c->d_xs_stats + sum_off : nullptr);
  };
  const dim3 gt(n_sb), gg(n_big);
  hipLaunchKernelGGL(k_big_terms<1>, gt, dim3(kBlock), 0, s, spans, info, sb_owner, state, members, cl_first, fit_pt,
                     fit_nrm, terms, n_el, bsum, n_blk, part);
  sums_of(4, 0);
  hipLaunchKernelGGL(k_big_gate<1>, gg, dim3(kWave), 0, s, spans, info, state, part, sums, c->th_angle, rec);
  hipLaunchKernelGGL(k_big_terms<2>, gt, dim3(kBlock), 0, s, spans, info, sb_owner, state, members, cl_first, fit_pt,
                     fit_nrm, terms, n_el, bsum, n_blk, part);
  sums_of(3, 4);
  hipLaunchKernelGGL(k_big_gate<2>, gg, dim3(kWave), 0, s, spans, info, state, part, sums, c->th_angle, rec);
  hipLaunchKernelGGL(k_big_terms<3>, gt, dim3(kBlock), 0, s, spans, info, sb_owner, state, members, cl_first, fit_pt,
                     fit_nrm, terms, n_el, bsum, n_blk, part);
  sums_of(6, 8);
  hipLaunchKernelGGL(k_big_gate<3>, gg, dim3(kWave), 0, s, spans, info, state, part, sums, c->th_angle, rec);
  return HOPE_OK;
}

template <class CK>
int stage_rest(hope_ctx* c, float4* d_vox, int* d_vox_frame, int zbits, int key2_bits)
{
  const int F = c->n_frames;
  const unsigned n_vox = c->n_vox_total;
  const int K = c->cfg.knn;
  hipStream_t s = c->stream;
  PrimAlloc pa{arena_alloc_cb, &c->arena};
  const unsigned total_cols = c->h_glob->total_cols;

  // ---- column index ---------------------------------------------------------------
  CK *ck_in, *ck_out;
  uint32_t *cv_in, *cv_out, *col_first;
  float4* spt;
  ALLOC(ck_in, CK, n_vox);
  ALLOC(ck_out, CK, n_vox);
  ALLOC(cv_in, uint32_t, n_vox);
  ALLOC(cv_out, uint32_t, n_vox);
  ALLOC(col_first, uint32_t, (size_t)total_cols + 1);
  ALLOC(spt, float4, n_vox);
  // VoxelGrid output arrives in voxel order (z cell, then y, then x): a STABLE sort by column alone leaves
  // every column ascending in z cell, which is all the searches need (one radix pass fewer).  A cloud
  // adopted as it is, or passed through, comes in input order and keeps the z cell in its key.
  const bool col_only = !c->col_sort_z && !c->input_is_downsampled && !(c->h_glob->any_overflow & 1u);
  const int czbits = col_only ? 0 : zbits;
  const int ckey_bits = key2_bits - zbits + czbits;
  size_t pm = prof_mark(c);
  // one cloud in VoxelGrid order: A6 over the 3-D occupancy index (grid_index.hpp), no column index of all points
  const bool use_grid = c->grid_index && !c->grid_off_once && F == 1 && c->normals_direct && !c->inj_normals &&
                        !(c->h_glob->any_overflow & 1u) && c->h_glob->max_cells <= kGridMaxCells &&
                        c->ring <= kGridMaxRing && c->ablate == 0 && n_vox > 0;
  uint2* grid_tab = nullptr;
  if (use_grid) {
    const unsigned n_words = (unsigned)(c->h_glob->max_cells / 32ull) + 2u;
    uint32_t *gcnt, *gsums;
    ALLOC(grid_tab, uint2, n_words);
    ALLOC(gcnt, uint32_t, (size_t)n_words + 4);
    ALLOC(gsums, uint32_t, scan_sums_words(n_words));
    HIPCHK(hipMemsetAsync(grid_tab, 0, sizeof(uint2) * (size_t)n_words, s));
    HIPCHK(hipMemsetAsync(&c->d_glob->grid_unsorted, 0, sizeof(unsigned), s));
    hipLaunchKernelGGL(k_grid_mark, dim3(nblk(n_vox)), dim3(kBlock), 0, s, (const float4*)d_vox, n_vox, c->d_fi, c->inv_xy,
                       c->inv_z, grid_tab, c->d_glob);
    hipLaunchKernelGGL(k_grid_popc, dim3(nblk(n_words)), dim3(kBlock), 0, s, (const uint2*)grid_tab, n_words, gcnt);
    scan_exclusive_u32(gcnt, gcnt, n_words, gsums, s);
    hipLaunchKernelGGL(k_grid_pref, dim3(nblk(n_words)), dim3(kBlock), 0, s, grid_tab, n_words, (const uint32_t*)gcnt);
    prof_close(c, ST_COLTABLE, pm, 16.0 * n_vox + 8.0 * n_words);
  } else if (col_only && c->col_index_lds && c->h_glob->max_cols <= kCntCap && (size_t)n_vox <= (size_t)200000 * F) {
    const size_t ci_lds = sizeof(unsigned) * (size_t)c->h_glob->max_cols;
    if (ci_lds > ((size_t)48 << 10))
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_column_index), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)ci_lds));
    hipLaunchKernelGGL(k_column_index, dim3(F), dim3(kRankBlock), ci_lds, s, d_vox, c->d_fi, F, c->inv_xy, n_vox, total_cols,
                       col_first, cv_in, spt, (const uint32_t*)c->d_frame_order);
    prof_close(c, ST_COLTABLE, pm, (16.0 + 16.0) * n_vox + 4.0 * total_cols);
  } else {
  hipLaunchKernelGGL((k_column_keys<CK>), dim3(nblk(n_vox)), dim3(kBlock), 0, s, d_vox, d_vox_frame, c->d_fi,
                     n_vox, c->inv_xy, c->inv_z, czbits, ck_in, cv_in);
  prof_close(c, ST_COLKEYS, pm, (16.0 + 4.0 + sizeof(CK) + 4.0) * n_vox);
  pm = prof_mark(c);
  if (sizeof(CK) == 4) {
    HIPCHK(sort_pairs_u32(pa, (const uint32_t*)ck_in, (uint32_t*)ck_out, cv_in, cv_out, n_vox, ckey_bits, s));
  } else {
    HIPCHK(sort_pairs_u64(pa, (const uint64_t*)ck_in, (uint64_t*)ck_out, cv_in, cv_out, n_vox, ckey_bits, s));
  }
  prof_close(c, ST_COLSORT, pm, 2.0 * (sizeof(CK) + 4.0) * n_vox);
  pm = prof_mark(c);
  hipLaunchKernelGGL((k_column_table<CK>), dim3(nblk((size_t)n_vox + 1)), dim3(kBlock), 0, s, ck_out, n_vox,
                     total_cols, czbits, col_first);
  hipLaunchKernelGGL(k_gather_sorted, dim3(nblk(n_vox)), dim3(kBlock), 0, s, d_vox, cv_out, n_vox, spt);
  prof_close(c, ST_COLTABLE, pm, (4.0 + 16.0 + 16.0) * n_vox + 4.0 * total_cols);
  }

  // ---- A6 normals -----------------------------------------------------------------
  float4* d_normals;
  ALLOC(d_normals, float4, n_vox);
  c->d_normals = d_normals;
  pm = prof_mark(c);
  // 32x32-column super tiles when a super tile + halo usually fits in LDS (the bench's depth frames: ~700 points per
  // super tile, normals 2.06 -> 1.96 ms, k-NN 1.87 -> 1.76 ms), 16x16 tiles for dense clouds, where a super tile
  // would only be four tiles run one after the other by a quarter of the workgroups (config 4, 2800 points per
  // super tile: normals 12.2 instead of 6.3 ms)
  const unsigned sum_tiles = std::max(1u, c->h_glob->sum_tiles);
  // -- and only when there are enough super tiles to fill the chip (a single 640x480 frame has ~20: 1.08 -> 1.44 ms)
  const auto tile_side = [&](size_t points) {
    return sum_tiles >= 1024u && points <= (size_t)kTileCap * 2 / 3 * sum_tiles ? kSuper : kTile;
  };
  const auto tile_grid_of = [&](int side) {
    return dim3(std::max(1u, side == kSuper ? c->h_glob->max_tiles : c->h_glob->max_tiles16), F);
  };
  const int side_n = c->tile_side ? c->tile_side : tile_side(n_vox);
  const dim3 tile_grid = tile_grid_of(side_n);
  // the flags of the A7 gate, in index order and in the order of the column index: written by k_normals_direct
  // itself when it computes every normal (gate_fused), else by k_fit_flags / k_fit_sorted_flags below
  uint32_t *flags, *sflags;
  ALLOC(flags, uint32_t, (size_t)n_vox + 1);
  ALLOC(sflags, uint32_t, (size_t)n_vox + 1);
  bool gate_fused = false;
  if (c->inj_normals) {
    HIPCHK(hipMemcpyAsync(d_normals, c->inj_normals, sizeof(float4) * (size_t)n_vox,
                          c->inj_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
  } else if (use_grid) {
    gate_fused = c->own_lo == 0u && c->own_hi == 0xffffffffu;
    uint32_t* ff = gate_fused ? flags : nullptr;
    if (c->normals_cap_small)
      hipLaunchKernelGGL((k_normals_grid<2>), dim3(nblk(n_vox)), dim3(kBlock), 0, s, (const float4*)d_vox,
                         (const uint2*)grid_tab, c->d_fi, c->inv_xy, c->inv_z, c->radius, c->r2, c->ring, c->own_lo,
                         c->own_hi, (unsigned)n_vox, d_normals, c->th_norm, ff);
    else
      hipLaunchKernelGGL((k_normals_grid<kNbCap>), dim3(nblk(n_vox)), dim3(kBlock), 0, s, (const float4*)d_vox,
                         (const uint2*)grid_tab, c->d_fi, c->inv_xy, c->inv_z, c->radius, c->r2, c->ring, c->own_lo,
                         c->own_hi, (unsigned)n_vox, d_normals, c->th_norm, ff);
  } else if (c->normals_direct) {
    const int* vf = F == 1 ? nullptr : d_vox_frame;
    gate_fused = c->own_lo == 0u && c->own_hi == 0xffffffffu && c->ablate == 0;
    uint32_t* ff = gate_fused ? flags : nullptr;
    if (c->normals_cap_small)
      hipLaunchKernelGGL((k_normals_direct<2>), dim3((nblk(n_vox) + 7u) / 8u * 8u), dim3(kBlock), 0, s, spt, col_first, c->d_fi, c->inv_xy,
                         c->inv_z, c->radius, c->r2, c->ring, c->own_lo, c->own_hi, (unsigned)n_vox, vf, d_normals,
                         c->ablate % 10, c->th_norm, ff, sflags);
    else
      hipLaunchKernelGGL((k_normals_direct<kNbCap>), dim3((nblk(n_vox) + 7u) / 8u * 8u), dim3(kBlock), 0, s, spt, col_first, c->d_fi,
                         c->inv_xy, c->inv_z, c->radius, c->r2, c->ring, c->own_lo, c->own_hi, (unsigned)n_vox, vf,
                         d_normals, c->ablate % 10, c->th_norm, ff, sflags);
  } else {
#define HOPE_NORMALS(CAP, SIDE)                                                                                       \
  hipLaunchKernelGGL((k_normals_tile<CAP, SIDE>), tile_grid, dim3(kBlock), 0, s, spt, col_first, c->d_fi, c->inv_xy, \
                     c->inv_z, c->radius, c->r2, c->ring, c->own_lo, c->own_hi, d_normals, c->ablate % 10)
    if (c->normals_cap_small) {
      if (side_n == kSuper) HOPE_NORMALS(2, kSuper); else HOPE_NORMALS(2, kTile);
    } else {
      if (side_n == kSuper) HOPE_NORMALS(kNbCap, kSuper); else HOPE_NORMALS(kNbCap, kTile);
    }
#undef HOPE_NORMALS
  }
  prof_close(c, ST_NORMALS, pm, (16.0 + 16.0) * n_vox);
  // the derived cells did not ascend (a centroid rounded across a cell face; an adopted cloud that is no VoxelGrid
  // output): the occupancy index does not describe this cloud -- the same call again over the column index
  const auto grid_retry = [&]() {
    c->grid_off_once = 1;
    const int st = stage_rest<CK>(c, d_vox, d_vox_frame, zbits, key2_bits);
    c->grid_off_once = 0;
    return st;
  };
  if (c->stop_stage == 1) {
    if (use_grid) {
      int st = sync_globals(c);
      if (st != HOPE_OK) return st;
      if (c->h_glob->grid_unsorted) return grid_retry();
    }
    return HOPE_OK;
  }
  // ---- A7 gate + compaction ---------------------------------------------------------
  uint32_t* fscan;
  ALLOC(fscan, uint32_t, (size_t)n_vox + 1);
  pm = prof_mark(c);
  if (!gate_fused)
    hipLaunchKernelGGL(k_fit_flags, dim3(nblk((size_t)n_vox + 1)), dim3(kBlock), 0, s, d_normals, n_vox, c->th_norm,
                       flags);
  HIPCHK(arena_scan_u32(c, flags, fscan, (size_t)n_vox + 1, s));
  hipLaunchKernelGGL(k_frame_fit_ranges, dim3(nblk(F)), dim3(kBlock), 0, s, c->d_fi, F, fscan, c->d_glob);
  HIPCHK(hipMemcpyAsync(&c->d_glob->pad, fscan + n_vox, sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
  prof_close(c, ST_FIT, pm, (16.0 + 4.0 + 4.0 + 4.0) * n_vox);
  {
    int st = sync_globals(c);  // n_fit (pad) and the largest per-frame count (max_fit)
    if (st != HOPE_OK) return st;
  }
  if (use_grid && c->h_glob->grid_unsorted) return grid_retry();
  uint32_t* h_cnt = reinterpret_cast<uint32_t*>(&c->h_glob->pad);
  const unsigned n_fit = *h_cnt;
  const unsigned max_fit = c->h_glob->max_fit;
  c->n_fit_total = n_fit;
  int* fit_rank;
  ALLOC(fit_rank, int, n_vox);
  uint32_t* fit_idx;
  float4 *fit_pt, *fit_nrm;
  int* fit_frame;
  ALLOC(fit_idx, uint32_t, std::max(n_fit, 1u));
  ALLOC(fit_pt, float4, std::max(n_fit, 1u));
  ALLOC(fit_nrm, float4, std::max(n_fit, 1u));
  ALLOC(fit_frame, int, std::max(n_fit, 1u));
  c->d_fit_idx = fit_idx;
  c->d_fit_pt = fit_pt;
  pm = prof_mark(c);
  hipLaunchKernelGGL(k_fit_scatter, dim3(nblk(n_vox)), dim3(kBlock), 0, s, flags, fscan, n_vox, d_vox, d_normals,
                     d_vox_frame, fit_rank, fit_idx, fit_pt, fit_nrm, fit_frame);
  prof_close(c, ST_FIT, pm, (4.0 + 4.0 + 4.0) * n_vox + (2.0 * (16.0 + 16.0) + 4.0 + 4.0 + 4.0) * n_fit);
  if (n_fit == 0) return HOPE_OK;

  HIPCHK(hipEventRecord(c->ev_a, s));  // the reference's timer window opens (plane_segment.cpp:157)

  // ---- A8 kNN -----------------------------------------------------------------------
  int* knn;
  ALLOC(knn, int, (size_t)n_fit * K);
  c->d_knn = knn;
  uint32_t* todo;
  ALLOC(todo, uint32_t, n_fit);
  // the column index of the fit points alone: the k-NN tiles then hold and scan only points that can
  // be neighbours
  uint32_t *fpos, *col_first_fit;
  float4* spt_fit;
  ALLOC(fpos, uint32_t, (size_t)n_vox + 1);
  ALLOC(col_first_fit, uint32_t, (size_t)total_cols + 1);
  ALLOC(spt_fit, float4, n_fit);
  pm = prof_mark(c);
  if (use_grid) {
    // the fit points are in VoxelGrid order (z cell, then y, then x): a stable sort by column alone leaves every
    // column ascending in z cell; 12x fewer points than the whole down-sampled cloud on the 1e8-point scene
    hipLaunchKernelGGL((k_column_keys<CK>), dim3(nblk(n_fit)), dim3(kBlock), 0, s, (const float4*)fit_pt,
                       (const int*)fit_frame, c->d_fi, n_fit, c->inv_xy, c->inv_z, 0, ck_in, cv_in);
    if (sizeof(CK) == 4) {
      HIPCHK(sort_pairs_u32(pa, (const uint32_t*)ck_in, (uint32_t*)ck_out, cv_in, cv_out, n_fit, key2_bits - zbits, s));
    } else {
      HIPCHK(sort_pairs_u64(pa, (const uint64_t*)ck_in, (uint64_t*)ck_out, cv_in, cv_out, n_fit, key2_bits - zbits, s));
    }
    hipLaunchKernelGGL((k_column_table<CK>), dim3(nblk((size_t)n_fit + 1)), dim3(kBlock), 0, s, ck_out, n_fit, total_cols,
                       0, col_first_fit);
    hipLaunchKernelGGL(k_gather_coarse, dim3(nblk(n_fit)), dim3(kBlock), 0, s, (const float4*)fit_pt, cv_out, n_fit,
                       spt_fit);
  } else {
  if (!gate_fused)
    hipLaunchKernelGGL(k_fit_sorted_flags, dim3(nblk((size_t)n_vox + 1)), dim3(kBlock), 0, s, spt, fit_rank, n_vox,
                       sflags);
  HIPCHK(arena_scan_u32(c, sflags, fpos, (size_t)n_vox + 1, s));
  hipLaunchKernelGGL(k_fit_sorted_scatter, dim3(nblk(n_vox)), dim3(kBlock), 0, s, spt, fit_rank, sflags, fpos, n_vox,
                     spt_fit);
  hipLaunchKernelGGL(k_fit_col_table, dim3(nblk((size_t)total_cols + 1)), dim3(kBlock), 0, s, col_first, fpos,
                     total_cols, col_first_fit);
  }
  prof_close(c, ST_FITIDX, pm, (16.0 + 4.0 + 4.0 + 8.0 + 8.0) * n_vox + 32.0 * (double)n_fit + 8.0 * total_cols);
  pm = prof_mark(c);
  HIPCHK(hipMemsetAsync(&c->d_glob->n_todo, 0, sizeof(unsigned), s));
  if (c->inj_knn) {
    HIPCHK(hipMemcpyAsync(knn, c->inj_knn, sizeof(int) * (size_t)n_fit * K,
                          c->inj_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
  } else if (c->knn_direct) {
    const int* ff = F == 1 ? nullptr : fit_frame;
    if (K <= 8)
      hipLaunchKernelGGL((k_knn_direct<8>), dim3((nblk(n_fit) + 7u) / 8u * 8u), dim3(kBlock), 0, s, spt_fit, col_first_fit, c->d_fi, ff,
                         c->inv_xy, c->inv_z, c->cfg.th_xy, K, fit_idx, c->own_lo, c->own_hi, n_fit, knn, todo,
                         &c->d_glob->n_todo);
    else
      hipLaunchKernelGGL((k_knn_direct<kMaxKnn>), dim3((nblk(n_fit) + 7u) / 8u * 8u), dim3(kBlock), 0, s, spt_fit, col_first_fit, c->d_fi,
                         ff, c->inv_xy, c->inv_z, c->cfg.th_xy, K, fit_idx, c->own_lo, c->own_hi, n_fit, knn, todo,
                         &c->d_glob->n_todo);
  } else {
    const int side_k = c->tile_side ? c->tile_side : tile_side(n_fit);
#define HOPE_KNN(KL, SIDE)                                                                                              \
  hipLaunchKernelGGL((k_knn_tile<KL, SIDE>), tile_grid_of(SIDE), dim3(kBlock), 0, s, spt_fit, col_first_fit, nullptr,   \
                     c->d_fi, c->inv_xy, c->inv_z, c->cfg.th_xy, K, fit_idx, c->own_lo, c->own_hi, knn, todo,          \
                     &c->d_glob->n_todo, (c->ablate / 10) % 10)
    if (K <= 8) {
      if (side_k == kSuper) HOPE_KNN(8, kSuper); else HOPE_KNN(8, kTile);
    } else {
      if (side_k == kSuper) HOPE_KNN(kMaxKnn, kSuper); else HOPE_KNN(kMaxKnn, kTile);
    }
#undef HOPE_KNN
  }
  prof_close(c, ST_KNN, pm, (16.0 + 4.0 * K) * (double)n_fit);
  {
    int st = sync_globals(c);
    if (st != HOPE_OK) return st;
  }
  const unsigned n_todo = c->h_glob->n_todo;
  c->n_knn_todo = n_todo;
  if (n_todo > 0) {
    // queries whose neighbours lie outside their LDS tile: a whole-frame ring expansion, over the
    // fine index when they are few, over a coarse index of the fit points when they are many
    pm = prof_mark(c);
    const unsigned gt = nblk(n_todo, kBlock / kWave);  // one wave per query
    if (n_todo <= std::max(c->knn_fine_limit, c->knn_fine_limit ? n_fit / 16 : 0u)) {
      if (K <= 8)
        hipLaunchKernelGGL((k_knn_todo<CoarseCols, 8>), dim3(gt), dim3(kBlock), 0, s, todo, &c->d_glob->n_todo, fit_pt,
                           fit_frame, spt_fit, col_first_fit, nullptr, c->d_fi, c->inv_xy, c->inv_z, c->cfg.th_xy, 1, K,
                           knn);
      else
        hipLaunchKernelGGL((k_knn_todo<CoarseCols, kMaxKnn>), dim3(gt), dim3(kBlock), 0, s, todo, &c->d_glob->n_todo,
                           fit_pt, fit_frame, spt_fit, col_first_fit, nullptr, c->d_fi, c->inv_xy, c->inv_z,
                           c->cfg.th_xy, 1, K, knn);
    } else {
      const unsigned total_ccols = c->h_glob->total_ccols;
      CK *qk_in, *qk_out;
      uint32_t *qv_in, *qv_out, *ccol_first;
      float4* cspt;
      ALLOC(qk_in, CK, n_fit);
      ALLOC(qk_out, CK, n_fit);
      ALLOC(qv_in, uint32_t, n_fit);
      ALLOC(qv_out, uint32_t, n_fit);
      ALLOC(ccol_first, uint32_t, (size_t)total_ccols + 1);
      ALLOC(cspt, float4, n_fit);
      hipLaunchKernelGGL((k_coarse_keys<CK>), dim3(nblk(n_fit)), dim3(kBlock), 0, s, fit_pt, fit_frame, c->d_fi, n_fit,
                         c->inv_xy, c->inv_z, zbits, qk_in, qv_in);
      const int cbits = zbits + bits_for(total_ccols);
      if (sizeof(CK) == 4) {
        HIPCHK(sort_pairs_u32(pa, (const uint32_t*)qk_in, (uint32_t*)qk_out, qv_in, qv_out, n_fit, cbits, s));
      } else {
        HIPCHK(sort_pairs_u64(pa, (const uint64_t*)qk_in, (uint64_t*)qk_out, qv_in, qv_out, n_fit, cbits, s));
      }
      hipLaunchKernelGGL((k_column_table<CK>), dim3(nblk((size_t)n_fit + 1)), dim3(kBlock), 0, s, qk_out, n_fit,
                         total_ccols, zbits, ccol_first);
      hipLaunchKernelGGL(k_gather_coarse, dim3(nblk(n_fit)), dim3(kBlock), 0, s, fit_pt, qv_out, n_fit, cspt);
      if (K <= 8 && c->todo_threads)  // many queries: a thread each
        hipLaunchKernelGGL((k_knn_todo_threads<8>), dim3((nblk(n_todo) + 7u) / 8u * 8u), dim3(kBlock), 0, s, todo, &c->d_glob->n_todo,
                           fit_pt, fit_frame, cspt, ccol_first, c->d_fi, c->inv_xy, c->inv_z, c->cfg.th_xy, kCoarse, K,
                           knn);
      else if (K <= 8)
        hipLaunchKernelGGL((k_knn_todo<CoarseCols, 8>), dim3(gt), dim3(kBlock), 0, s, todo, &c->d_glob->n_todo, fit_pt,
                           fit_frame, cspt, ccol_first, nullptr, c->d_fi, c->inv_xy, c->inv_z, c->cfg.th_xy, kCoarse, K,
                           knn);
      else
        hipLaunchKernelGGL((k_knn_todo<CoarseCols, kMaxKnn>), dim3(gt), dim3(kBlock), 0, s, todo, &c->d_glob->n_todo,
                           fit_pt, fit_frame, cspt, ccol_first, nullptr, c->d_fi, c->inv_xy, c->inv_z, c->cfg.th_xy,
                           kCoarse, K, knn);
    }
    prof_close(c, ST_KNN_TODO, pm, (16.0 + 4.0 * K) * (double)n_todo);
  }
  if (c->stop_stage == 2) return HOPE_OK;
  // ---- A9 region growing as min-rank reachability -----------------------------------
  int *rank, *seed_node, *parent, *L, *root, *label;
  ALLOC(rank, int, n_fit);
  ALLOC(seed_node, int, n_fit);
  ALLOC(parent, int, n_fit);
  ALLOC(L, int, n_fit);
  ALLOC(root, int, n_fit);
  ALLOC(label, int, n_fit);
  const unsigned gb = nblk(n_fit);
  const unsigned gb8 = (gb + 7u) / 8u * 8u;  // (kernels that map their blocks with xcd_block())
  pm = prof_mark(c);
  hipLaunchKernelGGL(k_seed_ranks, dim3(gb), dim3(kBlock), 0, s, c->d_fi, fit_frame, n_fit, rank, seed_node, parent,
                     L);
  uint32_t* cross;
  uint2* edges;
  ALLOC(cross, uint32_t, n_fit);
  ALLOC(edges, uint2, (size_t)n_fit * K);
  unsigned* d_n_edges = &c->d_glob->n_runs;  // free again after the voxel stage
  // tiles of a frame: as large as LDS allows, except that a single cloud (38 tiles of 24 576 points for the 10^8-point
  // scene: 38 of 256 CUs at work) is cut so that there are two tiles per CU
  unsigned uf_tile = (unsigned)kUfTile;
  if (F == 1 && c->union_small_tiles && max_fit > 4u * (unsigned)kUfTile)
    uf_tile = c->union_small_tiles >= kUfBlock
                  ? std::min<unsigned>(kUfTile, (unsigned)c->union_small_tiles / kUfBlock * kUfBlock)
                  : std::min<unsigned>(kUfTile, std::max<unsigned>(4096u, (max_fit / 512u + kUfBlock - 1u) / kUfBlock * kUfBlock));
  const dim3 uf_grid(std::max(1u, nblk(max_fit, (int)uf_tile)), F);
  const size_t uf_lds = sizeof(int) * std::min<size_t>(uf_tile, std::max<size_t>(max_fit, 1));  // the largest tile
  HIPCHK(hipMemsetAsync(d_n_edges, 0, sizeof(unsigned), s));
  if (uf_lds > ((size_t)48 << 10))
    HIPCHK(hipFuncSetAttribute(K == 8 ? reinterpret_cast<const void*>(&k_union_local<true>)
                                      : reinterpret_cast<const void*>(&k_union_local<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)uf_lds));
  if (K == 8)
    hipLaunchKernelGGL(k_union_local<true>, uf_grid, dim3(kUfBlock), uf_lds, s, knn, fit_pt, c->d_fi, K, c->cfg.th_z,
                       parent, cross, edges, d_n_edges, (const uint32_t*)nullptr, uf_tile);  // (frame order: 0.76 -> 0.83 ms here)
  else
    hipLaunchKernelGGL(k_union_local<false>, uf_grid, dim3(kUfBlock), uf_lds, s, knn, fit_pt, c->d_fi, K, c->cfg.th_z,
                       parent, cross, edges, d_n_edges, (const uint32_t*)nullptr, uf_tile);  // (frame order: 0.76 -> 0.83 ms here)
  hipLaunchKernelGGL(k_union_cross, dim3(gb8), dim3(kBlock), 0, s, knn, cross, n_fit, K, parent, (c->ablate / 100) % 10);
  hipLaunchKernelGGL(k_flatten_roots, dim3((nblk(n_fit, kBlock * kSizeRun) + 7u) / 8u * 8u), dim3(kBlock), 0, s, parent, rank,
                     n_fit, root, L);
  if (K == 8)
    hipLaunchKernelGGL(k_cross_edges<true>, dim3(gb8), dim3(kBlock), 0, s, knn, fit_pt, root, fit_frame, c->d_fi, n_fit, K,
                       c->cfg.th_z, edges, d_n_edges, uf_tile);
  else
    hipLaunchKernelGGL(k_cross_edges<false>, dim3(gb8), dim3(kBlock), 0, s, knn, fit_pt, root, fit_frame, c->d_fi, n_fit,
                       K, c->cfg.th_z, edges, d_n_edges, uf_tile);
  prof_close(c, ST_LABELS, pm, (2.0 * 4.0 * K + 2.0 * 16.0 + 6 * 4.0) * n_fit);
  const unsigned ge = std::min(gb, 2048u);
  // Sweeps until a whole (shortcut, sweep) pair changes nothing.  Only the last pair of a batch is
  // watched, so the usual case (a fixed point within three pairs) costs one read-back, not one per pair.
  bool converged = false;
  for (int iter = 0; iter < 100000; ++iter) {
    pm = prof_mark(c);
    const int pairs = iter == 0 ? 4 : 2;
    for (int rep = 0; rep < pairs; ++rep) {
      if (rep == pairs - 1) HIPCHK(hipMemsetAsync(&c->d_glob->changed, 0, sizeof(unsigned), s));
      hipLaunchKernelGGL(k_label_shortcut, dim3(gb), dim3(kBlock), 0, s, root, seed_node, n_fit, L,
                         &c->d_glob->changed);
      hipLaunchKernelGGL(k_label_sweep_edges, dim3(ge), dim3(kBlock), 0, s, edges, d_n_edges, L,
                         &c->d_glob->changed);
    }
    prof_close(c, ST_LABELS, pm, 2.0 * 4.0 * n_fit);
    int st = sync_globals(c);
    if (st != HOPE_OK) return st;
    if (!c->h_glob->changed) {
      converged = true;
      break;
    }
  }
  if (!converged) {
    c->err = "label sweeps did not reach a fixed point";
    return HOPE_E_RANGE;
  }

  // ---- segments -> clusters (size filter, z_growing.cpp:149-162) ---------------------
  uint32_t* cnt;
  U3 *seg_flags, *seg_scan;  // (is a segment, is a cluster, points of the cluster) per seed rank and their scans
  ALLOC(cnt, uint32_t, (size_t)n_fit + 1);
  ALLOC(seg_flags, U3, (size_t)n_fit + 1);
  ALLOC(seg_scan, U3, (size_t)n_fit + 1);
  pm = prof_mark(c);
  HIPCHK(hipMemsetAsync(cnt, 0, sizeof(uint32_t) * ((size_t)n_fit + 1), s));
  hipLaunchKernelGGL(k_segment_sizes, dim3((nblk(n_fit, kBlock * kSizeRun) + 7u) / 8u * 8u), dim3(kBlock), 0, s, root, L,
                     n_fit, label, cnt);
  hipLaunchKernelGGL(k_segment_flags, dim3(nblk((size_t)n_fit + 1)), dim3(kBlock), 0, s, cnt, n_fit,
                     c->cfg.min_cluster, seg_flags);
  HIPCHK(exclusive_scan_u32x3(pa, seg_flags, seg_scan, (size_t)n_fit + 1, s));
  hipLaunchKernelGGL(k_frame_cluster_ranges, dim3(nblk(F)), dim3(kBlock), 0, s, c->d_fi, F, seg_scan, c->d_glob);
  prof_close(c, ST_SEGMENTS, pm, (4.0 + 4.0 + 4.0 + 7 * 4.0) * n_fit);
  // the cluster count comes back together with the number of large clusters and their super blocks
  const int big_from = c->verify_big_from > 0 ? std::max(c->verify_big_from, 3) : INT_MAX;
  HIPCHK(hipMemsetAsync(&c->d_glob->n_big, 0, 2 * sizeof(unsigned), s));
  if (max_fit >= (unsigned)big_from)
    hipLaunchKernelGGL(k_big_count, dim3(nblk((size_t)n_fit + 1)), dim3(kBlock), 0, s, cnt, seg_flags, n_fit + 1,
                       big_from, c->d_glob);
  HIPCHK(hipMemcpyAsync(&c->d_glob->pad, &seg_scan[n_fit].b, sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
  {
    int st = sync_globals(c);
    if (st != HOPE_OK) return st;
  }
  const unsigned n_cl = *h_cnt;
  const unsigned n_big = c->h_glob->n_big, n_big_sb = c->h_glob->n_big_sb;
  c->n_cl_total = n_cl;

  int* cid;
  uint32_t *mkey, *mkey_s, *mval, *members, *cl_first;
  ClusterRec* rec;
  ALLOC(cid, int, n_fit);
  ALLOC(mkey, uint32_t, n_fit);
  ALLOC(mkey_s, uint32_t, n_fit);
  ALLOC(mval, uint32_t, n_fit);
  ALLOC(members, uint32_t, n_fit);
  ALLOC(cl_first, uint32_t, std::max(n_cl, 1u));
  ALLOC(rec, ClusterRec, std::max(n_cl, 1u));
  c->d_cid = cid;
  c->d_members = members;
  c->d_cl_first = cl_first;
  pm = prof_mark(c);
  // members by cluster, ascending index within each: a batch of frames walks its frames (one workgroup each: no
  // sort); a single cloud sorts (cluster, index) pairs
  const unsigned mfc = c->h_glob->max_frame_clusters;
  const bool by_frame = c->members_by_frame && (F >= 64 || c->members_by_frame == 2) &&
                        mfc <= (unsigned)kMemberClusters && (max_fit <= 65536u || c->members_by_frame == 2);
  hipLaunchKernelGGL(k_point_clusters, dim3(gb), dim3(kBlock), 0, s, label, seg_flags, seg_scan, n_fit, n_cl, cid,
                     by_frame ? (uint32_t*)nullptr : mkey, mval);
  if (n_cl > 0) {
    if (!by_frame) HIPCHK(sort_pairs_u32(pa, mkey, mkey_s, mval, members, n_fit, bits_for(n_cl), s));
    hipLaunchKernelGGL(k_cluster_heads, dim3(gb), dim3(kBlock), 0, s, cnt, seg_flags, seg_scan, seed_node,
                       fit_frame, c->d_fi, n_fit, cl_first, rec);
    if (by_frame)
      hipLaunchKernelGGL(k_members_by_frame, dim3(F), dim3(kBlock), 0, s, c->d_fi, (const int*)cid,
                         (const uint32_t*)cl_first, bits_for(mfc > 0u ? mfc - 1u : 0u), members,
                         (const uint32_t*)c->d_frame_order);
    prof_close(c, ST_SEGMENTS, pm, (4.0 + 4.0 + 2 * 8.0 + 2 * 8.0) * n_fit);
    pm = prof_mark(c);
    // ---- A10..A12 -------------------------------------------------------------------
    // a cluster holds points of one frame, so max_fit bounds its size: depth batches never take the second launch.
    // A floor or a wall gets longer tiles (more LDS per workgroup, so for these alone); running the two launches
    // side by side on two streams was measured and gains nothing over one after the other
    uint32_t* v_order;
    unsigned* v_ends;
    ALLOC(v_order, uint32_t, n_cl);
    ALLOC(v_ends, unsigned, 2);
    HIPCHK(hipMemsetAsync(v_ends, 0, 2 * sizeof(unsigned), s));
    hipLaunchKernelGGL(k_cluster_order, dim3(nblk(n_cl)), dim3(kBlock), 0, s, (const ClusterRec*)rec, n_cl, v_ends,
                       v_order);
    hipLaunchKernelGGL(k_cluster_verify_b<kVTile>, dim3(n_cl), dim3(kBlock), 0, s, members, cl_first, fit_pt, fit_nrm,
                       n_cl, c->th_angle, 0, big_from, rec, (const uint32_t*)v_order);
    if (n_big > 0) {
      int st = verify_big(c, n_cl, n_big, n_big_sb, big_from, members, cl_first, fit_pt, fit_nrm, rec);
      if (st != HOPE_OK) return st;
    }
    prof_close(c, ST_VERIFY, pm, (4.0 + 16.0 + 16.0) * n_fit + 32.0 * n_cl);
    c->h_rec.resize(n_cl);
    HIPCHK(hipMemcpyAsync(c->h_rec.data(), rec, sizeof(ClusterRec) * n_cl, hipMemcpyDeviceToHost, s));
  }
  else {
    prof_close(c, ST_SEGMENTS, pm);
  }
  HIPCHK(hipEventRecord(c->ev_b, s));  // the window closes (plane_segment.cpp:168)
  return HOPE_OK;
}

inline int adopt_downsampled(hope_ctx* c, const CloudSrc& src, size_t n, float4*& d_vox, int*& d_vox_frame)
{
  ALLOC(d_vox, float4, n);
  ALLOC(d_vox_frame, int, n);
  hipLaunchKernelGGL((k_adopt_downsampled<CloudSrc>), dim3(nblk(n)), dim3(kBlock), 0, c->stream, src, c->d_fi, d_vox,
                     d_vox_frame);
  c->n_vox_total = (unsigned)n;
  return HOPE_OK;
}
inline int adopt_downsampled(hope_ctx* c, const DepthSrc&, size_t, float4*&, int*&)
{
  c->err = "a depth frame cannot be an already down-sampled cloud";
  return HOPE_E_INVALID;
}

// Decides once per camera model whether the 3-instruction division by a constant reproduces the
// IEEE quotient on every dividend A1 can produce (k_verify_fast_div); the result is cached.
int check_fast_div(hope_ctx* c, const DepthSrc& src)
{
  const float key[9] = {src.fx, src.fy, src.cx, src.cy, src.scale, src.min_d, src.max_d, (float)src.w, (float)src.h};
  if (std::memcmp(key, c->fast_div_key, sizeof(key)) == 0) return HOPE_OK;
  c->fast_div = 0;
  if (!c->force_ieee_div) {
    HIPCHK(hipMemsetAsync(&c->d_glob->changed, 0, sizeof(unsigned), c->stream));
    hipLaunchKernelGGL(k_verify_fast_div, dim3(65536 / kBlock, std::max(src.w, src.h)), dim3(kBlock), 0, c->stream, src,
                       &c->d_glob->changed);
    int st = sync_globals(c);
    if (st != HOPE_OK) return st;
    c->fast_div = c->h_glob->changed ? 0 : 1;
  }
  std::memcpy(c->fast_div_key, key, sizeof(key));
  return HOPE_OK;
}

constexpr int kRetryPerPoint = -1000;  // internal: never leaves the library

template <class Src>
int run_pipeline(hope_ctx* c, const Src& src, size_t n_per_frame)
{
  const int F = c->n_frames;
  hipStream_t s = c->stream;
  const auto t0 = std::chrono::steady_clock::now();
  tl_mark(c, "run_pipeline: enter");
  c->ref_ms = 0;
  c->prof.recs.clear();
  c->prof.used = 0;
  c->n_vox_total = c->n_fit_total = c->n_cl_total = 0;
  c->one_pass = false;
  c->dense = false;
  c->d_frame_order = nullptr;
  c->d_slots = nullptr;
  c->h_rec.clear();
  c->h_nplanes.assign(F, 0);
  c->h_max.assign(F, hope_max_plane{});
  c->h_max_members.assign(F, std::vector<int32_t>());
  c->d_vox = nullptr;
  c->d_normals = nullptr;
  c->d_fit_idx = nullptr;
  c->d_knn = nullptr;
  c->d_cid = nullptr;
  c->d_members = nullptr;
  c->d_fit_pt = nullptr;
  c->d_cl_first = nullptr;
  if ((unsigned long long)n_per_frame * (unsigned long long)F >= 0xffffffffull) {
    c->err = "input larger than 2^32-1 points per call is not supported";
    return HOPE_E_RANGE;
  }

  // per-frame state
  c->h_fi.assign(F, FrameInfo{});
  for (int f = 0; f < F; ++f) {
    FrameInfo& fi = c->h_fi[f];
    make_affine(c->frame_poses.empty() ? c->pose : c->frame_poses[f], fi.aff);
    for (int d = 0; d < 3; ++d) {
      fi.bb_min[d] = 0xffffffffu;
      fi.bb_max[d] = 0u;
    }
  }
  FrameInfo* d_fi;
  ALLOC(d_fi, FrameInfo, F);
  c->d_fi = d_fi;
  HIPCHK(hipMemcpyAsync(d_fi, c->h_fi.data(), sizeof(FrameInfo) * F, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemsetAsync(c->d_glob, 0, sizeof(Globals), s));
  tl_mark(c, "frame records uploaded");

  // ---- A1..A3 + bbox ------------------------------------------------------------------
  int st;
  if (n_per_frame > 0) {
    size_t pm = prof_mark(c);
    st = launch_bbox(c, src, n_per_frame);
    if (st != HOPE_OK) return st;
    prof_close(c, ST_BBOX, pm, src_point_bytes(src) * n_per_frame * F);
  }
  if (c->forced_bbox) {
    const float* b = c->forced_bbox;
    hipLaunchKernelGGL(k_force_bbox, dim3(1), dim3(1), 0, s, d_fi, b[0], b[1], b[2], b[3], b[4], b[5]);
  }
  {
    size_t pm = prof_mark(c);
    hipLaunchKernelGGL(k_frame_grid, dim3(1), dim3(1024), 0, s, d_fi, F, c->inv_xy, c->inv_z, (unsigned)n_per_frame,
                       c->d_glob);
    prof_close(c, ST_GRID, pm, (double)sizeof(FrameInfo) * F);
    pm = prof_mark(c);
    st = scan_groups(c, src, n_per_frame);
    if (st != HOPE_OK) return st;
    if (use_groups(c, src)) prof_close(c, ST_VKEYS, pm, 8.0 * ((double)c->cpf * F));
  }
  st = sync_globals(c);
  if (st != HOPE_OK) return st;
  if (c->h_glob->any_overflow & ~1u) {
    c->err = (c->h_glob->any_overflow & 2u)
                 ? "leaf size too small for the input extent: more than 2^30 columns of th_xy in a frame, or cell "
                   "coordinates beyond 2^30 (PCL would pass the cloud through; no column index of that size is built)"
                 : "the column tables of this batch exceed 2^31 entries: pass fewer frames per call";
    return HOPE_E_RANGE;
  }
  if (c->h_glob->any_overflow & 1u) {
    // a frame with more than INT_MAX cells: PCL's VoxelGrid copies its input through.  Here every
    // point of such a frame becomes a voxel of its own in the per-point voxel stage; a depth batch on
    // the group path starts again on that stage (kRetryPerPoint: see hope_segment_depth_u16)
    if (c->forced_bbox || c->input_is_downsampled) {
      c->err = "staged cloud call: VoxelGrid pass-through (more than INT_MAX cells) is not supported across ranks";
      return HOPE_E_RANGE;
    }
    if (use_groups(c, src)) return kRetryPerPoint;
  }

  // ---- A5 voxel grid ------------------------------------------------------------------
  const int cell_bits = bits_for(c->h_glob->max_cells - 1);
  const int total_bits = cell_bits + bits_for((unsigned long long)F);
  float4* d_vox = nullptr;
  int* d_vox_frame = nullptr;
  if (c->dense && (c->h_glob->code_overflow || c->h_glob->max_cells > (1ull << 24) || total_bits > 32)) {
    st = dense_fallback(c, src, n_per_frame);
    if (st != HOPE_OK) return st;
  }
  if (n_per_frame > 0 && c->input_is_downsampled) {
    st = adopt_downsampled(c, src, n_per_frame, d_vox, d_vox_frame);
    if (st != HOPE_OK) return st;
  } else if (n_per_frame > 0) {
    if (total_bits <= 32)
      st = stage_voxel<Src, uint32_t>(c, src, n_per_frame, cell_bits, total_bits, d_vox, d_vox_frame);
    else
      st = stage_voxel<Src, uint64_t>(c, src, n_per_frame, cell_bits, total_bits, d_vox, d_vox_frame);
    if (st != HOPE_OK) return st;
  }
  c->d_vox = d_vox;

  // ---- A6..A12 ------------------------------------------------------------------------
  bool timed = false;
  if (c->n_vox_total > 0 && !c->stop_after_voxel) {
    const int zbits = bits_for(c->h_glob->max_div_z - 1);
    const int key2_bits = zbits + bits_for(c->h_glob->total_cols);
    if (key2_bits <= 32)
      st = stage_rest<uint32_t>(c, d_vox, d_vox_frame, zbits, key2_bits);
    else
      st = stage_rest<uint64_t>(c, d_vox, d_vox_frame, zbits, key2_bits);
    if (st != HOPE_OK) return st;
    timed = c->n_fit_total > 0 && c->stop_stage == 0;  // (an early stop never records the closing event)
  }
  tl_mark(c, "all launched");
  HIPCHK(hipMemcpyAsync(c->h_fi.data(), d_fi, sizeof(FrameInfo) * F, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  tl_mark(c, "final synchronise done");
  prof_collect(c);
  tl_mark(c, "prof_collect done");
  if (timed) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev_a, c->ev_b) == hipSuccess) c->ref_ms = ms;
    else (void)hipGetLastError();  // (never leave a sticky error behind for the caller's own HIP calls)
  }
  if (c->n_raw_override >= 0) c->h_fi[0].n_raw = (int)c->n_raw_override;
  if (c->n_vox_total == 0)
    for (FrameInfo& fi : c->h_fi) fi.n_vox = fi.n_fit = fi.n_segments = fi.n_clusters = 0;
  else if (c->n_fit_total == 0)
    for (FrameInfo& fi : c->h_fi) fi.n_segments = fi.n_clusters = 0;
  rt_post(c);
  tl_mark(c, "rt_post done");
  int worst = HOPE_OK;
  for (FrameInfo& fi : c->h_fi) {
    fi.status = fi.n_vox == 0 ? HOPE_EMPTY
                              : ((fi.n_fit == 0 && !c->stop_after_voxel && c->stop_stage != 1) ? HOPE_NO_NORMAL_FIT : HOPE_OK);
    worst = std::max(worst, fi.status);
  }
  c->whole_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  tl_mark(c, "return");
  tl_print(c);
  return worst;
}

}  // namespace

/* =====================================================================================
 * C ABI
 * =================================================================================== */
extern "C" {

HOPE_API int hope_config_default(hope_config* cfg)
{
  if (!cfg) return HOPE_E_INVALID;
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->th_xy = 0.05f;
  cfg->th_z = 0.02f;
  cfg->min_depth = 0.3f;
  cfg->max_depth = 8.0f;
  cfg->fx = 517.0f;
  cfg->fy = 517.0f;
  cfg->cx = 318.6f;
  cfg->cy = 255.3f;
  cfg->depth_scale = 5000.0f;
  cfg->knn = 8;
  cfg->min_cluster = 3;
  cfg->mode = HOPE_MODE_OFFLINE;
  cfg->min_height = 0.8f;
  cfg->max_height = 1.5f;
  cfg->aggressive_merge = 1;
  cfg->device = 0;
  return HOPE_OK;
}

HOPE_API int hope_create(const hope_config* cfg, hope_ctx** out)
{
  if (!cfg || !out) return HOPE_E_INVALID;
  if (!(cfg->th_xy > 0.f) || !(cfg->th_z > 0.f) || cfg->knn < 1 || cfg->knn > kMaxKnn || cfg->min_cluster < 1)
    return HOPE_E_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || cfg->device < 0 || cfg->device >= ndev) return HOPE_E_HIP;
  if (hipSetDevice(cfg->device) != hipSuccess) return HOPE_E_HIP;
  hope_ctx* c = new hope_ctx();
  c->cfg = *cfg;
  derive_thresholds(c);
  c->pose = Pose{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 1.f}};
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&c->ev_a) != hipSuccess || hipEventCreate(&c->ev_b) != hipSuccess ||
      hipMalloc((void**)&c->d_glob, sizeof(Globals)) != hipSuccess ||
      hipHostMalloc((void**)&c->h_glob, sizeof(Globals) + 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
    delete c;
    return HOPE_E_HIP;
  }
  std::memset(c->h_glob, 0, sizeof(Globals) + 64);
  c->h_seq = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(c->h_glob) + (sizeof(Globals) + 15) / 16 * 16);
  {
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, c->h_glob, 0) == hipSuccess && dp) {
      c->h_glob_dev = static_cast<Globals*>(dp);
      c->h_seq_dev = reinterpret_cast<unsigned*>(static_cast<char*>(dp) + (reinterpret_cast<char*>(c->h_seq) - reinterpret_cast<char*>(c->h_glob)));
    } else {
      (void)hipGetLastError();
      c->mail_poll = 0;
    }
    const char* dk = std::getenv("HOPE_DEBUG");
    const char* sk = (dk && dk[0] == '1') ? std::getenv("HOPE_DEBUG_SYNC") : nullptr;
    if (sk && std::strcmp(sk, "memcpy") == 0) c->mail_poll = 0;
  }
  {
    // the device's address of the static scratch array, once per device, with that device current
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (cfg->device < kMaxDevices && !g_scratch_ptr[cfg->device]) {
      void* p = nullptr;
      const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
      if (e != hipSuccess || !p) {
        // only hope_shard_z_histogram_static needs it: that entry point reports the missing scratch, every other
        // one works without
        std::fprintf(stderr, "hope_create: hipGetSymbolAddress(scratch) failed: %s\n", hipGetErrorString(e));
        (void)hipGetLastError();
      } else {
        g_scratch_ptr[cfg->device] = static_cast<StaticScratch*>(p);
      }
    }
    c->d_scratch = cfg->device < kMaxDevices ? g_scratch_ptr[cfg->device] : nullptr;
  }
  // Test knobs: every HOPE_DEBUG_* variable below forces a path the data rarely takes (results are the same on
  // every path).  They are read only when HOPE_DEBUG=1 is set as well, so that a stray variable in a production
  // environment cannot change which kernels run.
  const char* dbg_on = std::getenv("HOPE_DEBUG");
  const bool dbg = dbg_on && dbg_on[0] == '1';
  const auto debug_env = [dbg](const char* name) -> const char* { return dbg ? std::getenv(name) : nullptr; };
  if (cfg->sub_batches >= 1) c->n_sub = std::min((int)cfg->sub_batches, 16);
  const char* knob = debug_env("HOPE_DEBUG_SMALL_NB_CAP");
  c->normals_cap_small = (knob && knob[0] == '1') ? 1 : 0;
#ifdef HOPE_ABLATE  // timing experiments (tools/ablate.sh builds its own library): never in the shipped build
  knob = debug_env("HOPE_DEBUG_ABLATE");
  c->ablate = knob ? std::atoi(knob) : 0;
#endif
  knob = debug_env("HOPE_DEBUG_KNN_FINE_LIMIT");
  if (knob) c->knn_fine_limit = (unsigned)std::atoll(knob);
  knob = debug_env("HOPE_DEBUG_IEEE_DIV");
  c->force_ieee_div = (knob && knob[0] == '1') ? 1 : 0;
  knob = debug_env("HOPE_DEBUG_PIXEL_SORT");
  c->pixel_sort = (knob && knob[0] == '1') ? 1 : 0;
  knob = debug_env("HOPE_DEBUG_TWO_PASS");
  c->two_pass = (knob && knob[0] == '1') ? 1 : 0;
  knob = debug_env("HOPE_DEBUG_COLSORT_Z");
  c->col_sort_z = (knob && knob[0] == '1') ? 1 : 0;
  knob = debug_env("HOPE_DEBUG_SUB_MIN");
  if (knob && std::atoi(knob) >= 1) c->sub_min_frames = std::atoi(knob);
  knob = debug_env("HOPE_DEBUG_DENSE_RECORDS");
  c->dense_records = (knob && knob[0] == '0') ? 0 : 1;
  knob = debug_env("HOPE_DEBUG_CAP_MUL");
  if (knob && std::atoi(knob) >= 1 && std::atoi(knob) <= kChunk) c->cap_mul = std::atoi(knob);
  knob = debug_env("HOPE_DEBUG_MEMBERS_SORT");  // 1: always the sort; 0: never (when the clusters fit), however few frames
  c->members_by_frame = knob ? (knob[0] == '1' ? 0 : 2) : 1;
  knob = debug_env("HOPE_DEBUG_FRAME_ORDER");
  c->frame_order = (knob && knob[0] == '0') ? 0 : 1;
  knob = debug_env("HOPE_DEBUG_UNION_TILES");
  c->union_small_tiles = knob ? atoi(knob) : 1;  // (a number of points from kUfBlock up: that tile)
  knob = debug_env("HOPE_DEBUG_TIMELINE");
  c->timeline = (knob && knob[0] == '1') ? 1 : 0;
  knob = debug_env("HOPE_DEBUG_SEGSORT");
  c->seg_sort = (knob && knob[0] == '0') ? 0 : 1;
  knob = debug_env("HOPE_DEBUG_COL_INDEX");
  c->col_index_lds = (knob && knob[0] == '0') ? 0 : 1;
  knob = debug_env("HOPE_DEBUG_TILE_SIDE");
  if (knob) c->tile_side = std::atoi(knob) == kSuper ? kSuper : (std::atoi(knob) == kTile ? kTile : 0);
  knob = debug_env("HOPE_DEBUG_TODO_THREADS");
  if (knob) c->todo_threads = std::atoi(knob);
  knob = debug_env("HOPE_DEBUG_NEIGHBOURS");  // tiles | tiles-normals | tiles-knn
  if (knob && std::strncmp(knob, "tiles", 5) == 0) {
    c->normals_direct = std::strcmp(knob, "tiles-knn") == 0 ? 1 : 0;
    c->knn_direct = std::strcmp(knob, "tiles-normals") == 0 ? 1 : 0;
  }
  knob = debug_env("HOPE_DEBUG_VOXEL_PART");
  c->voxel_part = (knob && knob[0] == '0') ? 0 : 1;
  knob = debug_env("HOPE_DEBUG_GRID_INDEX");
  c->grid_index = (knob && knob[0] == '0') ? 0 : 1;
  knob = debug_env("HOPE_DEBUG_VERIFY_BIG");
  if (knob) c->verify_big_from = std::atoi(knob);
  knob = debug_env("HOPE_DEBUG_VOXEL_RANK");
  c->voxel_rank = (knob && knob[0] == '0') ? 0 : c->seg_sort;
  *out = c;
  return HOPE_OK;
}

HOPE_API void hope_destroy(hope_ctx* c)
{
  if (!c) return;
  for (hope_ctx* k : c->kids) hope_destroy(k);
  c->kids.clear();
  (void)hipSetDevice(c->cfg.device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  c->arena.release();
  if (c->d_glob) (void)hipFree(c->d_glob);
  if (c->d_xs_stats) (void)hipFree(c->d_xs_stats);
  if (c->h_glob) (void)hipHostFree(c->h_glob);
  for (hipEvent_t e : c->prof.ev) (void)hipEventDestroy(e);
  if (c->ev_a) (void)hipEventDestroy(c->ev_a);
  if (c->ev_b) (void)hipEventDestroy(c->ev_b);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

HOPE_API const char* hope_last_error(const hope_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

HOPE_API int hope_set_pose(hope_ctx* c, const float t[3], const float q[4])
{
  if (!c || !t || !q) return HOPE_E_INVALID;
  for (int i = 0; i < 3; ++i) c->pose.t[i] = t[i];
  for (int i = 0; i < 4; ++i) c->pose.q[i] = q[i];
  c->frame_poses.clear();
  return HOPE_OK;
}

// The pose Transform::doTransform(cloud, out, roll, pitch, yaw) builds (transform.cpp:108-130): translation
// (0, 0, 1) m (transform.cpp:4-6), rotation tf2's setRPY in double with y negated, narrowed to float.
HOPE_API int hope_set_pose_rpy(hope_ctx* c, float roll, float pitch, float yaw)
{
  if (!c) return HOPE_E_INVALID;
  const double hr = 0.5 * (double)roll, hp = 0.5 * (double)pitch, hy = 0.5 * (double)yaw;
  const double cr = std::cos(hr), sr = std::sin(hr), cp = std::cos(hp), sp = std::sin(hp), cy = std::cos(hy),
               sy = std::sin(hy);
  const double x = sr * cp * cy - cr * sp * sy, y = cr * sp * cy + sr * cp * sy, z = cr * cp * sy - sr * sp * cy,
               w = cr * cp * cy + sr * sp * sy;
  const float t[3] = {0.f, 0.f, 1.f}, q[4] = {(float)x, (float)-y, (float)z, (float)w};
  return hope_set_pose(c, t, q);
}

HOPE_API int hope_set_frame_poses(hope_ctx* c, int n, const float* t, const float* q)
{
  if (!c || n < 0 || (n > 0 && (!t || !q))) return HOPE_E_INVALID;
  c->frame_poses.resize(n);
  for (int f = 0; f < n; ++f) {
    for (int i = 0; i < 3; ++i) c->frame_poses[f].t[i] = t[3 * f + i];
    for (int i = 0; i < 4; ++i) c->frame_poses[f].q[i] = q[4 * f + i];
  }
  return HOPE_OK;
}

HOPE_API int hope_set_height_window(hope_ctx* c, float mn, float mx)
{
  if (!c) return HOPE_E_INVALID;
  c->cfg.min_height = mn;
  c->cfg.max_height = mx;
  return HOPE_OK;
}

HOPE_API int hope_segment_depth_u16(hope_ctx* c, const uint16_t* depth, int w, int h, int n_frames, int on_device);

// A depth batch in n_sub ranges of frames, one child context and one host thread per range.
static int segment_depth_split(hope_ctx* c, const uint16_t* depth, int w, int h, int n_frames, int on_device, int n_sub)
{
  const auto t0 = std::chrono::steady_clock::now();
  while ((int)c->kids.size() < n_sub) {
    hope_ctx* k = nullptr;
    const int st = hope_create(&c->cfg, &k);
    if (st != HOPE_OK) {
      c->err = "could not create a sub-batch context";
      return st;
    }
    k->n_sub = 1;
    c->kids.push_back(k);
  }
  std::vector<int> first(n_sub + 1);
  for (int k = 0; k <= n_sub; ++k) first[k] = (int)((long long)n_frames * k / n_sub);
  const size_t ppf = (size_t)w * h;
  std::vector<int> status(n_sub, HOPE_OK);
  std::vector<std::thread> th;
  for (int k = 0; k < n_sub; ++k) {
    hope_ctx* kid = c->kids[k];
    kid->cfg = c->cfg;  // (the height window may have moved since the kid was made)
    kid->pose = c->pose;
    kid->frame_poses.clear();
    if (!c->frame_poses.empty())
      kid->frame_poses.assign(c->frame_poses.begin() + first[k], c->frame_poses.begin() + first[k + 1]);
    kid->prof.on = c->prof.on;
    th.emplace_back([=, &status]() {
      status[k] = hope_segment_depth_u16(kid, depth + (size_t)first[k] * ppf, w, h, first[k + 1] - first[k], on_device);
    });
  }
  for (std::thread& t : th) t.join();
  int worst = HOPE_OK;
  for (int k = 0; k < n_sub; ++k) {
    if (status[k] < 0) {
      c->err = c->kids[k]->err;
      return status[k];
    }
    worst = std::max(worst, status[k]);
  }
  c->kid_first = first;
  c->n_frames = n_frames;
  c->src_kind = 1;
  c->h_fi.clear();
  c->ref_ms = 0;
  for (int k = 0; k < n_sub; ++k) c->ref_ms += c->kids[k]->ref_ms;
  c->whole_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return worst;
}

// frame of a split call -> (kid, frame within the kid)
static bool route_frame(hope_ctx* c, int frame, hope_ctx*& kid, int& local)
{
  if (frame < 0 || frame >= c->kid_first.back()) return false;
  const int k = (int)(std::upper_bound(c->kid_first.begin(), c->kid_first.end(), frame) - c->kid_first.begin()) - 1;
  kid = c->kids[k];
  local = frame - c->kid_first[k];
  return true;
}

// first statement of every per-frame accessor: after a split call the frame lives in a kid
#define ROUTE(c, frame, CALL)                                \
  if ((c) && !(c)->kid_first.empty()) {                      \
    hope_ctx* kid__ = nullptr;                               \
    int local__ = 0;                                         \
    if (!route_frame((c), (frame), kid__, local__)) {        \
      (c)->err = "frame index out of range";                 \
      return HOPE_E_RANGE;                                   \
    }                                                        \
    const int st__ = CALL;                                   \
    if (st__ < 0) (c)->err = kid__->err;                     \
    return st__;                                             \
  }

HOPE_API int hope_segment_depth_u16(hope_ctx* c, const uint16_t* depth, int w, int h, int n_frames, int on_device)
{
  if (!c) return HOPE_E_INVALID;
  if (!depth || w <= 0 || h <= 0 || n_frames <= 0) {
    c->err = "hope_segment_depth_u16: bad argument";
    return HOPE_E_INVALID;
  }
  if (!c->frame_poses.empty() && (int)c->frame_poses.size() != n_frames) {
    c->err = "frame pose count does not match n_frames";
    return HOPE_E_INVALID;
  }
  HIPCHK(hipSetDevice(c->cfg.device));
  {
    const int n_sub = std::min(c->n_sub, n_frames / std::max(c->sub_min_frames, 1));
    if (n_sub > 1) {
      c->kid_first.clear();
      c->upon_kid = nullptr;
      return segment_depth_split(c, depth, w, h, n_frames, on_device, n_sub);
    }
  }
  c->arena.reset();
  c->upon_mark_valid = false;
  c->kid_first.clear();
  c->upon_kid = nullptr;
  c->upon.clear();
  c->n_frames = n_frames;
  c->src_kind = 1;
  const size_t ppf = (size_t)w * h;
  const uint16_t* d_depth = depth;
  if (!on_device) {
    uint16_t* buf;
    ALLOC(buf, uint16_t, ppf * n_frames);
    HIPCHK(hipMemcpyAsync(buf, depth, sizeof(uint16_t) * ppf * n_frames, hipMemcpyHostToDevice, c->stream));
    d_depth = buf;
  }
  DepthSrc src;
  src.depth = d_depth;
  src.w = w;
  src.h = h;
  src.ppf = (unsigned)ppf;
  src.fx = c->cfg.fx;
  src.fy = c->cfg.fy;
  src.cx = c->cfg.cx;
  src.cy = c->cfg.cy;
  src.scale = c->cfg.depth_scale;
  src.min_d = c->cfg.min_depth;
  src.max_d = c->cfg.max_depth;
  src.rcp_fx = 1.0f / src.fx;
  src.rcp_fy = 1.0f / src.fy;
  src.rcp_scale = 1.0f / src.scale;
  c->dsrc = src;
  int st = check_fast_div(c, src);
  if (st != HOPE_OK) return st;
  const std::vector<size_t> mark = c->arena.mark();
  st = run_pipeline(c, src, ppf);
  if (st == kRetryPerPoint) {
    const int knob = c->pixel_sort;
    c->pixel_sort = 1;
    c->arena.rollback(mark);
    st = run_pipeline(c, src, ppf);
    c->pixel_sort = knob;
  }
  return st;
}

HOPE_API int hope_segment_cloud_xyz(hope_ctx* c, const void* xyz, size_t n, size_t stride, int apply, int on_device)
{
  if (!c) return HOPE_E_INVALID;
  if ((!xyz && n > 0) || stride < 12 || (stride % 4) != 0) {
    c->err = "hope_segment_cloud_xyz: bad argument";
    return HOPE_E_INVALID;
  }
  HIPCHK(hipSetDevice(c->cfg.device));
  c->arena.reset();
  c->upon_mark_valid = false;
  c->kid_first.clear();
  c->upon_kid = nullptr;
  c->upon.clear();
  c->n_frames = 1;
  c->src_kind = 2;
  const void* d_xyz = xyz;
  if (!on_device && n > 0) {
    char* buf;
    ALLOC(buf, char, n * stride);
    HIPCHK(hipMemcpyAsync(buf, xyz, n * stride, hipMemcpyHostToDevice, c->stream));
    d_xyz = buf;
  }
  std::vector<Pose> saved;
  saved.swap(c->frame_poses);  // a cloud call always uses the single pose
  CloudSrc src;
  src.base = static_cast<const char*>(d_xyz);
  src.stride = stride;
  src.n = (unsigned)n;
  src.apply = apply ? 1 : 0;
  src.min_d = c->cfg.min_depth;
  src.max_d = c->cfg.max_depth;
  c->csrc = src;
  const int st = run_pipeline(c, src, n);
  c->frame_poses.swap(saved);
  return st;
}

/* ---- one cloud across several GPUs ------------------------------------------------------------ */
namespace {

// device-resident view of a caller's cloud (copied up when it lives on the host)
int make_cloud_src(hope_ctx* c, const void* xyz, size_t n, size_t stride, int apply, int on_device, CloudSrc& src)
{
  if ((!xyz && n > 0) || stride < 12 || (stride % 4) != 0 || n >= 0xffffffffull) {
    c->err = "bad cloud argument";
    return HOPE_E_INVALID;
  }
  HIPCHK(hipSetDevice(c->cfg.device));
  const void* d_xyz = xyz;
  if (!on_device && n > 0) {
    char* buf;
    ALLOC(buf, char, n * stride);
    HIPCHK(hipMemcpyAsync(buf, xyz, n * stride, hipMemcpyHostToDevice, c->stream));
    d_xyz = buf;
  }
  src.base = static_cast<const char*>(d_xyz);
  src.stride = stride;
  src.n = (unsigned)n;
  src.apply = apply ? 1 : 0;
  src.min_d = c->cfg.min_depth;
  src.max_d = c->cfg.max_depth;
  return HOPE_OK;
}

// FrameInfo[1] with the pose and either an empty or a forced bounding box; grid derived on device
int shard_frame(hope_ctx* c, const float* bbox)
{
  c->n_frames = 1;
  c->h_fi.assign(1, FrameInfo{});
  FrameInfo& fi = c->h_fi[0];
  make_affine(c->pose, fi.aff);
  for (int d = 0; d < 3; ++d) {
    fi.bb_min[d] = 0xffffffffu;
    fi.bb_max[d] = 0u;
  }
  FrameInfo* d_fi;
  ALLOC(d_fi, FrameInfo, 1);
  c->d_fi = d_fi;
  HIPCHK(hipMemcpyAsync(d_fi, c->h_fi.data(), sizeof(FrameInfo), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(c->d_glob, 0, sizeof(Globals), c->stream));
  if (bbox) {
    hipLaunchKernelGGL(k_force_bbox, dim3(1), dim3(1), 0, c->stream, d_fi, bbox[0], bbox[1], bbox[2], bbox[3], bbox[4],
                       bbox[5]);
    // k_frame_grid only builds a grid for a frame with points
    const int one = 1;
    HIPCHK(hipMemcpyAsync(&d_fi->n_raw, &one, sizeof(int), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_frame_grid, dim3(1), dim3(1024), 0, c->stream, d_fi, 1, c->inv_xy, c->inv_z, 0u, c->d_glob);
  }
  return HOPE_OK;
}

}  // namespace

HOPE_API int hope_grid_from_bbox(hope_ctx* c, const float bbox[6], int32_t min_b[3], int32_t div_b[3])
{
  if (!c || !bbox || !min_b || !div_b) return HOPE_E_INVALID;
  // VoxelGrid::applyFilter (PCL 1.8.1 voxel_grid.hpp): the arithmetic of k_frame_grid on the host
  const float inv[3] = {c->inv_xy, c->inv_xy, c->inv_z};
  long long dxyz = 1;
  for (int d = 0; d < 3; ++d) {
    dxyz *= (long long)((bbox[3 + d] - bbox[d]) * inv[d]) + 1;
    min_b[d] = (int)std::floor(bbox[d] * inv[d]);
    div_b[d] = (int)std::floor(bbox[3 + d] * inv[d]) - min_b[d] + 1;
  }
  return dxyz > 2147483647LL ? HOPE_E_RANGE : HOPE_OK;
}

HOPE_API int hope_shard_bbox(hope_ctx* c, const void* xyz, size_t n, size_t stride, int apply, int on_device,
                             float bbox[6], int64_t* n_raw)
{
  if (!c || !bbox || !n_raw) return HOPE_E_INVALID;
  c->src_kind = 0;  // (no segment result: the frame accessors that re-read the input refuse)
  c->arena.reset();
  c->upon_mark_valid = false;
  c->kid_first.clear();
  c->upon_kid = nullptr;
  c->upon.clear();
  CloudSrc src;
  int st = make_cloud_src(c, xyz, n, stride, apply, on_device, src);
  if (st != HOPE_OK) return st;
  st = shard_frame(c, nullptr);
  if (st != HOPE_OK) return st;
  if (n > 0) launch_bbox(c, src, n);
  HIPCHK(hipMemcpyAsync(c->h_fi.data(), c->d_fi, sizeof(FrameInfo), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  const FrameInfo& fi = c->h_fi[0];
  *n_raw = fi.n_raw;
  auto ord2f_host = [](unsigned u) {
    const unsigned v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    std::memcpy(&f, &v, 4);
    return f;
  };
  for (int d = 0; d < 3; ++d) {
    bbox[d] = fi.n_raw > 0 ? ord2f_host(fi.bb_min[d]) : INFINITY;
    bbox[3 + d] = fi.n_raw > 0 ? ord2f_host(fi.bb_max[d]) : -INFINITY;
  }
  return HOPE_OK;
}

HOPE_API int hope_shard_z_histogram(hope_ctx* c, const void* xyz, size_t n, size_t stride, int apply, int on_device,
                                    const float bbox[6], int64_t* hist, int n_bins)
{
  if (!c || !bbox || !hist || n_bins <= 0) return HOPE_E_INVALID;
  c->src_kind = 0;  // (no segment result: the frame accessors that re-read the input refuse)
  c->arena.reset();
  c->upon_mark_valid = false;
  c->kid_first.clear();
  c->upon_kid = nullptr;
  c->upon.clear();
  CloudSrc src;
  int st = make_cloud_src(c, xyz, n, stride, apply, on_device, src);
  if (st != HOPE_OK) return st;
  st = shard_frame(c, bbox);
  if (st != HOPE_OK) return st;
  unsigned long long* d_hist;
  ALLOC(d_hist, unsigned long long, n_bins);
  HIPCHK(hipMemsetAsync(d_hist, 0, sizeof(unsigned long long) * n_bins, c->stream));
  if (n > 0) {
    const unsigned bx = std::max(1u, std::min<unsigned>(nblk(n, kBlock * 16), 2048u));
    hipLaunchKernelGGL((k_z_histogram<CloudSrc>), dim3(bx), dim3(kBlock), 0, c->stream, src, c->d_fi, c->inv_z, n_bins,
                       d_hist);
  }
  HIPCHK(hipMemcpyAsync(hist, d_hist, sizeof(int64_t) * n_bins, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return HOPE_OK;
}

HOPE_API int hope_shard_z_histogram_static(hope_ctx* c, const void* xyz_dev, size_t n, size_t stride, int apply,
                                           const float bbox[6], int64_t* hist, int n_bins, int hist_on_device)
{
  if (!c || !bbox || !hist || n_bins <= 0) return HOPE_E_INVALID;
  if ((!xyz_dev && n > 0) || stride < 12 || (stride % 4) != 0 || n >= 0xffffffffull) {
    c->err = "hope_shard_z_histogram_static: bad cloud argument";
    return HOPE_E_INVALID;
  }
  if (n_bins > kStaticHistBins) {
    c->err = "hope_shard_z_histogram_static: more than 8192 z layers do not fit the static scratch buffer "
             "(use hope_shard_z_histogram)";
    return HOPE_E_RANGE;
  }
  if (!c->d_scratch) {
    c->err = "hope_shard_z_histogram_static: no static scratch on this device";
    return HOPE_E_HIP;
  }
  HIPCHK(hipSetDevice(c->cfg.device));
  std::lock_guard<std::mutex> in_use(g_scratch_use[c->cfg.device]);  // (one call at a time per GPU: the array is shared)
  StaticScratch* sp = c->d_scratch;  // (the device address: never the array's name)
  hipStream_t s = c->stream;
  CloudSrc src;
  src.base = static_cast<const char*>(xyz_dev);
  src.stride = stride;
  src.n = (unsigned)n;
  src.apply = apply ? 1 : 0;
  src.min_d = c->cfg.min_depth;
  src.max_d = c->cfg.max_depth;
  // every call on this GPU shares the array, which keeps the last call's contents: the frame record is
  // written whole, the scalars and the bins in use are cleared
  StaticFrame sf;
  make_affine(c->pose, sf.aff);
  for (int d = 0; d < 6; ++d) sf.bbox[d] = bbox[d];
  hipLaunchKernelGGL(k_static_frame, dim3(1), dim3(1), 0, s, &sp->fi, sf);
  HIPCHK(hipMemsetAsync(&sp->glob, 0, sizeof(Globals), s));
  HIPCHK(hipMemsetAsync(sp->hist, 0, sizeof(unsigned long long) * (size_t)n_bins, s));
  hipLaunchKernelGGL(k_frame_grid, dim3(1), dim3(1024), 0, s, &sp->fi, 1, c->inv_xy, c->inv_z, 0u, &sp->glob);
  if (n > 0) {
    const unsigned bx = std::max(1u, std::min<unsigned>(nblk(n, kBlock * 16), 2048u));
    hipLaunchKernelGGL((k_z_histogram<CloudSrc>), dim3(bx), dim3(kBlock), 0, s, src, &sp->fi, c->inv_z, n_bins, sp->hist);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(hist, sp->hist, sizeof(int64_t) * (size_t)n_bins,
                        hist_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return HOPE_OK;
}

HOPE_API int hope_shard_route(hope_ctx* c, const void* xyz, size_t n, size_t stride, int apply, int on_device,
                              const float bbox[6], const int32_t* layer_owner, int n_bins, int n_ranks, float* out_xyz,
                              int out_on_device, int64_t* counts)
{
  if (!c || !bbox || !layer_owner || n_bins <= 0 || n_ranks <= 0 || !counts) return HOPE_E_INVALID;
  c->src_kind = 0;  // (no segment result: the frame accessors that re-read the input refuse)
  c->arena.reset();
  c->upon_mark_valid = false;
  c->kid_first.clear();
  c->upon_kid = nullptr;
  c->upon.clear();
  CloudSrc src;
  int st = make_cloud_src(c, xyz, n, stride, apply, on_device, src);
  if (st != HOPE_OK) return st;
  st = shard_frame(c, bbox);
  if (st != HOPE_OK) return st;
  for (int r = 0; r < n_ranks; ++r) counts[r] = 0;
  if (n == 0) return HOPE_OK;
  PrimAlloc pa{arena_alloc_cb, &c->arena};
  int* d_owner;
  uint32_t *k_in, *k_out, *v_in, *v_out;
  unsigned long long* d_counts;
  float* d_out;
  ALLOC(d_owner, int, n_bins);
  ALLOC(k_in, uint32_t, n);
  ALLOC(k_out, uint32_t, n);
  ALLOC(v_in, uint32_t, n);
  ALLOC(v_out, uint32_t, n);
  ALLOC(d_counts, unsigned long long, n_ranks + 2);
  HIPCHK(hipMemcpyAsync(d_owner, layer_owner, sizeof(int) * n_bins, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL((k_route_keys<CloudSrc>), dim3(nblk(n)), dim3(kBlock), 0, c->stream, src, c->d_fi, c->inv_z,
                     d_owner, n_bins, n_ranks, k_in, v_in);
  HIPCHK(sort_pairs_u32(pa, k_in, k_out, v_in, v_out, n, bits_for((unsigned long long)n_ranks), c->stream));
  hipLaunchKernelGGL(k_route_counts, dim3(1), dim3(64), 0, c->stream, k_out, (unsigned)n, n_ranks, d_counts);
  if (out_on_device) {
    d_out = out_xyz;
  } else {
    ALLOC(d_out, float, 3 * n);
  }
  hipLaunchKernelGGL((k_route_export<CloudSrc>), dim3(nblk(n)), dim3(kBlock), 0, c->stream, src, c->d_fi, k_out, v_out,
                     n_ranks, d_out);
  std::vector<unsigned long long> h_counts(n_ranks + 1);
  HIPCHK(hipMemcpyAsync(h_counts.data(), d_counts, sizeof(unsigned long long) * (n_ranks + 1), hipMemcpyDeviceToHost,
                        c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int r = 0; r < n_ranks; ++r) counts[r] = (int64_t)(h_counts[r + 1] - h_counts[r]);
  if (!out_on_device && h_counts[n_ranks] > 0) {
    HIPCHK(hipMemcpyAsync(out_xyz, d_out, sizeof(float) * 3 * h_counts[n_ranks], hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  return HOPE_OK;
}

static int run_cloud_staged(hope_ctx* c, const void* xyz, size_t n, size_t stride, int on_device, const float bbox[6],
                            int stop_after_voxel, int is_downsampled, long long n_raw_override)
{
  c->arena.reset();
  c->upon_mark_valid = false;
  c->kid_first.clear();
  c->upon_kid = nullptr;
  c->upon.clear();
  c->n_frames = 1;
  c->src_kind = 2;
  CloudSrc src;
  int st = make_cloud_src(c, xyz, n, stride, 0, on_device, src);
  if (st != HOPE_OK) return st;
  std::vector<Pose> saved;
  saved.swap(c->frame_poses);
  c->csrc = src;
  c->forced_bbox = bbox;
  c->stop_after_voxel = stop_after_voxel;
  c->input_is_downsampled = is_downsampled;
  c->n_raw_override = n_raw_override;
  st = run_pipeline(c, src, n);
  c->forced_bbox = nullptr;
  c->stop_after_voxel = 0;
  c->input_is_downsampled = 0;
  c->n_raw_override = -1;
  c->frame_poses.swap(saved);
  return st;
}

HOPE_API int hope_downsample_cloud_bbox(hope_ctx* c, const void* xyz, size_t n, size_t stride, int on_device,
                                        const float bbox[6])
{
  if (!c || !bbox) return HOPE_E_INVALID;
  return run_cloud_staged(c, xyz, n, stride, on_device, bbox, 1, 0, -1);
}

HOPE_API int hope_segment_downsampled(hope_ctx* c, const void* vox, size_t n_vox, size_t stride, int on_device,
                                      const float bbox[6], int64_t n_raw_total)
{
  if (!c || !bbox) return HOPE_E_INVALID;
  return run_cloud_staged(c, vox, n_vox, stride, on_device, bbox, 0, 1, (long long)n_raw_total);
}

HOPE_API int hope_segment_downsampled_staged(hope_ctx* c, const void* vox, size_t n_vox, size_t stride, int on_device,
                                             const float bbox[6], int64_t n_raw_total, int64_t own_first,
                                             int64_t own_count, const float* normals_all, const int32_t* knn_all,
                                             int stop_after)
{
  if (!c || !bbox) return HOPE_E_INVALID;
  if (own_first < 0 || own_count < 0 || (uint64_t)(own_first + own_count) > n_vox || stop_after < 0 || stop_after > 2 ||
      (knn_all && !normals_all)) {
    c->err = "hope_segment_downsampled_staged: bad argument";
    return HOPE_E_INVALID;
  }
  c->own_lo = (unsigned)own_first;
  c->own_hi = (unsigned)(own_first + own_count);
  c->inj_normals = normals_all;
  c->inj_knn = knn_all;
  c->inj_on_device = on_device;
  c->stop_stage = stop_after;
  const int st = run_cloud_staged(c, vox, n_vox, stride, on_device, bbox, 0, 1, (long long)n_raw_total);
  c->own_lo = 0u;
  c->own_hi = 0xffffffffu;
  c->inj_normals = nullptr;
  c->inj_knn = nullptr;
  c->stop_stage = 0;
  return st;
}

#define GET_FRAME(c, frame)                                   \
  if (!(c)) return HOPE_E_INVALID;                            \
  if ((frame) < 0 || (frame) >= (int)(c)->h_fi.size()) {      \
    (c)->err = "frame index out of range";                    \
    return HOPE_E_RANGE;                                      \
  }                                                           \
  const FrameInfo& F = (c)->h_fi[(frame)];                    \
  (void)F;

HOPE_API int hope_num_frames(const hope_ctx* c)
{
  if (!c) return HOPE_E_INVALID;
  return c->kid_first.empty() ? (int)c->h_fi.size() : c->kid_first.back();
}

HOPE_API int hope_frame_counts(hope_ctx* c, int frame, hope_counts* out)
{
  ROUTE(c, frame, hope_frame_counts(kid__, local__, out));
  GET_FRAME(c, frame);
  if (!out) return HOPE_E_INVALID;
  std::memset(out, 0, sizeof(*out));
  out->n_in = c->src_kind == 1 ? (int64_t)c->dsrc.ppf : (int64_t)c->csrc.n;
  out->n_raw = F.n_raw;
  out->n_vox = F.n_vox;
  out->n_fit = F.n_fit;
  out->n_segments = F.n_segments;
  out->n_clusters = F.n_clusters;
  out->n_planes = c->h_nplanes[frame];
  for (int d = 0; d < 3; ++d) {  // (a pass-through frame has no VoxelGrid grid: div_b / min_b only index columns)
    out->grid[d] = F.overflow ? 0 : F.div_b[d];
    out->min_b[d] = F.overflow ? 0 : F.min_b[d];
  }
  out->vox_passthrough = F.overflow ? 1 : 0;
  out->status = F.status;
  return HOPE_OK;
}

static int copy_out(hope_ctx* c, void* dst, const void* src, size_t bytes)
{
  if (bytes == 0) return HOPE_OK;
  if (!dst || !src) return HOPE_E_INVALID;
  HIPCHK(hipSetDevice(c->cfg.device));
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return HOPE_OK;
}

HOPE_API int hope_get_transformed(hope_ctx* c, int frame, float* xyz, int32_t* src_index)
{
  ROUTE(c, frame, hope_get_transformed(kid__, local__, xyz, src_index));
  GET_FRAME(c, frame);
  if (!xyz || !src_index) return HOPE_E_INVALID;
  HIPCHK(hipSetDevice(c->cfg.device));
  const size_t n = c->src_kind == 1 ? c->dsrc.ppf : c->csrc.n;
  if (n == 0 || F.n_raw == 0) return HOPE_OK;
  if (c->src_kind == 0) {
    c->err = "hope_get_transformed: the last call on this ctx kept no input (staged shard call)";
    return HOPE_E_INVALID;
  }
  const ArenaScope scope(c->arena);  // (scratch of this export only: given back on return)
  hipStream_t s = c->stream;
  uint32_t *flags, *scan;
  float* d_xyz;
  int* d_src;
  ALLOC(flags, uint32_t, n + 1);
  ALLOC(scan, uint32_t, n + 1);
  ALLOC(d_xyz, float, (size_t)F.n_raw * 3);
  ALLOC(d_src, int, F.n_raw);
  if (c->src_kind == 1) {
    hipLaunchKernelGGL((k_valid_flags<DepthSrc>), dim3(nblk(n + 1)), dim3(kBlock), 0, s, c->dsrc, c->d_fi, frame,
                       flags);
    HIPCHK(arena_scan_u32(c, flags, scan, n + 1, s));
    hipLaunchKernelGGL((k_export_transformed<DepthSrc>), dim3(nblk(n)), dim3(kBlock), 0, s, c->dsrc, c->d_fi, frame,
                       scan, d_xyz, d_src);
  } else {
    hipLaunchKernelGGL((k_valid_flags<CloudSrc>), dim3(nblk(n + 1)), dim3(kBlock), 0, s, c->csrc, c->d_fi, frame,
                       flags);
    HIPCHK(arena_scan_u32(c, flags, scan, n + 1, s));
    hipLaunchKernelGGL((k_export_transformed<CloudSrc>), dim3(nblk(n)), dim3(kBlock), 0, s, c->csrc, c->d_fi, frame,
                       scan, d_xyz, d_src);
  }
  int st = copy_out(c, xyz, d_xyz, sizeof(float) * 3 * (size_t)F.n_raw);
  if (st != HOPE_OK) return st;
  return copy_out(c, src_index, d_src, sizeof(int) * (size_t)F.n_raw);
}

// ---- F1: objects on the winning plane ----------------------------------------------------
extern "C++" {
namespace {
struct SizeRec { unsigned size, run; };
bool compare_point_clusters(const SizeRec& a, const SizeRec& b) { return a.size < b.size; }  // pcl comparePointClusters

template <class Src>
int upon_plane(hope_ctx* c, const Src& src, size_t n, int frame, const float2* d_contour, const UponFilter& flt, float tolerance,
               int min_size, int max_size, int32_t* n_clusters, int64_t* n_upon)
{
  hipStream_t s = c->stream;
  PrimAlloc pa{arena_alloc_cb, &c->arena};
  uint32_t *flags, *scan;
  ALLOC(flags, uint32_t, n + 1);
  ALLOC(scan, uint32_t, n + 1);
  hipLaunchKernelGGL((k_upon_flags<Src>), dim3(nblk(n + 1)), dim3(kBlock), 0, s, src, c->d_fi, frame, d_contour, flt, flags);
  HIPCHK(arena_scan_u32(c, flags, scan, n + 1, s));
  uint32_t n_keep = 0;
  HIPCHK(hipMemcpyAsync(&n_keep, scan + n, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (n_upon) *n_upon = n_keep;
  if (n_keep == 0) return HOPE_OK;  // PCLBase::initCompute fails on an empty cloud: no clusters
  float4 *pt, *sorted_pt;
  int *src_index, *parent;
  unsigned* bbox_ord;
  ALLOC(pt, float4, n_keep);
  ALLOC(sorted_pt, float4, n_keep);
  ALLOC(src_index, int, n_keep);
  ALLOC(parent, int, n_keep);
  ALLOC(bbox_ord, unsigned, 6);
  const unsigned init_ord[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  HIPCHK(hipMemcpyAsync(bbox_ord, init_ord, sizeof(init_ord), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL((k_upon_compact<Src>), dim3(nblk(n)), dim3(kBlock), 0, s, src, c->d_fi, frame, flags, scan, pt, src_index, bbox_ord);
  unsigned ord[6];
  HIPCHK(hipMemcpyAsync(ord, bbox_ord, sizeof(ord), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  auto ord2f_h = [](unsigned u) { u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; float f; memcpy(&f, &u, 4); return f; };
  UponGrid g;
  // cells 1/16 wider than the tolerance: two points closer than the tolerance are then at most one cell
  // apart on every axis whatever the rounding of the cell arithmetic
  const double cell = (double)tolerance * 1.0625;
  g.inv_cell = (float)(1.0 / cell);
  unsigned long long dims[3];
  for (int d = 0; d < 3; ++d) {
    g.mn[d] = ord2f_h(ord[d]);
    const double ext = (double)ord2f_h(ord[3 + d]) - (double)g.mn[d];
    dims[d] = (unsigned long long)(ext * (double)g.inv_cell) + 2;
    if (dims[d] > 0x7fffffffull) { c->err = "hope_clusters_upon_plane: tolerance too small for the extent of the cloud"; return HOPE_E_RANGE; }
  }
  g.nx = (unsigned)dims[0]; g.ny = (unsigned)dims[1]; g.nz = (unsigned)dims[2];
  const long double cells = (long double)dims[0] * dims[1] * dims[2];
  if (cells >= 9.0e18L) { c->err = "hope_clusters_upon_plane: cell grid exceeds 63 bits"; return HOPE_E_RANGE; }
  uint64_t *keys, *keys_s;
  uint32_t *vals, *vals_s;
  ALLOC(keys, uint64_t, n_keep);
  ALLOC(keys_s, uint64_t, n_keep);
  ALLOC(vals, uint32_t, n_keep);
  ALLOC(vals_s, uint32_t, n_keep);
  hipLaunchKernelGGL(k_upon_keys, dim3(nblk(n_keep)), dim3(kBlock), 0, s, pt, n_keep, g, keys, vals, parent);
  HIPCHK(sort_pairs_u64(pa, keys, keys_s, vals, vals_s, n_keep, bits_for((unsigned long long)cells), s));
  hipLaunchKernelGGL(k_upon_gather, dim3(nblk(n_keep)), dim3(kBlock), 0, s, pt, vals_s, n_keep, sorted_pt);
  const double tol = (double)tolerance;
  const float r2 = (float)(tol * tol);  // KdTreeFLANN::radiusSearch: static_cast<float>(radius * radius)
  hipLaunchKernelGGL(k_upon_union, dim3(nblk(n_keep)), dim3(kBlock), 0, s, sorted_pt, keys_s, n_keep, g, r2, parent);
  // labels = smallest member index; a stable sort by label keeps members ascending and components in
  // the order PCL's seed loop meets them
  uint32_t *label = vals, *iota = vals_s, *label_s, *order, *uniq, *counts, *d_nruns;
  ALLOC(label_s, uint32_t, n_keep);
  ALLOC(order, uint32_t, n_keep);
  ALLOC(uniq, uint32_t, n_keep);
  ALLOC(counts, uint32_t, n_keep);
  ALLOC(d_nruns, uint32_t, 1);
  hipLaunchKernelGGL(k_upon_labels, dim3(nblk(n_keep)), dim3(kBlock), 0, s, parent, n_keep, label, iota);
  HIPCHK(sort_pairs_u32(pa, label, label_s, iota, order, n_keep, bits_for(n_keep), s));
  HIPCHK(rle_u32(pa, label_s, n_keep, uniq, counts, d_nruns, s));
  ALLOC(c->d_upon_xyz, float, (size_t)n_keep * 3);
  ALLOC(c->d_upon_src, int, n_keep);
  hipLaunchKernelGGL(k_upon_output, dim3(nblk(n_keep)), dim3(kBlock), 0, s, pt, src_index, order, n_keep, c->d_upon_xyz, c->d_upon_src);
  uint32_t n_runs = 0;
  HIPCHK(hipMemcpyAsync(&n_runs, d_nruns, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  std::vector<uint32_t> h_counts(n_runs);
  HIPCHK(hipMemcpyAsync(h_counts.data(), counts, sizeof(uint32_t) * n_runs, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipGetLastError());
  std::vector<unsigned> first(n_runs);
  std::vector<SizeRec> kept;
  unsigned acc = 0;
  for (uint32_t r = 0; r < n_runs; ++r) {
    first[r] = acc;
    acc += h_counts[r];
    if ((long long)h_counts[r] >= min_size && (long long)h_counts[r] <= max_size) kept.push_back({h_counts[r], r});
  }
  // EuclideanClusterExtraction::extract: std::sort(clusters.rbegin(), clusters.rend(), comparePointClusters)
  std::sort(kept.rbegin(), kept.rend(), compare_point_clusters);
  for (const SizeRec& k : kept) c->upon.push_back({first[k.run], k.size});
  *n_clusters = (int32_t)c->upon.size();
  return HOPE_OK;
}
}  // namespace
}  // extern "C++"

HOPE_API int hope_clusters_upon_plane(hope_ctx* c, int frame, const float* contour_xyz, int n_contour, float tolerance, int min_size,
                                      int max_size, int32_t* n_clusters, int64_t* n_upon)
{
  if (c && !c->kid_first.empty()) {
    hope_ctx* kid = nullptr;
    int local = 0;
    if (!route_frame(c, frame, kid, local)) {
      c->err = "frame index out of range";
      return HOPE_E_RANGE;
    }
    c->upon_kid = kid;
    const int st = hope_clusters_upon_plane(kid, local, contour_xyz, n_contour, tolerance, min_size, max_size, n_clusters, n_upon);
    if (st < 0) c->err = kid->err;
    return st;
  }
  GET_FRAME(c, frame);
  if (!contour_xyz || n_contour < 0 || !n_clusters || !(tolerance > 0.f)) return HOPE_E_INVALID;
  HIPCHK(hipSetDevice(c->cfg.device));
  if (c->upon_mark_valid) c->arena.rollback(c->upon_mark);
  else { c->upon_mark = c->arena.mark(); c->upon_mark_valid = true; }
  c->upon.clear();
  *n_clusters = 0;
  if (n_upon) *n_upon = 0;
  const size_t n = c->src_kind == 1 ? c->dsrc.ppf : c->csrc.n;
  if (n == 0 || F.n_raw == 0) return HOPE_OK;
  if (c->src_kind == 0) {
    c->err = "hope_clusters_upon_plane: the last call on this ctx kept no input (staged shard call)";
    return HOPE_E_INVALID;
  }
  // getCloudZInfo (utilities.cpp:751-769) and getStraightRect2D (utilities.cpp:1538-1560) of the contour: host scalars
  std::vector<float2> h_contour(std::max(n_contour, 1));
  float z_max = -1000.0f, mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
  for (int i = 0; i < n_contour; ++i) {
    const float x = contour_xyz[3 * i], y = contour_xyz[3 * i + 1], z = contour_xyz[3 * i + 2];
    h_contour[i] = make_float2(x, y);
    if (std::isfinite(z) && z > z_max) z_max = z;
    if (std::isfinite(x) && std::isfinite(y) && std::isfinite(z)) {
      mnx = std::min(mnx, x); mny = std::min(mny, y); mxx = std::max(mxx, x); mxy = std::max(mxy, y);
    }
  }
  if (n_contour == 0) { mnx = mny = 3.402823466e+38f; mxx = mxy = -3.402823466e+38f; }  // getMinMax3D of an empty cloud
  UponFilter flt;
  flt.z_thr = (double)z_max + 0.01;
  flt.cx = (mxx + mnx) * 0.5f;
  flt.cy = (mxy + mny) * 0.5f;
  flt.half_w = (double)(mxx - mnx) * 0.5;
  flt.half_h = (double)(mxy - mny) * 0.5;
  flt.nc = n_contour;
  float2* d_contour;
  ALLOC(d_contour, float2, h_contour.size());
  HIPCHK(hipMemcpyAsync(d_contour, h_contour.data(), sizeof(float2) * h_contour.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));   // h_contour is pageable and goes out of scope
  if (c->src_kind == 1) return upon_plane(c, c->dsrc, n, frame, d_contour, flt, tolerance, min_size, max_size, n_clusters, n_upon);
  return upon_plane(c, c->csrc, n, frame, d_contour, flt, tolerance, min_size, max_size, n_clusters, n_upon);
}

HOPE_API int hope_get_upon_cluster_sizes(hope_ctx* c, int32_t* sizes, int cap)
{
  if (!c || !sizes) return HOPE_E_INVALID;
  if (c->upon_kid) return hope_get_upon_cluster_sizes(c->upon_kid, sizes, cap);
  int n = 0;
  for (; n < (int)c->upon.size() && n < cap; ++n) sizes[n] = (int32_t)c->upon[n].count;
  return n;
}

HOPE_API int hope_get_upon_cluster(hope_ctx* c, int cluster, float* xyz, int32_t* src_index, int cap)
{
  if (!c || !xyz) return HOPE_E_INVALID;
  if (c->upon_kid) {
    const int st = hope_get_upon_cluster(c->upon_kid, cluster, xyz, src_index, cap);
    if (st < 0) c->err = c->upon_kid->err;
    return st;
  }
  if (cluster < 0 || cluster >= (int)c->upon.size()) { c->err = "cluster index out of range"; return HOPE_E_RANGE; }
  const hope_ctx::UponRange r = c->upon[cluster];
  if (cap < (int)r.count) { c->err = "buffer too small"; return HOPE_E_RANGE; }
  int st = copy_out(c, xyz, c->d_upon_xyz + 3 * (size_t)r.first, sizeof(float) * 3 * (size_t)r.count);
  if (st != HOPE_OK) return st;
  if (src_index) { st = copy_out(c, src_index, c->d_upon_src + r.first, sizeof(int) * (size_t)r.count); if (st != HOPE_OK) return st; }
  return (int)r.count;
}

HOPE_API int hope_get_downsampled(hope_ctx* c, int frame, float* xyz)
{
  ROUTE(c, frame, hope_get_downsampled(kid__, local__, xyz));
  GET_FRAME(c, frame);
  if (F.n_vox == 0) return HOPE_OK;
  if (!xyz) return HOPE_E_INVALID;
  std::vector<float4> tmp(F.n_vox);
  int st = copy_out(c, tmp.data(), c->d_vox + F.vox_start, sizeof(float4) * (size_t)F.n_vox);
  if (st != HOPE_OK) return st;
  for (int i = 0; i < F.n_vox; ++i) {
    xyz[3 * i] = tmp[i].x;
    xyz[3 * i + 1] = tmp[i].y;
    xyz[3 * i + 2] = tmp[i].z;
  }
  return HOPE_OK;
}

HOPE_API int hope_get_normals(hope_ctx* c, int frame, float* nxyzc)
{
  ROUTE(c, frame, hope_get_normals(kid__, local__, nxyzc));
  GET_FRAME(c, frame);
  if (F.n_vox == 0) return HOPE_OK;
  return copy_out(c, nxyzc, c->d_normals + F.vox_start, sizeof(float4) * (size_t)F.n_vox);
}

HOPE_API int hope_get_downsampled_to(hope_ctx* c, int frame, float* xyz, int out_on_device)
{
  ROUTE(c, frame, hope_get_downsampled_to(kid__, local__, xyz, out_on_device));
  if (!out_on_device) return hope_get_downsampled(c, frame, xyz);
  GET_FRAME(c, frame);
  if (F.n_vox == 0) return HOPE_OK;
  if (!xyz) return HOPE_E_INVALID;
  HIPCHK(hipSetDevice(c->cfg.device));
  hipLaunchKernelGGL(k_pack_xyz, dim3(nblk((size_t)F.n_vox)), dim3(kBlock), 0, c->stream, c->d_vox + F.vox_start,
                     (unsigned)F.n_vox, xyz);
  HIPCHK(hipStreamSynchronize(c->stream));
  return HOPE_OK;
}

HOPE_API int hope_get_normals_rows(hope_ctx* c, int frame, int64_t first, int64_t count, float* out, int out_on_device)
{
  ROUTE(c, frame, hope_get_normals_rows(kid__, local__, first, count, out, out_on_device));
  GET_FRAME(c, frame);
  if (first < 0 || count < 0 || first + count > F.n_vox) return HOPE_E_RANGE;
  if (count == 0) return HOPE_OK;
  if (!out || !c->d_normals) return HOPE_E_INVALID;
  HIPCHK(hipSetDevice(c->cfg.device));
  HIPCHK(hipMemcpyAsync(out, c->d_normals + F.vox_start + first, sizeof(float4) * (size_t)count,
                        out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return HOPE_OK;
}

HOPE_API int hope_get_knn_rows(hope_ctx* c, int frame, int64_t first, int64_t count, int32_t* out, int out_on_device)
{
  ROUTE(c, frame, hope_get_knn_rows(kid__, local__, first, count, out, out_on_device));
  GET_FRAME(c, frame);
  if (first < 0 || count < 0 || first + count > F.n_fit) return HOPE_E_RANGE;
  if (count == 0) return HOPE_OK;
  if (!out || !c->d_knn) return HOPE_E_INVALID;
  if (out_on_device && F.fit_start != 0) {
    c->err = "hope_get_knn_rows: device output is for single-frame calls";
    return HOPE_E_INVALID;
  }
  const int K = c->cfg.knn;
  HIPCHK(hipSetDevice(c->cfg.device));
  HIPCHK(hipMemcpyAsync(out, c->d_knn + ((size_t)F.fit_start + first) * K, sizeof(int32_t) * (size_t)count * K,
                        out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (!out_on_device && F.fit_start != 0)
    for (size_t i = 0; i < (size_t)count * K; ++i)
      if (out[i] >= 0) out[i] -= F.fit_start;
  return HOPE_OK;
}

HOPE_API int hope_get_fit_indices(hope_ctx* c, int frame, int32_t* idx)
{
  ROUTE(c, frame, hope_get_fit_indices(kid__, local__, idx));
  GET_FRAME(c, frame);
  if (F.n_fit == 0) return HOPE_OK;
  int st = copy_out(c, idx, c->d_fit_idx + F.fit_start, sizeof(int32_t) * (size_t)F.n_fit);
  if (st != HOPE_OK) return st;
  for (int i = 0; i < F.n_fit; ++i) idx[i] -= F.vox_start;
  return HOPE_OK;
}

HOPE_API int hope_get_knn(hope_ctx* c, int frame, int32_t* nbr)
{
  ROUTE(c, frame, hope_get_knn(kid__, local__, nbr));
  GET_FRAME(c, frame);
  if (F.n_fit == 0) return HOPE_OK;
  const size_t cnt = (size_t)F.n_fit * c->cfg.knn;
  int st = copy_out(c, nbr, c->d_knn + (size_t)F.fit_start * c->cfg.knn, sizeof(int32_t) * cnt);
  if (st != HOPE_OK) return st;
  for (size_t i = 0; i < cnt; ++i)
    if (nbr[i] >= 0) nbr[i] -= F.fit_start;
  return HOPE_OK;
}

HOPE_API int hope_get_labels(hope_ctx* c, int frame, int32_t* labels)
{
  ROUTE(c, frame, hope_get_labels(kid__, local__, labels));
  GET_FRAME(c, frame);
  if (F.n_fit == 0) return HOPE_OK;
  int st = copy_out(c, labels, c->d_cid + F.fit_start, sizeof(int32_t) * (size_t)F.n_fit);
  if (st != HOPE_OK) return st;
  for (int i = 0; i < F.n_fit; ++i)
    if (labels[i] >= 0) labels[i] -= F.cl_start;
  return HOPE_OK;
}

HOPE_API int hope_get_clusters(hope_ctx* c, int frame, hope_cluster* out, int cap)
{
  ROUTE(c, frame, hope_get_clusters(kid__, local__, out, cap));
  GET_FRAME(c, frame);
  if (!out && cap > 0) return HOPE_E_INVALID;
  const int n = std::min(cap, F.n_clusters);
  if (n > 0) std::memcpy(out, c->h_rec.data() + F.cl_start, sizeof(hope_cluster) * (size_t)n);
  return n;
}

HOPE_API int hope_get_cluster_indices(hope_ctx* c, int frame, int cluster, int32_t* idx, int cap)
{
  ROUTE(c, frame, hope_get_cluster_indices(kid__, local__, cluster, idx, cap));
  GET_FRAME(c, frame);
  if (cluster < 0 || cluster >= F.n_clusters) return HOPE_E_RANGE;
  const ClusterRec& r = c->h_rec[F.cl_start + cluster];
  if (r.size > cap || !idx) return HOPE_E_RANGE;
  uint32_t first = 0;
  int st = copy_out(c, &first, c->d_cl_first + F.cl_start + cluster, sizeof(uint32_t));
  if (st != HOPE_OK) return st;
  st = copy_out(c, idx, c->d_members + first, sizeof(int32_t) * (size_t)r.size);
  if (st != HOPE_OK) return st;
  for (int i = 0; i < r.size; ++i) idx[i] -= F.fit_start;
  return r.size;
}

// rows [fit_first, fit_first + fit_count) of the k-NN table: edges ZGrowing would follow (|dz| < th_z, validatePoint,
// z_growing.cpp:335-364) that end outside that range of fit points
__global__ void k_count_band_crossings(const int* knn, const float4* fit_pt, int K, float th_z, unsigned lo, unsigned hi,
                                       unsigned long long* n_cross)
{
  const unsigned g = lo + blockIdx.x * blockDim.x + threadIdx.x;
  unsigned mine = 0u;
  if (g < hi) {
    const float zg = fit_pt[g].z;
    for (int j = 0; j < K; ++j) {
      const int v = knn[(size_t)g * K + j];
      if (v < 0 || ((unsigned)v >= lo && (unsigned)v < hi)) continue;
      if (fabsf(zg - fit_pt[v].z) < th_z) ++mine;
    }
  }
  for (int o = kWave / 2; o > 0; o >>= 1) mine += (unsigned)__shfl_down((int)mine, o);
  if ((threadIdx.x & (kWave - 1)) == 0 && mine) atomicAdd(n_cross, (unsigned long long)mine);
}

HOPE_API int hope_count_band_crossings(hope_ctx* c, int frame, int64_t fit_first, int64_t fit_count, int64_t* n_cross)
{
  ROUTE(c, frame, hope_count_band_crossings(kid__, local__, fit_first, fit_count, n_cross));
  GET_FRAME(c, frame);
  if (!n_cross || fit_first < 0 || fit_count < 0 || fit_first + fit_count > F.n_fit) return HOPE_E_INVALID;
  *n_cross = 0;
  if (fit_count == 0) return HOPE_OK;
  if (!c->d_knn || !c->d_fit_pt) {
    c->err = "hope_count_band_crossings: the ctx holds no k-NN rows";
    return HOPE_E_INVALID;
  }
  if (c->h_fi.size() != 1) {
    c->err = "hope_count_band_crossings: single-cloud calls only";
    return HOPE_E_INVALID;
  }
  HIPCHK(hipSetDevice(c->cfg.device));
  const ArenaScope scope(c->arena);
  unsigned long long* d_n;
  ALLOC(d_n, unsigned long long, 1);
  HIPCHK(hipMemsetAsync(d_n, 0, sizeof(unsigned long long), c->stream));
  hipLaunchKernelGGL(k_count_band_crossings, dim3(nblk((size_t)fit_count)), dim3(kBlock), 0, c->stream, c->d_knn,
                     c->d_fit_pt, c->cfg.knn, c->cfg.th_z, (unsigned)fit_first, (unsigned)(fit_first + fit_count), d_n);
  unsigned long long n = 0;
  const int st = copy_out(c, &n, d_n, sizeof(n));
  if (st != HOPE_OK) return st;
  *n_cross = (int64_t)n;
  return HOPE_OK;
}

// F2: the members of a cluster that can be vertices of the convex hull of its xy projection (hull.hpp)
HOPE_API int hope_get_cluster_hull_candidates(hope_ctx* c, int frame, int cluster, int32_t* pos, int cap)
{
  ROUTE(c, frame, hope_get_cluster_hull_candidates(kid__, local__, cluster, pos, cap));
  GET_FRAME(c, frame);
  if (cluster < 0 || cluster >= F.n_clusters) return HOPE_E_RANGE;
  if (cap < 0 || (cap > 0 && !pos)) return HOPE_E_INVALID;
  const ClusterRec& r = c->h_rec[F.cl_start + cluster];
  if (r.size <= 0) return 0;
  if (!c->d_members || !c->d_fit_pt || !c->d_cl_first) {
    c->err = "hope_get_cluster_hull_candidates: no cluster member lists on this ctx";
    return HOPE_E_INVALID;
  }
  HIPCHK(hipSetDevice(c->cfg.device));
  const ArenaScope scope(c->arena);  // (scratch of this export only: given back on return)
  uint32_t *d_pos, *d_count;
  ALLOC(d_pos, uint32_t, (size_t)r.size);
  ALLOC(d_count, uint32_t, 1);
  hipLaunchKernelGGL(k_hull_candidates, dim3(1), dim3(kHullBlock), 0, c->stream, c->d_members,
                     c->d_cl_first + F.cl_start + cluster, (unsigned)r.size, c->d_fit_pt, d_pos, d_count);
  uint32_t n = 0;
  int st = copy_out(c, &n, d_count, sizeof(uint32_t));
  if (st != HOPE_OK) return st;
  const uint32_t take = std::min<uint32_t>(n, (uint32_t)cap);
  if (take > 0) {
    st = copy_out(c, pos, d_pos, sizeof(int32_t) * (size_t)take);
    if (st != HOPE_OK) return st;
  }
  return (int)n;
}

HOPE_API int hope_get_max_plane(hope_ctx* c, int frame, hope_max_plane* out)
{
  ROUTE(c, frame, hope_get_max_plane(kid__, local__, out));
  GET_FRAME(c, frame);
  if (!out) return HOPE_E_INVALID;
  *out = c->h_max[frame];
  return HOPE_OK;
}

HOPE_API int hope_get_max_plane_members(hope_ctx* c, int frame, int32_t* out, int cap)
{
  ROUTE(c, frame, hope_get_max_plane_members(kid__, local__, out, cap));
  GET_FRAME(c, frame);
  if (!out && cap > 0) return HOPE_E_INVALID;
  const std::vector<int32_t>& mem = c->h_max_members[frame];
  const int n = std::min<int>(cap < 0 ? 0 : cap, (int)mem.size());
  for (int i = 0; i < n; ++i) out[i] = mem[i];
  return (int)mem.size();
}

HOPE_API int hope_profile_enable(hope_ctx* c, int on)
{
  if (!c) return HOPE_E_INVALID;
  c->prof.on = on ? 1 : 0;
  for (int i = 0; i < ST_COUNT; ++i) {
    c->prof.ms[i] = 0;
    c->prof.bytes[i] = 0;
    c->prof.launches[i] = 0;
  }
  for (hope_ctx* k : c->kids) hope_profile_enable(k, on);
  return HOPE_OK;
}

// Instrumentation / test access to exact_sum.hpp: the sequential fp32 sums of n_chains arrays of n host floats
// (chain-major), evaluated the way verify_big.hpp evaluates a cluster's sums.  stats (may be null): blocks the
// wave had to evaluate itself, single float steps inside them, blocks finished by plain adds, super-block records
// applied.
HOPE_API int hope_debug_chain_sums(hope_ctx* c, const float* values, size_t n, int n_chains, float* sums,
                                   uint32_t stats[4])
{
  if (!c || !values || !sums || n_chains < 1 || n_chains > 64 || n == 0 || n > ((size_t)1 << 30)) return HOPE_E_INVALID;
  if (hipSetDevice(c->cfg.device) != hipSuccess) return HOPE_E_HIP;
  ArenaScope scope(c->arena);
  hipStream_t s = c->stream;
  const unsigned n_sb = (unsigned)((n + xs::kSuper - 1) / xs::kSuper);
  const size_t n_el = (size_t)n_sb * xs::kSuper, n_blk = (size_t)n_sb * xs::kSub;
  float *terms, *spred, *d_sums;
  double* bsum;
  xs::RecA *recA, *topA;
  int *recB, *topB;
  xs::Span* span;
  xs::Stats* d_stats;
  ALLOC(terms, float, n_chains * n_el);
  ALLOC(bsum, double, n_chains * n_blk);
  ALLOC(spred, float, n_chains * n_blk);
  ALLOC(recA, xs::RecA, n_chains * n_blk);
  ALLOC(recB, int, n_chains * n_blk);
  ALLOC(topA, xs::RecA, (size_t)n_chains * n_sb);
  ALLOC(topB, int, (size_t)n_chains * n_sb);
  ALLOC(d_sums, float, n_chains);
  ALLOC(span, xs::Span, 1);
  ALLOC(d_stats, xs::Stats, 8);
  const xs::Span h_span{0u, n_sb};
  HIPCHK(hipMemsetAsync(terms, 0, sizeof(float) * n_chains * n_el, s));
  HIPCHK(hipMemsetAsync(d_stats, 0, 8 * sizeof(xs::Stats), s));
  HIPCHK(hipMemcpyAsync(span, &h_span, sizeof(h_span), hipMemcpyHostToDevice, s));
  for (int q = 0; q < n_chains; ++q)
    HIPCHK(hipMemcpyAsync(terms + q * n_el, values + (size_t)q * n, sizeof(float) * n, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(xs::k_xs_blocksums, dim3(n_sb, n_chains), dim3(256), 0, s, terms, n_el, bsum, n_blk);
  hipLaunchKernelGGL(xs::k_xs_scan, dim3(1, n_chains), dim3(256), 0, s, span, bsum, n_blk, spred);
  hipLaunchKernelGGL(xs::k_xs_records, dim3(n_sb, n_chains), dim3(256), 0, s, terms, n_el, spred, n_blk, recA, recB, topA,
                     topB, (size_t)n_sb);
  hipLaunchKernelGGL(xs::k_xs_chain, dim3(1, n_chains), dim3(64), 0, s, span, terms, n_el, recA, recB, n_blk, topA, topB,
                     (size_t)n_sb, d_sums, n_chains, (const int*)nullptr, 0, d_stats);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(sums, d_sums, sizeof(float) * n_chains, hipMemcpyDeviceToHost, s));
  xs::Stats h_stats[8] = {};
  HIPCHK(hipMemcpyAsync(h_stats, d_stats, sizeof(h_stats), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (stats) {
    stats[0] = stats[1] = stats[2] = stats[3] = 0;
    for (const xs::Stats& h : h_stats) {
      stats[0] += h.slow_blocks;
      stats[1] += h.float_steps;
      stats[2] += h.plain_blocks;
      stats[3] += h.top_hits;
    }
  }
  return HOPE_OK;
}
// the same counters for the large clusters of the last segment call (all zero if none took that path)
HOPE_API int hope_debug_xs_stats(hope_ctx* c, uint32_t stats[4])
{
  if (!c || !stats) return HOPE_E_INVALID;
  xs::Stats h[24] = {};
  if (c->d_xs_stats) {
    if (hipSetDevice(c->cfg.device) != hipSuccess) return HOPE_E_HIP;
    HIPCHK(hipMemcpy(h, c->d_xs_stats, sizeof(h), hipMemcpyDeviceToHost));
  }
  stats[0] = stats[1] = stats[2] = stats[3] = 0;
  const bool show = std::getenv("HOPE_DEBUG_XS_PRINT") != nullptr;
  for (int i = 0; i < 24; ++i) {
    stats[0] += h[i].slow_blocks;
    stats[1] += h[i].float_steps;
    stats[2] += h[i].plain_blocks;
    stats[3] += h[i].top_hits;
    if (show && (h[i].slow_blocks || h[i].top_hits))
      std::fprintf(stderr, "xs chain %2d: slow blocks %u, float steps %u, plain blocks %u, records applied %u\n", i,
                   h[i].slow_blocks, h[i].float_steps, h[i].plain_blocks, h[i].top_hits);
  }
  return HOPE_OK;
}

HOPE_API int hope_fast_division_active(const hope_ctx* c)
{
  if (!c) return HOPE_E_INVALID;
  return c->kid_first.empty() ? c->fast_div : c->kids[0]->fast_div;
}

HOPE_API int hope_profile_stage_count(void) { return ST_COUNT; }

HOPE_API int hope_profile_stage(hope_ctx* c, int stage, const char** name, double* total_ms, int64_t* launches)
{
  if (!c || stage < 0 || stage >= ST_COUNT) return HOPE_E_RANGE;
  if (name) *name = kStageNames[stage];
  // (sub-batches: the regions of every kid; they overlap in time, so their sum exceeds the wall time)
  double ms = c->prof.ms[stage];
  long long n = c->prof.launches[stage];
  for (const hope_ctx* k : c->kids) {
    ms += k->prof.ms[stage];
    n += k->prof.launches[stage];
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  return HOPE_OK;
}

HOPE_API int hope_profile_stage_bytes(hope_ctx* c, int stage, double* alg_bytes)
{
  if (!c || stage < 0 || stage >= ST_COUNT || !alg_bytes) return HOPE_E_RANGE;
  *alg_bytes = c->prof.bytes[stage];
  for (const hope_ctx* k : c->kids) *alg_bytes += k->prof.bytes[stage];
  return HOPE_OK;
}

HOPE_API int hope_last_timing(hope_ctx* c, double* whole_ms, double* ref_ms)
{
  if (!c) return HOPE_E_INVALID;
  if (whole_ms) *whole_ms = c->whole_ms;
  if (ref_ms) *ref_ms = c->ref_ms;
  return HOPE_OK;
}

}  // extern "C"
