/*
 * hope_gpu.h — C ABI of the MI355X-native HoPE horizontal-plane extraction hot path.
 *
 * The reference (DrawZeroPoint/hope) has no FFI/plugin layer: its boundary is the C++
 * class surface PlaneSegment / PlaneSegmentRT (src/lib/plane_segment.h:114-146,243-280).
 * This header is the seam a maintainer binds *inside* those classes: everything between
 * the input cloud / depth frame and the per-plane results, i.e. the block
 * src/lib/plane_segment.cpp:123-159 (offline) and :755-764 + :791-795 (real-time).
 * INTEGRATION.md shows the adapter code.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types; never throws across the boundary
 *   - return value: 0 ok, >0 soft "nothing found" (the reference's warn+return paths),
 *     <0 hard error (hope_last_error() gives the text)
 *   - one ctx = one HIP stream + one workspace; a ctx is not thread-safe, but any number
 *     of ctxs may be used concurrently (the reference's file-scope thresholds,
 *     plane_segment.cpp:10-24, became ctx fields)
 *   - the ctx stream is a non-blocking stream of its own: it does not wait for work queued on
 *     the caller's streams.  A device input (on_device != 0) must be complete before the call
 *     (synchronise the stream that produces it); every entry point returns with its own work
 *     finished, so outputs may be read at once
 *   - results live inside the ctx until the next hope_segment_* call or hope_destroy()
 *   - "frame" = one independent extraction problem; a depth batch has n_frames of them,
 *     a cloud call has exactly one
 *
 * The CPU oracle (oracle/hope_oracle.cpp, test infrastructure only) exports the same
 * functions under the prefix hope_oracle_.
 */
#ifndef HOPE_GPU_H
#define HOPE_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define HOPE_API __attribute__((visibility("default")))
#else
#define HOPE_API
#endif

typedef struct hope_ctx hope_ctx;

enum {
  HOPE_OK = 0,
  HOPE_EMPTY = 1,           /* down-sampled cloud empty (plane_segment.cpp:145-148,757-760) */
  HOPE_NO_NORMAL_FIT = 2,   /* no point passed the |nz| gate (plane_segment.cpp:698,867-870) */
  HOPE_E_INVALID = -1,      /* bad argument */
  HOPE_E_HIP = -2,          /* HIP runtime error */
  HOPE_E_NOMEM = -3,
  HOPE_E_RANGE = -4,        /* a frame index / capacity is out of range */
  HOPE_E_PEER = -5          /* a collective multi-rank call: another rank failed (hope_amd/host/multi_gpu.hpp) */
};

/* Which class's post-clustering semantics to apply. */
enum {
  HOPE_MODE_OFFLINE = 0,    /* PlaneSegment   (plane_segment.cpp:350-389) */
  HOPE_MODE_RT = 1          /* PlaneSegmentRT (plane_segment.cpp:930-962): height window,
                               aggressive merge, keep largest plane */
};

/* All parameters of the path; replaces the reference's globals and literals.
 * Defaults (hope_config_default) are the values hard-coded upstream. */
typedef struct hope_config {
  float th_xy;            /* xy_resolution, hope_node.cpp:160 / hope_ros.cpp:22        (0.05) */
  float th_z;             /* z_resolution,  hope_node.cpp:161 / hope_ros.cpp:23        (0.02) */
  float min_depth;        /* th_min_depth_, plane_segment.cpp:19                       (0.3)  */
  float max_depth;        /* th_max_depth_, plane_segment.cpp:20                       (8.0)  */
  float fx, fy, cx, cy;   /* TUM intrinsics, get_cloud.cpp:215-217       (517,517,318.6,255.3) */
  float depth_scale;      /* get_cloud.cpp:218                                         (5000) */
  int32_t knn;            /* ZGrowing neighbours, plane_segment.cpp:333                (8)    */
  int32_t min_cluster;    /* plane_segment.cpp:330                                     (3)    */
  int32_t mode;           /* HOPE_MODE_*                                                     */
  float min_height;       /* cfg/hope.cfg:14 min_height_cfg                            (0.8)  */
  float max_height;       /* cfg/hope.cfg:15 max_height_cfg                            (1.5)  */
  int32_t aggressive_merge; /* plane_segment.cpp:719                                   (1)    */
  int32_t device;         /* HIP device ordinal (ignored by the oracle)                (0)    */
  int32_t sub_batches;    /* depth batches of many frames: run as up to this many frame ranges, each on a
                             stream and host thread of its own (0 or 1: never split; no reference counterpart:
                             src/hope_node.cpp:223-241 is one loop; ignored by the oracle)     (0)    */
  int32_t reserved[3];
} hope_config;

/* Per-frame stage sizes. */
typedef struct hope_counts {
  int64_t n_in;           /* pixels / input points                                   */
  int64_t n_raw;          /* after range filter (A1/A2)                              */
  int64_t n_vox;          /* after VoxelGrid (A5)                                    */
  int64_t n_fit;          /* after |nz| > th_norm (A7)                               */
  int32_t n_segments;     /* ZGrowing segments before the size filter (A9)           */
  int32_t n_clusters;     /* segments with size >= min_cluster (A9)                  */
  int32_t n_planes;       /* clusters that passed normalAnalysis (+ RT height gate)  */
  int32_t grid[3];        /* VoxelGrid div_b (A5)                                    */
  int32_t min_b[3];       /* VoxelGrid min_b (A5)                                    */
  int32_t vox_passthrough;/* 1 if PCL's "leaf size too small" rule copied the input
                             (more than INT_MAX cells: n_vox = n_raw, grid = min_b = 0) */
  int32_t status;         /* HOPE_OK / HOPE_EMPTY / HOPE_NO_NORMAL_FIT for this frame */
  int32_t reserved[3];
} hope_counts;

/* One ZGrowing cluster (in the reference's own cluster order = seed order). */
typedef struct hope_cluster {
  int32_t size;           /* number of member points                                 */
  int32_t pass;           /* Utilities::normalAnalysis verdict, utilities.cpp:151-255 */
  float z_mean;           /* getCloudZInfo, utilities.cpp:751-769                    */
  float min_x, min_y, max_x, max_y; /* setFeatures, plane_segment.cpp:408-420        */
  int32_t seed_rank;      /* position of the cluster's seed in the ZGrowing seed order */
} hope_cluster;

/* PlaneSegmentRT winner (plane_segment.cpp:953-960). */
typedef struct hope_max_plane {
  int32_t valid;          /* 0: no plane inside the height window                    */
  int32_t n_points;       /* max_plane_points_num_ (merged count when aggressive)    */
  float z;                /* max_plane_z_                                            */
  int32_t n_members;      /* number of clusters whose points form max_plane_cloud_ (no limit:
                             plane_segment.cpp:943-955 concatenates every merged cluster)      */
  int32_t members[16];    /* the first min(n_members, 16) of them, in max_plane_cloud_ point
                             order; the whole list: hope_get_max_plane_members               */
} hope_max_plane;

/* ---- lifecycle ------------------------------------------------------------------- */
HOPE_API int hope_config_default(hope_config* cfg);
/* replaces PlaneSegment::PlaneSegment / PlaneSegmentRT::PlaneSegmentRT threshold set-up
 * (plane_segment.cpp:46-50,721-725) */
HOPE_API int hope_create(const hope_config* cfg, hope_ctx** out);
HOPE_API void hope_destroy(hope_ctx* ctx);
HOPE_API const char* hope_last_error(const hope_ctx* ctx);

/* ---- pose / dynamic parameters --------------------------------------------------- */
/* replaces PlaneSegment::setT + setQ (plane_segment.cpp:93-107) and the tf lookup result
 * consumed by Transform::doTransform (transform.cpp:72-98); quaternion is NOT normalised,
 * exactly like the reference. Applies to every frame. */
HOPE_API int hope_set_pose(hope_ctx* ctx, const float t_xyz[3], const float q_xyzw[4]);
/* replaces PlaneSegment::setRPY (plane_segment.cpp:86-91) + the roll/pitch/yaw overload of
 * Transform::doTransform (transform.cpp:108-130) that the SYN cloud callback applies (plane_segment.cpp:620):
 * translation (0, 0, 1) m (transform.cpp:4-6), rotation tf2::Quaternion::setRPY in double with y negated
 * (transform.cpp:117-118), narrowed to float. Angles in radians. Applies to every frame, like hope_set_pose. */
HOPE_API int hope_set_pose_rpy(hope_ctx* ctx, float roll, float pitch, float yaw);
/* per-frame poses for a depth batch (the TUM_LIST loop, hope_node.cpp:228-241) */
HOPE_API int hope_set_frame_poses(hope_ctx* ctx, int n_frames, const float* t_xyz,
                                  const float* q_xyzw);
/* replaces PlaneSegmentRT::configCallback (plane_segment.cpp:798-801) */
HOPE_API int hope_set_height_window(hope_ctx* ctx, float min_height, float max_height);

/* ---- the hot path ---------------------------------------------------------------- */
/* Fused A1..A12 on TUM-style depth frames: replaces GetCloud::getColorCloud
 * (get_cloud.cpp:207-245) + PlaneSegment::getHorizontalPlanes (plane_segment.cpp:110-159).
 * depth: n_frames*h*w uint16, row-major; on_device != 0 means a HIP device pointer. */
HOPE_API int hope_segment_depth_u16(hope_ctx* ctx, const uint16_t* depth, int w, int h,
                                    int n_frames, int on_device);
/* A2..A12 on a point cloud: replaces getHorizontalPlanes for POINT_CLOUD input
 * (apply_filter_transform = 0, plane_segment.cpp:118-120) and the RT callback + loop
 * (apply_filter_transform = 1, plane_segment.cpp:749-766,778-796).
 * xyz: n points, each the first 3 floats of a record of stride_bytes bytes
 * (12 = packed, 16 = pcl::PointXYZ, 32 = pcl::PointXYZRGB). */
HOPE_API int hope_segment_cloud_xyz(hope_ctx* ctx, const void* xyz, size_t n,
                                    size_t stride_bytes, int apply_filter_transform,
                                    int on_device);

/* ---- results (host buffers supplied by the caller) -------------------------------- */
HOPE_API int hope_num_frames(const hope_ctx* ctx);
HOPE_API int hope_frame_counts(hope_ctx* ctx, int frame, hope_counts* out);
/* src_mono_cloud_ after A3 (n_raw*3 floats) and the source index of each point */
HOPE_API int hope_get_transformed(hope_ctx* ctx, int frame, float* xyz, int32_t* src_index);
/* src_sp_mono_ / src_dsp_mono_ (n_vox*3 floats, PCL VoxelGrid order) */
HOPE_API int hope_get_downsampled(hope_ctx* ctx, int frame, float* xyz);
/* src_normals_ (n_vox*4 floats: nx,ny,nz,curvature) */
HOPE_API int hope_get_normals(hope_ctx* ctx, int frame, float* nxyzc);
/* idx_norm_fit_ (n_fit int32 indices into the down-sampled cloud) */
HOPE_API int hope_get_fit_indices(hope_ctx* ctx, int frame, int32_t* idx);
/* ZGrowing::point_neighbours_ (n_fit*knn int32, -1 padded when n_fit < knn) */
HOPE_API int hope_get_knn(hope_ctx* ctx, int frame, int32_t* nbr);
/* per fit point: index into the cluster list (seed order, after the size filter), -1 none */
HOPE_API int hope_get_labels(hope_ctx* ctx, int frame, int32_t* labels);
/* seed_clusters_indices_ summary; returns the number of clusters written (<= cap) */
HOPE_API int hope_get_clusters(hope_ctx* ctx, int frame, hope_cluster* out, int cap);
/* seed_clusters_indices_[cluster].indices (ascending, into the fit cloud) */
HOPE_API int hope_get_cluster_indices(hope_ctx* ctx, int frame, int cluster, int32_t* idx,
                                      int cap);
/* F2, the device side of Utilities::computeHull (utilities.cpp:1531-1536, called plane_segment.cpp:957) and
 * PlaneSegment::computeHull (plane_segment.cpp:391-406), which hand every point of a plane to pcl::ConvexHull:
 * the positions (into the list hope_get_cluster_indices returns, ascending) of the members that are NOT strictly
 * inside the octagon of the cluster's eight extreme points. Every vertex of the convex hull of the cluster's xy
 * projection is among them, so the hull of these few points is the hull of the cluster. Returns their number
 * (which may exceed cap; at most cap are written). */
HOPE_API int hope_get_cluster_hull_candidates(hope_ctx* ctx, int frame, int cluster, int32_t* pos_out, int cap);
/* PlaneSegmentRT::max_plane_* */
HOPE_API int hope_get_max_plane(hope_ctx* ctx, int frame, hope_max_plane* out);
/* every cluster whose points form max_plane_cloud_ (plane_segment.cpp:943-955: combineCloud puts the
 * new cluster in front of the old max plane, utilities.cpp:1717-1727), in point order: writes
 * min(cap, n_members) cluster indices and returns n_members */
HOPE_API int hope_get_max_plane_members(hope_ctx* ctx, int frame, int32_t* out, int cap);
/* wall-clock of the last segment call in ms: whole path, and the reference's own timer
 * window (findAllPlanes only, plane_segment.cpp:157-169) */
HOPE_API int hope_last_timing(hope_ctx* ctx, double* whole_ms, double* ref_window_ms);

/* ---- objects on the winning plane ------------------------
 * replaces Utilities::getClustersUponPlane (utilities.cpp:1363-1408, caller PlaneSegmentRT::
 * postProcessing plane_segment.cpp:1064-1070) = getCloudZInfo + getStraightRect2D of the contour,
 * the z / rectangle / isInContour filter (utilities.cpp:1337-1360,1377-1392) over src_mono_cloud_ of
 * `frame` of the last segment call, and Utilities::extractClusters -> pcl::EuclideanClusterExtraction
 * (utilities.cpp:556-571: tolerance 0.01, sizes 10..240000 upstream).  The input of that segment call is
 * read again, so a device input (on_device != 0) must still be alive and unchanged.
 * contour_xyz: n_contour points (max_plane_contour_), 3 floats each, in hull order.
 * Returns HOPE_OK and the number of clusters kept in *n_clusters (0 = the reference's `return false`);
 * clusters are in the reference's order (std::sort by size, descending, as libstdc++ leaves ties).
 * Results stay valid until the next hope_segment_* or hope_clusters_upon_plane call on the ctx. */
HOPE_API int hope_clusters_upon_plane(hope_ctx* ctx, int frame, const float* contour_xyz, int n_contour,
                                      float tolerance, int min_size, int max_size, int32_t* n_clusters,
                                      int64_t* n_upon);
/* sizes of the clusters, in output order; returns how many were written (<= cap) */
HOPE_API int hope_get_upon_cluster_sizes(hope_ctx* ctx, int32_t* sizes, int cap);
/* one cluster: its points (size*3 floats, ascending order of the filtered cloud `temp`) and, optionally,
 * the index of each point in the segment call's input (pixel index for depth frames); src_index may be NULL */
HOPE_API int hope_get_upon_cluster(hope_ctx* ctx, int cluster, float* xyz, int32_t* src_index, int cap);

/* ---- one cloud across several GPUs ---------------------------------------------------
 * No counterpart in the reference (one process, one thread); SURVEY.md section 8e.  One process
 * per GPU holds a contiguous index range of the raw cloud.  The stages below are the device
 * work; the exchanges between them (bbox MIN/MAX and int64 z-histogram all-reduce, one
 * all-to-all of routed points, one all-gather of the down-sampled cloud) are the caller's
 * (hope_amd/sharding.py: torch.distributed over RCCL).  Results are bit-identical to one
 * hope_segment_cloud_xyz call on the whole cloud:
 *   1. hope_shard_bbox          local bounding box + survivor count after A2/A3
 *   2. hope_grid_from_bbox      VoxelGrid min_b / div_b of the all-reduced box (A5's grid)
 *   3. hope_shard_z_histogram   survivors per z layer of that grid (all-reduce: SUM)
 *   4. hope_shard_route         transformed survivors grouped by the rank owning their z layer,
 *                               source order kept (all-to-all; concatenating the received pieces
 *                               in rank order keeps ascending global index = VoxelGrid's sum order)
 *   5. hope_downsample_cloud_bbox  A5 on the received points with the global grid; the band
 *                               owners' outputs concatenated in rank order are the VoxelGrid
 *                               output of the whole cloud (hope_get_downsampled; all-gather)
 *   6. hope_segment_downsampled A6..A12 on the gathered down-sampled cloud (or its staged form
 *                               below, which shares A6 and A8 between the ranks)                 */
HOPE_API int hope_grid_from_bbox(hope_ctx* ctx, const float bbox_min_max[6], int32_t min_b[3],
                                 int32_t div_b[3]);
HOPE_API int hope_shard_bbox(hope_ctx* ctx, const void* xyz, size_t n, size_t stride_bytes,
                             int apply_filter_transform, int on_device, float bbox_min_max[6],
                             int64_t* n_raw);
HOPE_API int hope_shard_z_histogram(hope_ctx* ctx, const void* xyz, size_t n, size_t stride_bytes,
                                    int apply_filter_transform, int on_device,
                                    const float bbox_min_max[6], int64_t* hist, int n_bins);
/* hope_shard_z_histogram for callers that cannot allocate (tight serving loops, harnesses that forbid
 * hipMalloc, code that must not disturb an allocator's state).  No counterpart in the reference.  It is a
 * synchronous call (it ends with a stream synchronise, and with hist_on_device == 0 a copy to host memory),
 * so it cannot itself run inside a stream capture.  Same result, bit for bit;
 * it calls no hipMalloc / hipFree / allocating routine: the cloud must already be in device memory, and the
 * scratch its kernels need (frame record, 64-bit bins merged from the workgroups' LDS-private votes) is a
 * static __device__ array of the library, 66 KiB on every GPU the library runs on, held until the process
 * ends.  Limit: n_bins <= 8192 z layers (HOPE_E_RANGE beyond).  hist: n_bins int64, host or
 * (hist_on_device != 0) device memory.  Every call on a GPU shares that GPU's array; the library holds a
 * per-GPU lock across the call, so calls from several ctxs or threads of ONE process take turns (the call
 * blocks anyway); calls from different processes have their own copies of the array. */
HOPE_API int hope_shard_z_histogram_static(hope_ctx* ctx, const void* xyz_device, size_t n,
                                           size_t stride_bytes, int apply_filter_transform,
                                           const float bbox_min_max[6], int64_t* hist, int n_bins,
                                           int hist_on_device);
/* layer_owner[n_bins]: rank of every z layer; out_xyz: room for n_raw*3 floats; counts[n_ranks] */
HOPE_API int hope_shard_route(hope_ctx* ctx, const void* xyz, size_t n, size_t stride_bytes,
                              int apply_filter_transform, int on_device,
                              const float bbox_min_max[6], const int32_t* layer_owner, int n_bins,
                              int n_ranks, float* out_xyz, int out_on_device, int64_t* counts);
HOPE_API int hope_downsample_cloud_bbox(hope_ctx* ctx, const void* xyz, size_t n,
                                        size_t stride_bytes, int on_device,
                                        const float bbox_min_max[6]);
HOPE_API int hope_segment_downsampled(hope_ctx* ctx, const void* vox_xyz, size_t n_vox,
                                      size_t stride_bytes, int on_device,
                                      const float bbox_min_max[6], int64_t n_raw_total);
/* Step 6 with the two neighbour searches (A6 normals, A8 k-NN: most of its time) shared between
 * the ranks by z band; every rank holds the whole down-sampled cloud, so no halo is needed.  Three
 * calls with the same cloud, the all-gathers between them:
 *   a. normals_all = NULL, knn_all = NULL, stop_after = 1: A6 for the down-sampled points
 *      [own_first, own_first + own_count) only (hope_get_normals: rows of that range are valid);
 *      all-gather the rows in rank order -> normals of every point (n_vox x 4 floats)
 *   b. normals_all given, knn_all = NULL, stop_after = 2: A7 on all points (hope_get_fit_indices, the
 *      same on every rank), A8 for the fit points whose down-sampled index lies in the own range
 *      (hope_get_knn: those rows are valid); all-gather -> k-NN rows of every fit point (n_fit x knn)
 *   c. both given, stop_after = 0: A7, A9..A12; the result is that of hope_segment_downsampled.
 * normals_all / knn_all follow on_device like the cloud. */
/* Rows [first, first + count) of hope_get_normals / hope_get_knn of a frame, into host memory or
 * (out_on_device != 0; k-NN rows: single-frame calls only) straight into device memory: what a rank
 * contributes to the all-gathers of the staged call below. */
/* hope_get_downsampled into host or (out_on_device != 0) device memory: a rank's band of the VoxelGrid
 * output, as it goes into the all-gather */
HOPE_API int hope_get_downsampled_to(hope_ctx* ctx, int frame, float* xyz, int out_on_device);
HOPE_API int hope_get_normals_rows(hope_ctx* ctx, int frame, int64_t first, int64_t count,
                                   float* nxyzc, int out_on_device);
HOPE_API int hope_get_knn_rows(hope_ctx* ctx, int frame, int64_t first, int64_t count,
                               int32_t* nbr, int out_on_device);
/* After call b of the staged form: the number of edges ZGrowing would follow (neighbour j of fit point i with
 * |z_i - z_j| < th_z: validatePoint, z_growing.cpp:335-364) from the fit points [fit_first, fit_first + fit_count)
 * -- whose k-NN rows the ctx holds -- to fit points outside that range.  Zero on every rank means the z bands
 * are independent extraction problems: call c may then be given the own rows only (all others -1), and the
 * records of the own band are exactly those of the whole cloud (seed ranks are global).  No counterpart in the
 * reference. */
HOPE_API int hope_count_band_crossings(hope_ctx* ctx, int frame, int64_t fit_first, int64_t fit_count,
                                       int64_t* n_cross);
HOPE_API int hope_segment_downsampled_staged(hope_ctx* ctx, const void* vox_xyz, size_t n_vox,
                                             size_t stride_bytes, int on_device,
                                             const float bbox_min_max[6], int64_t n_raw_total,
                                             int64_t own_first, int64_t own_count,
                                             const float* normals_all, const int32_t* knn_all,
                                             int stop_after);

/* ---- instrumentation (HIP library only; no counterpart in the reference) ------------- */
/* Per-stage device time measured with HIP events on the ctx stream, accumulated over the
 * segment calls made while enabled.  hope_profile_enable also clears the accumulators. */
HOPE_API int hope_profile_enable(hope_ctx* ctx, int on);
HOPE_API int hope_profile_stage_count(void);
HOPE_API int hope_profile_stage(hope_ctx* ctx, int stage, const char** name, double* total_ms,
                                int64_t* launches);
/* ALGORITHMIC bytes (the bytes the stage must read and write once, from
 * the sizes of the calls made while enabled) moved by the same timed regions */
HOPE_API int hope_profile_stage_bytes(hope_ctx* ctx, int stage, double* alg_bytes);

/* The sequential fp32 sums s = fl(fl(a0 + a1) + a2 ...) of n_chains arrays of n host floats (values: chain-major),
 * evaluated the way the library evaluates the reference's sequential sums over a large cluster (getCloudZInfo
 * utilities.cpp:751-769, calNormalMean :257-282, normalAnalysis :151-255): in parallel, in integer arithmetic
 * per binade, bit-identical to the chain of float adds (hope_amd/csrc/exact_sum.hpp).  Test access; stats (may be
 * NULL): {blocks of 64 the wave evaluated itself, single float adds inside them, blocks finished by plain
 * sequential adds, 1024-member records applied}. */
HOPE_API int hope_debug_chain_sums(hope_ctx* ctx, const float* values, size_t n, int n_chains, float* sums,
                                   uint32_t stats[4]);
/* the same counters for the large clusters of the last segment call */
HOPE_API int hope_debug_xs_stats(hope_ctx* ctx, uint32_t stats[4]);

/* 1 if the last depth call used the 3-instruction division by the camera constants (it is
 * enabled only after an exhaustive on-device comparison with the IEEE quotient for every
 * dividend the camera model can produce), 0 if it used IEEE divisions */
HOPE_API int hope_fast_division_active(const hope_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* HOPE_GPU_H */
