#include "plane_segment.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <iostream>
#include <stdexcept>

#include "hull2d.hpp"

namespace hope {
namespace {

hope_ctx* make_ctx(float th_xy, float th_z, int mode, int aggressive)
{
  hope_config cfg;
  hope_config_default(&cfg);
  cfg.th_xy = th_xy; cfg.th_z = th_z; cfg.mode = mode; cfg.aggressive_merge = aggressive;
  hope_ctx* ctx = nullptr;
  const int st = hope_create(&cfg, &ctx);
  if (st != HOPE_OK || !ctx) throw std::runtime_error("hope_create failed (no usable MI355X device or library not built?)");
  return ctx;
}

// cluster `c` of `frame` as a cloud: fit cloud gathered by seed_clusters_indices_[c] (plane_segment.cpp:355-360)
struct FrameArrays {
  std::vector<float> vox;         // n_vox * 3
  std::vector<int32_t> fit;       // n_fit
  std::vector<hope_cluster> clusters;
  bool load(hope_ctx* ctx, int frame, const hope_counts& c)
  {
    vox.resize((size_t)c.n_vox * 3); fit.resize((size_t)c.n_fit); clusters.resize((size_t)c.n_clusters);
    if (c.n_vox && hope_get_downsampled(ctx, frame, vox.data()) < 0) return false;
    if (c.n_fit && hope_get_fit_indices(ctx, frame, fit.data()) < 0) return false;
    if (c.n_clusters && hope_get_clusters(ctx, frame, clusters.data(), c.n_clusters) < 0) return false;
    return true;
  }
  bool cluster_cloud(hope_ctx* ctx, int frame, int c, PointCloudMono& out, bool append) const
  {
    std::vector<int32_t> idx((size_t)clusters[c].size);
    if (hope_get_cluster_indices(ctx, frame, c, idx.data(), (int)idx.size()) < 0) return false;
    if (!append) out.points.clear();
    for (int32_t i : idx) { const float* p = &vox[(size_t)fit[i] * 3]; out.points.push_back(PointXYZ(p[0], p[1], p[2])); }
    out.width = (uint32_t)out.points.size(); out.height = 1; out.is_dense = true;
    return true;
  }
  // F2: the few members of cluster c that can be hull vertices (reduced on the device), as indices into a cloud in
  // which the cluster's points start at `offset`
  bool hull_candidates(hope_ctx* ctx, int frame, int c, int offset, std::vector<int>& out) const
  {
    std::vector<int32_t> pos((size_t)clusters[c].size);
    const int n = hope_get_cluster_hull_candidates(ctx, frame, c, pos.data(), (int)pos.size());
    if (n < 0 || n > (int)pos.size()) return false;
    for (int i = 0; i < n; ++i) out.push_back(offset + pos[i]);
    return true;
  }
};

}  // namespace

PlaneSegment::PlaneSegment(data_type mode, float th_xy, float th_z, std::string base_frame, const std::string&)
    : type_(mode), base_frame_(std::move(base_frame))
{
  ctx_ = make_ctx(th_xy, th_z, HOPE_MODE_OFFLINE, 1);
}

PlaneSegment::~PlaneSegment() { hope_destroy(ctx_); }

void PlaneSegment::setRPY(float roll, float pitch, float yaw) { roll_ = roll; pitch_ = pitch; yaw_ = yaw; }
void PlaneSegment::setQ(float qx, float qy, float qz, float qw) { qx_ = qx; qy_ = qy; qz_ = qz; qw_ = qw; }
void PlaneSegment::setT(float tx, float ty, float tz) { tx_ = tx; ty_ = ty; tz_ = tz; }

void PlaneSegment::reset()   // plane_segment.cpp:662-678
{
  plane_results_.clear(); plane_points_.clear(); plane_hull_.clear(); plane_area_.clear(); plane_coeff_.clear();
  plane_max_result_.reset(); plane_max_points_.reset(); plane_max_hull_.reset(); plane_max_coeff_.clear();
}

void PlaneSegment::afterCall(int st)
{
  last_status_ = st;
  hope_last_timing(ctx_, &whole_ms_, &ref_ms_);
  if (st < 0) { std::cerr << "PlaneSegment: " << hope_last_error(ctx_) << std::endl; return; }
  collect(0);
}

bool PlaneSegment::collect(int frame)
{
  if (hope_frame_counts(ctx_, frame, &counts_) < 0) { std::cerr << "PlaneSegment: " << hope_last_error(ctx_) << std::endl; return false; }
  last_status_ = counts_.status;
  std::cout << "Point number after down sampling: #" << counts_.n_vox << std::endl;   // plane_segment.cpp:143
  if (counts_.status == HOPE_EMPTY) {
    std::cerr << "PlaneSegment: Source cloud is empty." << std::endl;                  // :145-148, results left as they were
    return false;
  }
  reset();
  if (counts_.status == HOPE_OK) {
    FrameArrays fa;
    if (!fa.load(ctx_, frame, counts_)) { std::cerr << "PlaneSegment: " << hope_last_error(ctx_) << std::endl; return false; }
    size_t best = 0;
    for (int c = 0; c < counts_.n_clusters; ++c) {
      const hope_cluster& cl = fa.clusters[c];
      if (!cl.pass) continue;                                                         // gaussianImageAnalysis, :362
      PointCloudMono::Ptr pts(new PointCloudMono);
      if (!fa.cluster_cloud(ctx_, frame, c, *pts, false)) return false;
      plane_points_.push_back(pts);                                                   // :373
      plane_results_.push_back(PointCloud::Ptr(new PointCloud));                      // :372, never filled upstream
      plane_coeff_.push_back({cl.z_mean, cl.min_x, cl.min_y, cl.max_x, cl.max_y});    // :408-420
      if (cal_hull_) {                                                                // :376-378 (upstream hulls the empty rgb cloud)
        PointCloudMono hull;
        std::vector<int> cand;
        if (!fa.hull_candidates(ctx_, frame, c, 0, cand)) return false;
        plane_area_.push_back((float)computeHull(*pts, hull, nullptr, &cand));        // :391-406 on the device-reduced points
        PointCloud::Ptr hull_rgb(new PointCloud);
        for (const PointXYZ& p : hull.points) { PointXYZRGB q; q.x = p.x; q.y = p.y; q.z = p.z; hull_rgb->push_back(q); }
        plane_hull_.push_back(hull_rgb);
      }
      if (pts->size() > best) {                                                       // the largest plane (commented out upstream, :380-388)
        best = pts->size();
        plane_max_points_ = pts; plane_max_result_ = plane_results_.back(); plane_max_coeff_ = plane_coeff_.back();
        if (cal_hull_) plane_max_hull_ = plane_hull_.back();
      }
    }
  }
  ids_.setID(plane_coeff_);                                                           // :171
  return true;
}

void PlaneSegment::getHorizontalPlanes(PointCloud::Ptr cloud)
{
  if (type_ == SYN) {
    // plane_segment.cpp:113-117 spins until the subscriber's cloudCallback (:604-622) has delivered a cloud; there is
    // no ROS here, so the caller delivers it through cloudCallback and this call consumes it.
    if (!syn_cloud_) {
      std::cerr << "PlaneSegment: SYN mode has no source cloud yet (cloudCallback has not run)." << std::endl;
      last_status_ = HOPE_E_INVALID;
      return;
    }
    PointCloud::ConstPtr src = syn_cloud_;
    syn_cloud_.reset();
    hope_set_pose_rpy(ctx_, roll_, pitch_, yaw_);                                                                // :620
    afterCall(hope_segment_cloud_xyz(ctx_, src->points.data(), src->points.size(), sizeof(PointXYZRGB), 1, 0)); // :617-621
    return;
  }
  if (!cloud) { last_status_ = HOPE_E_INVALID; return; }
  int st;
  if (type_ == POINT_CLOUD) {
    st = hope_segment_cloud_xyz(ctx_, cloud->points.data(), cloud->points.size(), sizeof(PointXYZRGB), 0, 0);   // :118-120
  } else {
    const float t[3] = {tx_, ty_, tz_}, q[4] = {qx_, qy_, qz_, qw_};
    hope_set_pose(ctx_, t, q);                                                                                   // :127
    st = hope_segment_cloud_xyz(ctx_, cloud->points.data(), cloud->points.size(), sizeof(PointXYZRGB), 1, 0);   // :123-127
  }
  afterCall(st);
}

void PlaneSegment::cloudCallback(const PointCloud::ConstPtr& src_temp)
{
  if (!src_temp || src_temp->empty()) { std::cerr << "HoPE: PointCloud is empty." << std::endl; return; }       // :606-609
  syn_cloud_ = src_temp;
}

void PlaneSegment::getHorizontalPlanes(const uint16_t* depth, int width, int height)
{
  const float t[3] = {tx_, ty_, tz_}, q[4] = {qx_, qy_, qz_, qw_};
  hope_set_pose(ctx_, t, q);
  afterCall(hope_segment_depth_u16(ctx_, depth, width, height, 1, 0));
}

bool PlaneSegment::runBatch(const uint16_t* depth, int w, int h, int n, const float* t, const float* q)
{
  int st = hope_set_frame_poses(ctx_, n, t, q);
  if (st >= 0) st = hope_segment_depth_u16(ctx_, depth, w, h, n, 0);
  last_status_ = st;
  hope_last_timing(ctx_, &whole_ms_, &ref_ms_);
  if (st < 0) { std::cerr << "PlaneSegment: " << hope_last_error(ctx_) << std::endl; return false; }
  return true;
}

// ------------------------------------------------------------------------------------------- RT
PlaneSegmentRT::PlaneSegmentRT(float th_xy, float th_z, std::string base_frame, const std::string&)
    : aggressive_merge_(true), max_plane_cloud_(new PointCloudMono), max_plane_contour_(new PointCloudMono), max_plane_z_(-1000.0f),
      base_frame_(std::move(base_frame)), th_xy_(th_xy), th_z_(th_z)
{
  ctx_ = make_ctx(th_xy, th_z, HOPE_MODE_RT, 1);
  hope_set_height_window(ctx_, min_height_, max_height_);
}

PlaneSegmentRT::~PlaneSegmentRT() { hope_destroy(ctx_); }

void PlaneSegmentRT::reset()   // plane_segment.cpp:882-892
{
  max_plane_cloud_.reset(new PointCloudMono);
  max_plane_contour_.reset(new PointCloudMono);
  max_plane_area_ = 0.0;
}

void PlaneSegmentRT::configCallback(float min_height_cfg, float max_height_cfg)
{
  min_height_ = min_height_cfg; max_height_ = max_height_cfg;
  hope_set_height_window(ctx_, min_height_, max_height_);
}

void PlaneSegmentRT::cloudCallback(const PointCloudMono::ConstPtr& src_temp, const double t_xyz[3], const double q_xyzw[4])
{
  if (!src_temp || src_temp->empty()) { std::cerr << "HoPE: Source point cloud is empty." << std::endl; return; }   // :780-783
  for (int i = 0; i < 3; ++i) t_[i] = (float)t_xyz[i];
  for (int i = 0; i < 4; ++i) q_[i] = (float)q_xyzw[i];
  hope_set_pose(ctx_, t_, q_);
  src_temp_ = src_temp;
  have_cloud_ = true;
}

void PlaneSegmentRT::getHorizontalPlanes()
{
  if (!have_cloud_) { std::cerr << "HoPE: no source cloud yet (cloudCallback has not run)." << std::endl; return; }   // getSourceCloud spins upstream
  have_cloud_ = false;
  // aggressive_merge_ is a public member upstream and may be flipped between calls; in the C ABI it is part
  // of the ctx configuration, so a change means a new ctx.
  if (aggressive_merge_ != ctx_aggressive_) {
    hope_destroy(ctx_);
    ctx_ = make_ctx(th_xy_, th_z_, HOPE_MODE_RT, aggressive_merge_ ? 1 : 0);
    ctx_aggressive_ = aggressive_merge_;
    hope_set_height_window(ctx_, min_height_, max_height_);
    hope_set_pose(ctx_, t_, q_);
  }
  const int st = hope_segment_cloud_xyz(ctx_, src_temp_->points.data(), src_temp_->points.size(), sizeof(PointXYZ), 1, 0);
  last_status_ = st;
  if (st < 0) { std::cerr << "HoPE: " << hope_last_error(ctx_) << std::endl; return; }
  if (st == HOPE_EMPTY) { std::cerr << "HoPE: Down sampled source cloud is empty." << std::endl; return; }   // :757-760
  reset();                                                                                                   // :762
  if (st == HOPE_NO_NORMAL_FIT) std::cerr << "HoPE: No point fits the normal criteria" << std::endl;         // :867-870
  hope_counts c;
  hope_max_plane mp;
  if (st == HOPE_OK && hope_frame_counts(ctx_, 0, &c) >= 0 && hope_get_max_plane(ctx_, 0, &mp) >= 0 && mp.valid) {
    FrameArrays fa;
    if (!fa.load(ctx_, 0, c)) return;
    std::vector<int32_t> members((size_t)std::max(mp.n_members, 1));   // (every merged cluster: upstream has no limit)
    if (hope_get_max_plane_members(ctx_, 0, members.data(), mp.n_members) != mp.n_members) return;
    std::vector<int> cand;                                          // a vertex of the merged hull is one of its own cluster's hull
    for (int m = 0; m < mp.n_members; ++m) {                        // combineCloud order, :943-955 / utilities.cpp:1717-1727
      const int offset = (int)max_plane_cloud_->size();
      if (!fa.cluster_cloud(ctx_, 0, members[m], *max_plane_cloud_, true)) return;
      if (!fa.hull_candidates(ctx_, 0, members[m], offset, cand)) return;
    }
    max_plane_z_ = mp.z;                                            // :958
    max_plane_area_ = computeHull(*max_plane_cloud_, *max_plane_contour_, nullptr, &cand);   // :957, on the device-reduced points
  }
  if (!max_plane_cloud_->empty())
    std::printf("Max plane z=%.3f detected in height range %.3f, %.3f\n", max_plane_z_, min_height_, max_height_);   // :983-989
  else
    std::printf("No plane detected in height range %.3f, %.3f\n", min_height_, max_height_);
}

bool PlaneSegmentRT::getClustersUponPlane(std::vector<PointCloudMono::Ptr>& clusters)
{
  if (max_plane_contour_->empty()) {
    std::cerr << "HoPE: No valid plane for extracting objects on top." << std::endl;   // plane_segment.cpp:1126-1129
    return false;
  }
  std::vector<float> contour;
  for (const PointXYZ& p : max_plane_contour_->points) { contour.push_back(p.x); contour.push_back(p.y); contour.push_back(p.z); }
  int32_t n = 0;
  const int st = hope_clusters_upon_plane(ctx_, 0, contour.data(), (int)max_plane_contour_->size(), 0.01f, 10, 240000, &n, nullptr);   // utilities.cpp:1397
  if (st < 0) { std::cerr << "HoPE: " << hope_last_error(ctx_) << std::endl; return false; }
  std::vector<int32_t> sizes((size_t)std::max(n, 1));
  hope_get_upon_cluster_sizes(ctx_, sizes.data(), n);
  for (int k = 0; k < n; ++k) {
    std::vector<float> xyz((size_t)sizes[k] * 3);
    if (hope_get_upon_cluster(ctx_, k, xyz.data(), nullptr, sizes[k]) < 0) return false;
    PointCloudMono::Ptr cl(new PointCloudMono);
    for (int i = 0; i < sizes[k]; ++i) cl->push_back(PointXYZ(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
    clusters.push_back(cl);
  }
  std::printf("HoPE: Object clusters on plane #: %d\n", (int)clusters.size());   // plane_segment.cpp:1069
  return !clusters.empty();
}

}  // namespace hope
