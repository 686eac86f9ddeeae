// Host side of the drop-in: the reference's class surface (src/lib/plane_segment.h:89,114-146,243-280)
// over the C ABI of include/hope_gpu.h.  Same class names, constructor arguments, setters, public result
// members and warn-and-return behaviour; ROS, tf, the PCL visualiser and the dynamic_reconfigure server are
// not here (absent from this image and outside SURVEY.md section 8), so the two ROS callbacks of
// PlaneSegmentRT are ordinary methods the caller invokes with the message content.
#pragma once
#include <string>
#include <vector>

#include "hope_cloud.hpp"
#include "hope_gpu.h"
#include "plane_id.hpp"

namespace hope {

enum data_type { SYN, POINT_CLOUD, TUM_SINGLE, TUM_LIST };   // plane_segment.h:89

class PlaneSegment {
 public:
  // plane_segment.h:125 / plane_segment.cpp:27-63.  base_frame and cloud_topic are accepted and ignored
  // (SYN mode's ROS source is not available).
  PlaneSegment(data_type mode, float th_xy, float th_z, std::string base_frame = "", const std::string& cloud_topic = "");
  ~PlaneSegment();
  PlaneSegment(const PlaneSegment&) = delete;
  PlaneSegment& operator=(const PlaneSegment&) = delete;

  // plane_segment.cpp:110-173
  void getHorizontalPlanes(PointCloud::Ptr cloud);
  // The same for a TUM depth image without building the host cloud first (GetCloud::getColorCloud,
  // get_cloud.cpp:207-245, runs inside the library); results land in the same members.
  void getHorizontalPlanes(const uint16_t* depth, int width, int height);
  // TUM_LIST as one batched device call (hope_node.cpp:228-241): n frames, per-frame pose; `each` is called
  // once per frame, in order, after that frame's results have been placed in the public members.
  template <class F>
  void getHorizontalPlanesBatch(const uint16_t* depth, int width, int height, int n_frames, const float* t_xyz, const float* q_xyzw, F each)
  {
    if (!runBatch(depth, width, height, n_frames, t_xyz, q_xyzw)) return;
    for (int f = 0; f < n_frames; ++f) { if (collect(f)) each(f); }
  }

  /// Container for storing final results (plane_segment.h:129-142)
  std::vector<PointCloud::Ptr> plane_results_;      // empty clouds, like upstream (plane_segment.cpp:367,372)
  std::vector<PointCloudMono::Ptr> plane_points_;
  std::vector<PointCloud::Ptr> plane_hull_;         // filled when cal_hull_ (hull of plane_points_[i])
  std::vector<float> plane_area_;                   // qhull totvol of that hull
  std::vector<std::vector<float>> plane_coeff_;
  PointCloud::Ptr plane_max_result_;
  PointCloudMono::Ptr plane_max_points_;
  PointCloud::Ptr plane_max_hull_;
  std::vector<float> plane_max_coeff_;

  // SYN mode's cloud source (plane_segment.cpp:604-622): upstream a ROS subscriber hands the message to
  // cloudCallback; here the caller does.  The next getHorizontalPlanes(any) range-filters the sensor-frame
  // cloud and transforms it with the setRPY angles (transform.cpp:108-150, hope_set_pose_rpy).
  void cloudCallback(const PointCloud::ConstPtr& src_temp);

  void setRPY(float roll = 0.0, float pitch = 0.0, float yaw = 0.0);
  void setQ(float qx = 0.0, float qy = 0.0, float qz = 0.0, float qw = 1.0);
  void setT(float tx = 0.0, float ty = 0.0, float tz = 0.0);

  bool cal_hull_ = false;                                        // file-scope flag upstream, plane_segment.cpp:22
  const std::vector<int>& plane_ids() const { return ids_.global_id_temp_; }   // setID result, plane_segment.cpp:496-517
  int last_status() const { return last_status_; }               // HOPE_OK / HOPE_EMPTY / HOPE_NO_NORMAL_FIT / <0
  const hope_counts& last_counts() const { return counts_; }
  double last_ms() const { return whole_ms_; }
  double last_timer_window_ms() const { return ref_ms_; }        // the reference's hst_ window (findAllPlanes)
  hope_ctx* ctx() { return ctx_; }

 protected:
  data_type type_;

 private:
  float roll_ = 0.f, pitch_ = 0.f, yaw_ = 0.f;
  float qx_ = 0.f, qy_ = 0.f, qz_ = 0.f, qw_ = 1.f;
  float tx_ = 0.f, ty_ = 0.f, tz_ = 0.f;
  std::string base_frame_;
  PointCloud::ConstPtr syn_cloud_;          // SYN: the cloud of the last cloudCallback, consumed by the next call
  PlaneIdTracker ids_;
  hope_ctx* ctx_ = nullptr;
  hope_counts counts_{};
  int last_status_ = 0;
  double whole_ms_ = 0, ref_ms_ = 0;

  void reset();
  bool collect(int frame);
  bool runBatch(const uint16_t* depth, int w, int h, int n, const float* t, const float* q);
  void afterCall(int st);
};

class PlaneSegmentRT {
 public:
  // plane_segment.h:255 / plane_segment.cpp:704-747 without the ros::NodeHandle
  PlaneSegmentRT(float th_xy, float th_z, std::string base_frame = "", const std::string& cloud_topic = "");
  ~PlaneSegmentRT();
  PlaneSegmentRT(const PlaneSegmentRT&) = delete;
  PlaneSegmentRT& operator=(const PlaneSegmentRT&) = delete;

  bool aggressive_merge_;
  void getHorizontalPlanes();                                     // plane_segment.cpp:749-766

  PointCloudMono::Ptr max_plane_cloud_;
  PointCloudMono::Ptr max_plane_contour_;
  float max_plane_z_;
  double max_plane_area_ = 0.0;

  // cloudCallback (plane_segment.cpp:778-796): the sensor-frame cloud of the message and the tf lookup
  // result base_frame <- msg frame (translation, quaternion x y z w; doubles narrowed like transform.cpp:75-81).
  void cloudCallback(const PointCloudMono::ConstPtr& src_temp, const double t_xyz[3], const double q_xyzw[4]);
  // configCallback (plane_segment.cpp:798-801)
  void configCallback(float min_height_cfg, float max_height_cfg);

  // The clustering branch of postProcessing (plane_segment.cpp:1064-1070): Utilities::getClustersUponPlane
  // (utilities.cpp:1363-1408) on src_mono_cloud_ of the last getHorizontalPlanes() with max_plane_contour_.
  // Returns false when there is no valid contour or no cluster; pose fitting of the clusters is not part
  // of this library.
  bool getClustersUponPlane(std::vector<PointCloudMono::Ptr>& clusters);

  int last_status() const { return last_status_; }
  hope_ctx* ctx() { return ctx_; }

 private:
  std::string base_frame_;
  float min_height_ = 0.8f, max_height_ = 1.5f;                   // cfg/hope.cfg:14-15
  PointCloudMono::ConstPtr src_temp_;
  bool have_cloud_ = false;
  float th_xy_, th_z_;
  float t_[3] = {0.f, 0.f, 0.f}, q_[4] = {0.f, 0.f, 0.f, 1.f};
  bool ctx_aggressive_ = true;
  hope_ctx* ctx_ = nullptr;
  int last_status_ = 0;
  void reset();
};

}  // namespace hope
