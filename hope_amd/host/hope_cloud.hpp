// Host-side cloud containers with the memory layout of the PCL types the reference passes around
// (src/lib/utilities.h:84-94: PointCloudMono = pcl::PointCloud<pcl::PointXYZ>, PointCloud =
// pcl::PointCloud<pcl::PointXYZRGB>, CloudN = pcl::PointCloud<pcl::Normal>), so that a caller's
// `cloud->points.data()` can be handed to the C ABI with stride 16 / 32 exactly like the real
// PCL buffers.  PCL itself is not available in this image; these are plain structs.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

namespace hope {

struct alignas(16) PointXYZ {
  float x = 0.f, y = 0.f, z = 0.f, w = 1.f;       // pcl::PointXYZ: float data[4], data[3] = 1
  PointXYZ() = default;
  PointXYZ(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
static_assert(sizeof(PointXYZ) == 16, "pcl::PointXYZ is 16 bytes");

struct alignas(16) PointXYZRGB {
  float x = 0.f, y = 0.f, z = 0.f, w = 1.f;
  uint8_t b = 0, g = 0, r = 0, a = 255;           // pcl::PointXYZRGB: BGRA packed in one float slot
  float pad[3] = {0.f, 0.f, 0.f};
};
static_assert(sizeof(PointXYZRGB) == 32, "pcl::PointXYZRGB is 32 bytes");

struct alignas(16) Normal {
  float normal_x = 0.f, normal_y = 0.f, normal_z = 0.f, w = 0.f;
  float curvature = 0.f;
  float pad[3] = {0.f, 0.f, 0.f};
};
static_assert(sizeof(Normal) == 32, "pcl::Normal is 32 bytes");

struct PointXY { float x = 0.f, y = 0.f; };

template <class P>
struct Cloud {
  typedef std::shared_ptr<Cloud<P>> Ptr;
  typedef std::shared_ptr<const Cloud<P>> ConstPtr;
  std::vector<P> points;
  uint32_t width = 0, height = 0;
  bool is_dense = true;
  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void clear() { points.clear(); width = height = 0; }
  void push_back(const P& p) { points.push_back(p); width = (uint32_t)points.size(); height = 1; }
  void resize(size_t n) { points.resize(n); width = (uint32_t)n; height = 1; }
};

typedef Cloud<PointXYZ> PointCloudMono;
typedef Cloud<PointXYZRGB> PointCloud;
typedef Cloud<Normal> CloudN;

struct PointIndices { std::vector<int> indices; };

}  // namespace hope
