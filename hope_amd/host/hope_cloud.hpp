// Host-side cloud containers with the memory layout of the PCL types the reference passes around
// (src/lib/utilities.h:84-94: PointCloudMono = pcl::PointCloud<pcl::PointXYZ>, PointCloud =
// pcl::PointCloud<pcl::PointXYZRGB>, CloudN = pcl::PointCloud<pcl::Normal>), so that a caller's
// `cloud->points.data()` can be handed to the C ABI with stride 16 / 32 exactly like the real
// PCL buffers.  PCL itself is not available in this image; these are plain structs.
//
// With PCL installed (a HoPE maintainer's machine: `find_package(PCL 1.7 REQUIRED)`, CMakeLists.txt:32) the same names
// are the real PCL types: define HOPE_WITH_PCL (or let __has_include find <pcl/point_cloud.h>; HOPE_NO_PCL keeps the
// shims) and PlaneSegment / PlaneSegmentRT of plane_segment.hpp take and return pcl::PointCloud<...>::Ptr exactly as
// src/lib/plane_segment.h:114-146,243-280 declares them.  The adapter code uses only what both have in common:
// `points`, size(), empty(), clear(), push_back(), resize(), width / height / is_dense, Ptr(new T) and the x, y, z
// (r, g, b) members.  That branch cannot be compiled in this image (no PCL); the shim branch is what the tests build.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#if !defined(HOPE_NO_PCL) && !defined(HOPE_WITH_PCL) && defined(__has_include)
#if __has_include(<pcl/point_cloud.h>) && __has_include(<pcl/point_types.h>)
#define HOPE_WITH_PCL 1
#endif
#endif

#ifdef HOPE_WITH_PCL
#include <pcl/PointIndices.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

namespace hope {

typedef pcl::PointXYZ PointXYZ;        // 16 bytes: handed to the C ABI with stride 16
typedef pcl::PointXYZRGB PointXYZRGB;  // 32 bytes: stride 32
typedef pcl::Normal Normal;
typedef pcl::PointXY PointXY;
static_assert(sizeof(PointXYZ) == 16 && sizeof(PointXYZRGB) == 32 && sizeof(Normal) == 32, "PCL point layouts");
typedef pcl::PointCloud<pcl::PointXYZ> PointCloudMono;   // src/lib/utilities.h:84-94
typedef pcl::PointCloud<pcl::PointXYZRGB> PointCloud;
typedef pcl::PointCloud<pcl::Normal> CloudN;
typedef pcl::PointIndices PointIndices;

}  // namespace hope

#else

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
  typedef P PointType;
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

#endif  // HOPE_WITH_PCL
