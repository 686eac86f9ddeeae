// Drives hope::PlaneSegmentRT like hope_ros.cpp:36-40 does, from a file of packed xyz floats:
//   rt_driver cloud.f32 tx ty tz qx qy qz qw min_height max_height aggressive
// prints one JSON line with the winner (z, points, hull indices' coordinates, area).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "plane_segment.hpp"

int main(int argc, char** argv)
{
  if (argc != 12) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  std::fseek(f, 0, SEEK_END);
  const long bytes = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<float> xyz(bytes / 4);
  if (std::fread(xyz.data(), 4, xyz.size(), f) != xyz.size()) return 2;
  std::fclose(f);
  hope::PointCloudMono::Ptr cloud(new hope::PointCloudMono);
  for (size_t i = 0; i + 2 < xyz.size(); i += 3) cloud->push_back(hope::PointXYZ(xyz[i], xyz[i + 1], xyz[i + 2]));
  double t[3], q[4];
  for (int i = 0; i < 3; ++i) t[i] = std::atof(argv[2 + i]);
  for (int i = 0; i < 4; ++i) q[i] = std::atof(argv[5 + i]);
  hope::PlaneSegmentRT rt(0.05f, 0.02f, "base_link", "/camera/depth/points");
  rt.configCallback((float)std::atof(argv[9]), (float)std::atof(argv[10]));
  rt.aggressive_merge_ = std::atoi(argv[11]) != 0;
  rt.cloudCallback(cloud, t, q);
  rt.getHorizontalPlanes();
  std::vector<hope::PointCloudMono::Ptr> on_top;
  rt.getClustersUponPlane(on_top);
  std::printf("{\"status\": %d, \"z\": %.9g, \"points\": %zu, \"area\": %.9g, \"contour\": [", rt.last_status(), rt.max_plane_z_,
              rt.max_plane_cloud_->size(), rt.max_plane_area_);
  for (size_t i = 0; i < rt.max_plane_contour_->size(); ++i) {
    const hope::PointXYZ& p = rt.max_plane_contour_->points[i];
    std::printf("%s[%.9g, %.9g, %.9g]", i ? ", " : "", p.x, p.y, p.z);
  }
  std::printf("], \"on_top\": [");
  for (size_t k = 0; k < on_top.size(); ++k) {
    double sx = 0, sy = 0, sz = 0;
    for (const hope::PointXYZ& p : on_top[k]->points) { sx += p.x; sy += p.y; sz += p.z; }
    std::printf("%s[%zu, %.9g, %.9g, %.9g]", k ? ", " : "", on_top[k]->size(), sx, sy, sz);
  }
  std::printf("]}\n");
  return 0;
}
