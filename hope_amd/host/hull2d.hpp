// Convex hull of a horizontal plane patch: what Utilities::computeHull
// (utilities.cpp:1531-1536, called plane_segment.cpp:957) and PlaneSegment::computeHull with
// setComputeAreaVolume (plane_segment.cpp:391-406) get from pcl::ConvexHull -> qhull.
//
// Restated from PCL 1.8.1 surface/impl/convex_hull.hpp (performReconstruction2D), which is not in this
// image: the cloud is projected on xy (the safe projection for a horizontal patch), the hull vertices are
// ORIGINAL input points (x, y, z kept), and they are returned sorted by atan2(y - cy, x - cx) about the
// float centroid of the vertices; with area enabled PCL reports qhull's `totvol`, the polygon area.
// Two things qhull leaves open are fixed here and PARITY IS UNPINNED on them: vertices that are collinear
// within rounding are dropped by an exact orientation test (qhull uses a tolerance), and the centroid sums
// the vertices in ascending input index (qhull's vertex list order is an implementation detail).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "hope_cloud.hpp"

namespace hope {

namespace hull_detail {
// sign of (b-a) x (c-a) on float inputs, exact: every difference of two floats and every product of two
// such differences is representable in long double's 64-bit mantissa only approximately, so do it in
// two doubles (error-free product via fma) and compare.
inline int orient(double ax, double ay, double bx, double by, double cx, double cy)
{
  const double abx = bx - ax, aby = by - ay, acx = cx - ax, acy = cy - ay;   // exact for float inputs
  const double p = abx * acy, q = aby * acx;
  const double pe = std::fma(abx, acy, -p), qe = std::fma(aby, acx, -q);     // exact low parts
  const double hi = p - q;
  if (hi != 0.0 && std::fabs(hi) > 4.0 * (std::fabs(pe) + std::fabs(qe))) return hi > 0 ? 1 : -1;
  const long double d = ((long double)p - (long double)q) + ((long double)pe - (long double)qe);
  return d > 0 ? 1 : d < 0 ? -1 : 0;
}
}  // namespace hull_detail

// indices (into `pts`) of the strict hull vertices of the xy projection, counter-clockwise from the
// lexicographically smallest point; n < 3 distinct points gives what is there.
// `Points`: any random-access container of points with x and y (std::vector<P>, or PCL's vector with Eigen's allocator).
template <class Points>
std::vector<int> hullIndicesXY(const Points& pts, const std::vector<int>* subset = nullptr)
{
  std::vector<int> id;
  if (subset) id = *subset; else { id.resize(pts.size()); for (size_t i = 0; i < pts.size(); ++i) id[i] = (int)i; }
  id.erase(std::remove_if(id.begin(), id.end(), [&](int i) { return !(std::isfinite(pts[i].x) && std::isfinite(pts[i].y)); }), id.end());
  std::sort(id.begin(), id.end(), [&](int a, int b) {
    if (pts[a].x != pts[b].x) return pts[a].x < pts[b].x;
    if (pts[a].y != pts[b].y) return pts[a].y < pts[b].y;
    return a < b; });
  id.erase(std::unique(id.begin(), id.end(), [&](int a, int b) { return pts[a].x == pts[b].x && pts[a].y == pts[b].y; }), id.end());
  const int n = (int)id.size();
  if (n < 3) return id;
  std::vector<int> h(2 * n);
  int k = 0;
  auto turn = [&](int a, int b, int c) { return hull_detail::orient(pts[a].x, pts[a].y, pts[b].x, pts[b].y, pts[c].x, pts[c].y); };
  for (int i = 0; i < n; ++i) { while (k >= 2 && turn(h[k - 2], h[k - 1], id[i]) <= 0) --k; h[k++] = id[i]; }
  for (int i = n - 2, t = k + 1; i >= 0; --i) { while (k >= t && turn(h[k - 2], h[k - 1], id[i]) <= 0) --k; h[k++] = id[i]; }
  h.resize(k - 1);
  return h;
}

// pcl::ConvexHull::reconstruct for a horizontal patch; returns the polygon area (qh totvol).
template <class CloudT>
double computeHull(const CloudT& cloud_2d, CloudT& cloud_hull, std::vector<int>* hull_indices = nullptr,
                   const std::vector<int>* candidates = nullptr)
{
  typedef typename CloudT::PointType P;
  std::vector<int> h = hullIndicesXY(cloud_2d.points, candidates);
  double area2 = 0.0;
  for (size_t i = 0; i + 1 < h.size() && h.size() >= 3; ++i) {   // fan from h[0], ccw order -> positive
    const P &a = cloud_2d.points[h[0]], &b = cloud_2d.points[h[i]], &c = cloud_2d.points[h[i + 1]];
    area2 += ((double)b.x - a.x) * ((double)c.y - a.y) - ((double)b.y - a.y) * ((double)c.x - a.x);
  }
  std::vector<int> asc = h;
  std::sort(asc.begin(), asc.end());
  float cx = 0.f, cy = 0.f;                                       // compute3DCentroid<PointT, float>
  for (int i : asc) { cx += cloud_2d.points[i].x; cy += cloud_2d.points[i].y; }
  if (!asc.empty()) { cx /= (float)asc.size(); cy /= (float)asc.size(); }
  std::vector<std::pair<float, int>> ang;
  for (int i : asc) ang.push_back({std::atan2(cloud_2d.points[i].y - cy, cloud_2d.points[i].x - cx), i});
  std::sort(ang.begin(), ang.end());                              // comparePoints2D; ties by index
  cloud_hull.points.clear();
  if (hull_indices) hull_indices->clear();
  for (auto& a : ang) { cloud_hull.points.push_back(cloud_2d.points[a.second]); if (hull_indices) hull_indices->push_back(a.second); }
  cloud_hull.width = (uint32_t)cloud_hull.points.size(); cloud_hull.height = 1; cloud_hull.is_dense = true;
  return 0.5 * std::fabs(area2);
}

}  // namespace hope
