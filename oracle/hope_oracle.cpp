/*
 * hope_oracle.cpp — CPU restatement of HoPE's PlaneSegment hot path.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (hope_amd/) never links, imports or calls it.
 *
 * PARITY UNPINNED at the PCL / FLANN / Eigen boundary: the reference cannot be built in
 * this image (every hot-path file includes ROS + PCL headers, none of which exist here;
 * SURVEY.md §8c) and its repository holds no test, golden vector or expected output for
 * this path (src/hope_test.cpp asserts nothing and never touches it).  What pins this
 * file: (1) HoPE's own sources, restated line by line below with file:line citations;
 * (2) the PCL 1.8.1 / FLANN 1.9 / Eigen 3.3 algorithms those sources call, restated from
 * their published implementations (each function says which); (3) the input-side anchors
 * of the shipped example pair (210 004 valid pixels, voxel grid 56x46x79, 5 699 voxels;
 * tests/test_oracle_example.py).  Orders that PCL leaves unspecified are fixed here and
 * flagged "ORDER FIXED".
 *
 * Single-threaded scalar C++17, float arithmetic exactly as written (build with
 * -O2 -ffp-contract=off), float libm calls replaced by include/hope_detmath.h (see that
 * header; -DHOPE_ORACLE_LIBM switches back to glibc for the sensitivity test).
 */
#include <algorithm>
#include <chrono>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <queue>
#include <string>
#include <utility>
#include <vector>

#include "../include/hope_detmath.h"
#include "../include/hope_gpu.h"

// Order study (tools/order_study.py): each macro flips one of the orders the oracle fixes where PCL / FLANN /
// Eigen leave them unspecified ("ORDER FIXED" below) to another order those libraries could legally produce.
#ifndef HOPE_ORACLE_VOXEL_ORDER
#define HOPE_ORACLE_VOXEL_ORDER 0
#endif
#ifndef HOPE_ORACLE_TIE_ORDER
#define HOPE_ORACLE_TIE_ORDER 0
#endif
#ifndef HOPE_ORACLE_COV_ORDER
#define HOPE_ORACLE_COV_ORDER 0
#endif

namespace {

#ifdef HOPE_ORACLE_LIBM
inline float o_atan2f(float y, float x) { return std::atan2(y, x); }
inline float o_cosf(float x) { return std::cos(x); }
inline float o_sinf(float x) { return std::sin(x); }
inline float o_asinf(float x) { return std::asin(x); }
inline float o_acosf(float x) { return std::acos(x); }
#else
inline float o_atan2f(float y, float x) { return hope_det_atan2f(y, x); }
inline float o_cosf(float x) { return hope_det_cosf(x); }
inline float o_sinf(float x) { return hope_det_sinf(x); }
inline float o_asinf(float x) { return hope_det_asinf(x); }
inline float o_acosf(float x) { return hope_det_acosf(x); }
#endif

const float kNaN = std::numeric_limits<float>::quiet_NaN();

struct Pose {
  float t[3];
  float q[4];  // x y z w
};

struct Cluster {
  std::vector<int> indices;  // ascending, into the fit cloud
  float z_mean = 0.f;
  int pass = 0;
  float min_x = 0.f, min_y = 0.f, max_x = 0.f, max_y = 0.f;
  int seed_rank = 0;
};

struct Frame {
  hope_counts counts;
  std::vector<float> tx, ty, tz;  // A3 output
  std::vector<int> src_index;
  std::vector<float> vx, vy, vz;           // A5 output
  std::vector<float> nx, ny, nz, curv;     // A6 output
  std::vector<int> fit_idx;                // A7
  std::vector<float> fx, fy, fz;           // cloud_norm_fit_mono_
  std::vector<float> fnx, fny, fnz;        // cloud_norm_fit_
  std::vector<int> knn;                    // A8, n_fit * K
  std::vector<int> labels;                 // per fit point -> cluster (filtered) or -1
  std::vector<Cluster> clusters;           // A9..A12
  hope_max_plane max_plane;
  std::vector<int> max_members;            // every cluster of max_plane_cloud_, in point order (no limit)
};

}  // namespace

struct hope_ctx {
  hope_config cfg;
  float th_theta, th_angle, th_norm;
  const float* forced_bbox = nullptr;  // staged (multi-rank) calls: the all-reduced bounding box
  int stop_after_voxel = 0, input_is_downsampled = 0;
  // staged call, z-band form: k-NN rows given from outside; a row of -1 is a point of another band (no
  // neighbours here), every other row must equal the oracle's own (knn_mismatch otherwise)
  const int32_t* band_knn = nullptr;
  mutable long long knn_mismatch = -1;
  Pose pose;
  std::vector<Pose> frame_poses;
  std::vector<Frame> frames;
  std::string err;
  double whole_ms = 0, ref_ms = 0;
  int use_bruteforce = 0;
  // F1 results (hope_oracle_clusters_upon_plane)
  std::vector<std::vector<int>> upon_clusters;   // indices into upon_xyz / upon_src, ascending
  std::vector<float> upon_x, upon_y, upon_z;
  std::vector<int> upon_src;
};

namespace {

/* ------------------------------------------------------------------------------------
 * Thresholds: plane_segment.cpp:46-50.  th_angle_ = atan(float) -> atanf;
 * th_norm_ = sqrt(1 / (1 + 2 * pow(float, int))) is evaluated in double and narrowed.
 * Host-side scalars, identical code in the product's host layer.
 * ---------------------------------------------------------------------------------- */
void derive_thresholds(hope_ctx* c)
{
  c->th_theta = c->cfg.th_z / c->cfg.th_xy;
  c->th_angle = std::atan(c->th_theta);
  c->th_norm = (float)std::sqrt(1 / (1 + 2 * std::pow((double)c->th_theta, 2)));
}

/* ------------------------------------------------------------------------------------
 * A3  Transform::doTransform, transform.cpp:152-191 (tf variant :72-98).
 * Eigen::Translation3f(t) * Eigen::Quaternionf(w,x,y,z): linear part is
 * Quaternion::toRotationMatrix() (Eigen 3.3 Geometry/Quaternion.h) on the UN-normalised
 * quaternion; Affine * Vector3f = translation + linear * v, where the 3-term dot product
 * of a fixed-size lazy product is Eigen's unrolled redux a0 + (a1 + a2).
 * ---------------------------------------------------------------------------------- */
struct Affine {
  float r[9];
  float t[3];
};

Affine make_affine(const Pose& p)
{
  const float x = p.q[0], y = p.q[1], z = p.q[2], w = p.q[3];
  const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  Affine a;
  a.r[0] = 1.f - (tyy + tzz);
  a.r[1] = txy - twz;
  a.r[2] = txz + twy;
  a.r[3] = txy + twz;
  a.r[4] = 1.f - (txx + tzz);
  a.r[5] = tyz - twx;
  a.r[6] = txz - twy;
  a.r[7] = tyz + twx;
  a.r[8] = 1.f - (txx + tyy);
  a.t[0] = p.t[0];
  a.t[1] = p.t[1];
  a.t[2] = p.t[2];
  return a;
}

inline void apply_affine(const Affine& a, float x, float y, float z, float& ox, float& oy, float& oz)
{
  ox = a.t[0] + (a.r[0] * x + (a.r[1] * y + a.r[2] * z));
  oy = a.t[1] + (a.r[3] * x + (a.r[4] * y + a.r[5] * z));
  oz = a.t[2] + (a.r[6] * x + (a.r[7] * y + a.r[8] * z));
}

/* ------------------------------------------------------------------------------------
 * A5  Utilities::downSampling -> pcl::VoxelGrid<PointXYZ> (utilities.cpp:455-468;
 * PCL 1.8.1 filters/impl/voxel_grid.hpp applyFilter + common/impl/centroid.hpp
 * CentroidPoint).  ORDER FIXED: PCL sorts (voxel idx, point idx) pairs with an unstable
 * std::sort on idx only; here points of one voxel are summed in ascending point index.
 * ---------------------------------------------------------------------------------- */
void voxel_grid(const hope_ctx* c, Frame& f)
{
  const size_t n = f.tx.size();
  f.counts.vox_passthrough = 0;
  for (int d = 0; d < 3; ++d) f.counts.grid[d] = f.counts.min_b[d] = 0;
  if (n == 0) return;
  const float leaf[3] = {c->cfg.th_xy, c->cfg.th_xy, c->cfg.th_z};
  float inv[3];
  for (int d = 0; d < 3; ++d) inv[d] = 1.0f / leaf[d];
  // getMinMax3D over finite points
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  size_t n_finite = 0;
  for (size_t i = 0; i < n; ++i) {
    const float p[3] = {f.tx[i], f.ty[i], f.tz[i]};
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    ++n_finite;
    for (int d = 0; d < 3; ++d) {
      mn[d] = std::min(mn[d], p[d]);
      mx[d] = std::max(mx[d], p[d]);
    }
  }
  if (n_finite == 0) return;
  if (c->forced_bbox) {
    for (int d = 0; d < 3; ++d) {
      mn[d] = c->forced_bbox[d];
      mx[d] = c->forced_bbox[3 + d];
    }
  }
  int64_t dxyz[3];
  for (int d = 0; d < 3; ++d) dxyz[d] = static_cast<int64_t>((mx[d] - mn[d]) * inv[d]) + 1;
  if ((dxyz[0] * dxyz[1] * dxyz[2]) > static_cast<int64_t>(std::numeric_limits<int32_t>::max())) {
    // "Leaf size is too small for the input dataset": output = *input_
    f.counts.vox_passthrough = 1;
    f.vx = f.tx;
    f.vy = f.ty;
    f.vz = f.tz;
    return;
  }
  int min_b[3], max_b[3], div_b[3];
  for (int d = 0; d < 3; ++d) {
    min_b[d] = static_cast<int>(std::floor(mn[d] * inv[d]));
    max_b[d] = static_cast<int>(std::floor(mx[d] * inv[d]));
    div_b[d] = max_b[d] - min_b[d] + 1;
    f.counts.grid[d] = div_b[d];
    f.counts.min_b[d] = min_b[d];
  }
  const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
  std::vector<std::pair<unsigned, unsigned>> iv;  // (idx, point)
  iv.reserve(n);
  for (size_t i = 0; i < n; ++i) {
    const float p[3] = {f.tx[i], f.ty[i], f.tz[i]};
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    int ijk[3];
    for (int d = 0; d < 3; ++d)
      ijk[d] = static_cast<int>(std::floor(p[d] * inv[d]) - static_cast<float>(min_b[d]));
    const int idx = ijk[0] * mul[0] + ijk[1] * mul[1] + ijk[2] * mul[2];
    iv.emplace_back((unsigned)idx, (unsigned)i);
  }
#if HOPE_ORACLE_VOXEL_ORDER == 1
  // order study (tools/order_study.py): PCL's own call, std::sort with operator< on idx only -- unstable, so the
  // points of a voxel are summed in whatever order libstdc++'s introsort leaves them
  std::sort(iv.begin(), iv.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
#elif HOPE_ORACLE_VOXEL_ORDER == 2
  // order study: descending point index inside a voxel
  std::stable_sort(iv.begin(), iv.end(), [](const auto& a, const auto& b) {
    return a.first < b.first || (a.first == b.first && a.second > b.second);
  });
#else
  std::stable_sort(iv.begin(), iv.end(),
                   [](const auto& a, const auto& b) { return a.first < b.first; });
#endif
  size_t i = 0;
  while (i < iv.size()) {
    size_t j = i;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    while (j < iv.size() && iv[j].first == iv[i].first) {
      const unsigned p = iv[j].second;
      sx += f.tx[p];
      sy += f.ty[p];
      sz += f.tz[p];
      ++j;
    }
    const float cnt = static_cast<float>(j - i);
    f.vx.push_back(sx / cnt);
    f.vy.push_back(sy / cnt);
    f.vz.push_back(sz / cnt);
    i = j;
  }
}

/* ------------------------------------------------------------------------------------
 * Exact neighbour search standing in for pcl::search::KdTree -> FLANN
 * KDTreeSingleIndex<L2_Simple<float>>.  Distances are FLANN's L2_Simple accumulation
 * ((dx*dx + dy*dy) + dz*dz in float).  ORDER FIXED: FLANN breaks distance ties by tree
 * traversal order; here ties are broken by ascending point index.
 * ---------------------------------------------------------------------------------- */
struct KdTree {
  const float *x, *y, *z;
  int n;
  std::vector<int> perm;
  struct Node {
    int lo, hi;       // leaf: range in perm
    int left, right;  // children (-1 for leaf)
    int dim;
    float split_lo, split_hi;  // max of left side / min of right side along dim
  };
  std::vector<Node> nodes;
  static constexpr int kLeaf = 12;

  inline float coord(int p, int d) const { return d == 0 ? x[p] : (d == 1 ? y[p] : z[p]); }

  int build(int lo, int hi)
  {
    Node nd;
    nd.lo = lo;
    nd.hi = hi;
    nd.left = nd.right = -1;
    nd.dim = 0;
    nd.split_lo = nd.split_hi = 0.f;
    const int id = (int)nodes.size();
    nodes.push_back(nd);
    if (hi - lo <= kLeaf) return id;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = lo; i < hi; ++i)
      for (int d = 0; d < 3; ++d) {
        const float v = coord(perm[i], d);
        mn[d] = std::min(mn[d], v);
        mx[d] = std::max(mx[d], v);
      }
    int dim = 0;
    for (int d = 1; d < 3; ++d)
      if (mx[d] - mn[d] > mx[dim] - mn[dim]) dim = d;
    if (!(mx[dim] > mn[dim])) return id;  // all identical: keep as (big) leaf
    const int mid = (lo + hi) / 2;
    std::nth_element(perm.begin() + lo, perm.begin() + mid, perm.begin() + hi,
                     [&](int a, int b) { return coord(a, dim) < coord(b, dim); });
    float slo = -FLT_MAX, shi = FLT_MAX;
    for (int i = lo; i < mid; ++i) slo = std::max(slo, coord(perm[i], dim));
    for (int i = mid; i < hi; ++i) shi = std::min(shi, coord(perm[i], dim));
    const int l = build(lo, mid);
    const int r = build(mid, hi);
    nodes[id].left = l;
    nodes[id].right = r;
    nodes[id].dim = dim;
    nodes[id].split_lo = slo;
    nodes[id].split_hi = shi;
    return id;
  }

  void init(const float* px, const float* py, const float* pz, int count)
  {
    x = px;
    y = py;
    z = pz;
    n = count;
    perm.resize(n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    nodes.clear();
    if (n > 0) build(0, n);
  }

  static inline float dist2(float ax, float ay, float az, float bx, float by, float bz)
  {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return (dx * dx + dy * dy) + dz * dz;
  }

#if HOPE_ORACLE_TIE_ORDER == 1
  struct DI {  // order study: equidistant neighbours in DESCENDING index
    float first;
    int second;
    DI() : first(0.f), second(0) {}
    DI(float d, int i) : first(d), second(i) {}
    bool operator<(const DI& o) const { return first < o.first || (first == o.first && second > o.second); }
  };
#else
  typedef std::pair<float, int> DI;  // lexicographic (dist2, index)
#endif

  void radius_rec(int id, float qx, float qy, float qz, float r2, std::vector<DI>& out) const
  {
    const Node& nd = nodes[id];
    if (nd.left < 0) {
      for (int i = nd.lo; i < nd.hi; ++i) {
        const int p = perm[i];
        const float d = dist2(qx, qy, qz, x[p], y[p], z[p]);
        if (d < r2) out.emplace_back(d, p);
      }
      return;
    }
    const float q = nd.dim == 0 ? qx : (nd.dim == 1 ? qy : qz);
    // left side holds coords <= split_lo, right side coords >= split_hi
    const float dl = q - nd.split_lo;  // > 0: query is beyond the left side
    const float dr = nd.split_hi - q;  // > 0: query is before the right side
    if (!(dl > 0.f && dl * dl >= r2)) radius_rec(nd.left, qx, qy, qz, r2, out);
    if (!(dr > 0.f && dr * dr >= r2)) radius_rec(nd.right, qx, qy, qz, r2, out);
  }

  void knn_rec(int id, float qx, float qy, float qz, int k, std::vector<DI>& best) const
  {
    const Node& nd = nodes[id];
    if (nd.left < 0) {
      for (int i = nd.lo; i < nd.hi; ++i) {
        const int p = perm[i];
        const DI cand(dist2(qx, qy, qz, x[p], y[p], z[p]), p);
        if ((int)best.size() < k) {
          best.insert(std::upper_bound(best.begin(), best.end(), cand), cand);
        } else if (cand < best.back()) {
          best.pop_back();
          best.insert(std::upper_bound(best.begin(), best.end(), cand), cand);
        }
      }
      return;
    }
    const float q = nd.dim == 0 ? qx : (nd.dim == 1 ? qy : qz);
    const float dl = q - nd.split_lo;
    const float dr = nd.split_hi - q;
    const int first = (dl <= 0.f || (dr > 0.f && dl < dr)) ? 0 : 1;
    for (int s = 0; s < 2; ++s) {
      const int side = s == 0 ? first : 1 - first;
      const float dd = side == 0 ? dl : dr;
      // a subtree can only be skipped when every point in it is strictly farther than
      // the current k-th candidate (equal distance may still win on index)
      if ((int)best.size() == k && dd > 0.f && dd * dd > best.back().first) continue;
      knn_rec(side == 0 ? nd.left : nd.right, qx, qy, qz, k, best);
    }
  }
};

/* ------------------------------------------------------------------------------------
 * pcl::computeRoots2 / computeRoots / eigen33 (smallest eigenpair), PCL 1.8.1
 * common/impl/eigen.hpp, for Eigen::Matrix3f.
 * ---------------------------------------------------------------------------------- */
void compute_roots2(float b, float c, float roots[3])
{
  roots[0] = 0.f;
  float d = (float)((double)(b * b) - 4.0 * (double)c);
  if (d < 0.0) d = 0.0;
  const float sd = (float)std::sqrt((double)d);
  roots[2] = 0.5f * (b + sd);
  roots[1] = 0.5f * (b - sd);
}

void compute_roots(const float m[9], float roots[3])
{
  // m is symmetric, row-major m[3*r+c]
  const float m00 = m[0], m01 = m[1], m02 = m[2], m11 = m[4], m12 = m[5], m22 = m[8];
  const float c0 = m00 * m11 * m22 + 2.f * m01 * m02 * m12 - m00 * m12 * m12 - m11 * m02 * m02 -
                   m22 * m01 * m01;
  const float c1 = m00 * m11 - m01 * m01 + m00 * m22 - m02 * m02 + m11 * m22 - m12 * m12;
  const float c2 = m00 + m11 + m22;
  if (std::fabs(c0) < std::numeric_limits<float>::epsilon()) {
    compute_roots2(c2, c1, roots);
    return;
  }
  const float s_inv3 = (float)(1.0 / 3.0);
  const float s_sqrt3 = std::sqrt(3.0f);
  const float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.f) a_over_3 = 0.f;
  const float half_b = 0.5f * (c0 + c2_over_3 * (2.f * c2_over_3 * c2_over_3 - c1));
  float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.f) q = 0.f;
  const float rho = std::sqrt(-a_over_3);
  const float theta = o_atan2f(std::sqrt(-q), half_b) * s_inv3;
  const float cos_theta = o_cosf(theta);
  const float sin_theta = o_sinf(theta);
  roots[0] = c2_over_3 + 2.f * rho * cos_theta;
  roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]);
  if (roots[1] >= roots[2]) {
    std::swap(roots[1], roots[2]);
    if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]);
  }
  if (roots[0] <= 0.f) compute_roots2(c2, c1, roots);
}

inline void cross3(const float* a, const float* b, float* o)
{
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

inline float sqnorm3(const float* v) { return v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]); }

void eigen33_smallest(const float mat[9], float& eigenvalue, float evec[3])
{
  float scale = 0.f;
  for (int i = 0; i < 9; ++i) scale = std::max(scale, std::fabs(mat[i]));
  if (scale <= std::numeric_limits<float>::min()) scale = 1.f;
  float sm[9];
  for (int i = 0; i < 9; ++i) sm[i] = mat[i] / scale;
  float ev[3];
  compute_roots(sm, ev);
  eigenvalue = ev[0] * scale;
  sm[0] -= ev[0];
  sm[4] -= ev[0];
  sm[8] -= ev[0];
  float v1[3], v2[3], v3[3];
  cross3(sm + 0, sm + 3, v1);
  cross3(sm + 0, sm + 6, v2);
  cross3(sm + 3, sm + 6, v3);
  const float l1 = sqnorm3(v1), l2 = sqnorm3(v2), l3 = sqnorm3(v3);
  const float* v;
  float l;
  if (l1 >= l2 && l1 >= l3) {
    v = v1;
    l = l1;
  } else if (l2 >= l1 && l2 >= l3) {
    v = v2;
    l = l2;
  } else {
    v = v3;
    l = l3;
  }
  const float s = std::sqrt(l);
  for (int d = 0; d < 3; ++d) evec[d] = v[d] / s;
}

/* ------------------------------------------------------------------------------------
 * A6  Utilities::estimateNorm -> pcl::NormalEstimation<PointXYZ,Normal>
 * (utilities.cpp:428-446; PCL 1.8.1 features/impl/normal_3d.hpp computeFeature,
 * features/normal_3d.h computePointNormal / flipNormalTowardsViewpoint,
 * common/impl/centroid.hpp computeMeanAndCovarianceMatrix single-pass float version).
 * Radius = (double)(float)(1.01 * th_xy)  (plane_segment.cpp:695 narrows to the float
 * parameter norm_r); FLANN tests dist < (float)(radius*radius).
 * ---------------------------------------------------------------------------------- */
void estimate_normals(const hope_ctx* c, Frame& f, const KdTree* tree)
{
  const int n = (int)f.vx.size();
  f.nx.assign(n, kNaN);
  f.ny.assign(n, kNaN);
  f.nz.assign(n, kNaN);
  f.curv.assign(n, kNaN);
  const float norm_r = (float)(1.01 * c->cfg.th_xy);
  const double radius = (double)norm_r;
  const float r2 = static_cast<float>(radius * radius);
  std::vector<KdTree::DI> nb;
  for (int i = 0; i < n; ++i) {
    nb.clear();
    const float qx = f.vx[i], qy = f.vy[i], qz = f.vz[i];
    if (tree) {
      tree->radius_rec(0, qx, qy, qz, r2, nb);
    } else {
      for (int p = 0; p < n; ++p) {
        const float d = KdTree::dist2(qx, qy, qz, f.vx[p], f.vy[p], f.vz[p]);
        if (d < r2) nb.emplace_back(d, p);
      }
    }
    std::sort(nb.begin(), nb.end());
    if (nb.size() < 3) continue;
    float accu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (const auto& di : nb) {
      const float x = f.vx[di.second], y = f.vy[di.second], z = f.vz[di.second];
      accu[0] += x * x;
      accu[1] += x * y;
      accu[2] += x * z;
      accu[3] += y * y;
      accu[4] += y * z;
      accu[5] += z * z;
      accu[6] += x;
      accu[7] += y;
      accu[8] += z;
    }
    const float cnt = static_cast<float>(nb.size());
    for (int k = 0; k < 9; ++k) accu[k] /= cnt;
    float cov[9];
    cov[0] = accu[0] - accu[6] * accu[6];
    cov[1] = accu[1] - accu[6] * accu[7];
    cov[2] = accu[2] - accu[6] * accu[8];
    cov[4] = accu[3] - accu[7] * accu[7];
    cov[5] = accu[4] - accu[7] * accu[8];
    cov[8] = accu[5] - accu[8] * accu[8];
    cov[3] = cov[1];
    cov[6] = cov[2];
    cov[7] = cov[5];
    float eval, evec[3];
    eigen33_smallest(cov, eval, evec);
    float nx = evec[0], ny = evec[1], nz = evec[2];
    const float eig_sum = cov[0] + cov[4] + cov[8];
    float curvature = 0.f;
    if (eig_sum != 0) curvature = std::fabs(eval / eig_sum);
    // flipNormalTowardsViewpoint(point, 0, 0, 0, ...)
    const float vpx = 0.f - qx, vpy = 0.f - qy, vpz = 0.f - qz;
    const float cos_theta = (vpx * nx + vpy * ny + vpz * nz);
    if (cos_theta < 0) {
      nx *= -1;
      ny *= -1;
      nz *= -1;
    }
    f.nx[i] = nx;
    f.ny[i] = ny;
    f.nz[i] = nz;
    f.curv[i] = curvature;
  }
}

/* ------------------------------------------------------------------------------------
 * A8 + A9  ZGrowing::extract, z_growing.cpp:121-165: findPointNeighbours (:196-226),
 * applySmoothRegionGrowingAlgorithm (:229-282), growRegion (:285-332), validatePoint
 * (:335-364, smooth mode), assembleRegions (:367-396), size filter (:149-162).
 * The seed order is whatever libstdc++ std::sort leaves when every key compares equal
 * (point_residual[i].first is value-initialised 0.0f; pcl::comparePair compares .first)
 * — obtained here by literally calling std::sort with that comparator.
 * ---------------------------------------------------------------------------------- */
bool compare_pair(std::pair<float, int> i, std::pair<float, int> j) { return (i.first < j.first); }

void z_growing(const hope_ctx* c, Frame& f, const KdTree* tree)
{
  const int n = (int)f.fx.size();
  const int K = c->cfg.knn;
  const int k = std::min(K, n);
  f.knn.assign((size_t)n * K, -1);
  std::vector<KdTree::DI> best;
  for (int i = 0; i < n; ++i) {
    best.clear();
    if (tree) {
      tree->knn_rec(0, f.fx[i], f.fy[i], f.fz[i], k, best);
    } else {
      for (int p = 0; p < n; ++p)
        best.emplace_back(KdTree::dist2(f.fx[i], f.fy[i], f.fz[i], f.fx[p], f.fy[p], f.fz[p]), p);
      std::sort(best.begin(), best.end());
      best.resize(k);
    }
    for (int j = 0; j < (int)best.size(); ++j) f.knn[(size_t)i * K + j] = best[j].second;
  }
  if (c->band_knn) {
    for (int i = 0; i < n; ++i) {
      const int32_t* row = c->band_knn + (size_t)i * K;
      bool foreign = true;
      for (int j = 0; j < K; ++j) foreign &= row[j] == -1;
      for (int j = 0; j < K; ++j) {
        if (foreign) f.knn[(size_t)i * K + j] = -1;
        else if (f.knn[(size_t)i * K + j] != row[j] && c->knn_mismatch < 0) c->knn_mismatch = (long long)i * K + j;
      }
    }
  }

  std::vector<int> point_labels(n, -1);
  std::vector<std::pair<float, int>> point_residual(n, std::pair<float, int>());
  for (int i = 0; i < n; ++i) point_residual[i].second = i;
  std::sort(point_residual.begin(), point_residual.end(), compare_pair);

  std::vector<int> num_pts_in_segment;
  std::vector<int> seg_seed_rank;
  const float z_threshold = c->cfg.th_z;
  int seed_counter = 0;
  int seed = n > 0 ? point_residual[0].second : 0;
  int segmented = 0;
  int number_of_segments = 0;
  while (segmented < n) {
    // growRegion(seed, number_of_segments)
    std::queue<int> seeds;
    seeds.push(seed);
    point_labels[seed] = number_of_segments;
    int num = 1;
    while (!seeds.empty()) {
      const int curr = seeds.front();
      seeds.pop();
      for (int j = 0; j < K; ++j) {
        const int index = f.knn[(size_t)curr * K + j];
        if (index < 0) break;
        if (point_labels[index] != -1) continue;
        const float delta_z = std::fabs(f.fz[curr] - f.fz[index]);
        if (!(delta_z < z_threshold)) continue;
        point_labels[index] = number_of_segments;
        ++num;
        seeds.push(index);
      }
    }
    segmented += num;
    num_pts_in_segment.push_back(num);
    seg_seed_rank.push_back(seed_counter);
    ++number_of_segments;
    for (int i_seed = seed_counter + 1; i_seed < n; ++i_seed) {
      const int index = point_residual[i_seed].second;
      if (point_labels[index] == -1) {
        seed = index;
        seed_counter = i_seed;
        break;
      }
    }
  }
  f.counts.n_segments = number_of_segments;
  // assembleRegions + size filter
  std::vector<std::vector<int>> regions(number_of_segments);
  for (int s = 0; s < number_of_segments; ++s) regions[s].reserve(num_pts_in_segment[s]);
  for (int i = 0; i < n; ++i)
    if (point_labels[i] != -1) regions[point_labels[i]].push_back(i);
  f.labels.assign(n, -1);
  for (int s = 0; s < number_of_segments; ++s) {
    const int sz = (int)regions[s].size();
    if (sz >= c->cfg.min_cluster && sz <= INT_MAX) {
      Cluster cl;
      cl.indices.swap(regions[s]);
      cl.seed_rank = seg_seed_rank[s];
      for (int p : cl.indices) f.labels[p] = (int)f.clusters.size();
      f.clusters.push_back(std::move(cl));
    }
  }
  f.counts.n_clusters = (int)f.clusters.size();
}

/* ------------------------------------------------------------------------------------
 * A11  Utilities::normalAnalysis + calNormalMean, utilities.cpp:151-282.
 * Eigen rowwise().mean() on a column-major 3xN matrix is a sequential sum per row
 * divided by N.  Vector3f::norm() is sqrt(a0 + (a1 + a2)) (Eigen's unrolled redux).
 * ORDER FIXED: tmp * tmp^T goes through Eigen's GEMM for N >= ~16, whose blocking order
 * is unspecified; here the three covariance sums are sequential in point order.
 * The 2x2 general EigenSolver is replaced by the closed-form symmetric solution (larger
 * eigenvalue's eigenvector; (0,1) when the matrix is a multiple of the identity, which
 * is the column the reference picks when the eigenvalues tie).
 * ---------------------------------------------------------------------------------- */
void principal_axis_2x2(float a, float b, float d, float& ax, float& ay)
{
  const float p = 0.5f * (a - d);
  const float z = std::sqrt(p * p + b * b);
  if (z == 0.f) {
    ax = 0.f;
    ay = 1.f;
    return;
  }
  float vx, vy;
  if (p >= 0.f) {
    vx = p + z;
    vy = b;
  } else {
    vx = b;
    vy = z - p;
  }
  const float nrm = std::sqrt(vx * vx + vy * vy);
  ax = vx / nrm;
  ay = vy / nrm;
}

bool normal_analysis(const Frame& f, const std::vector<int>& idx, float th_angle)
{
  const size_t sz = idx.size();
  if (sz <= 2) return false;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (size_t i = 0; i < sz; ++i) {
    s0 += f.fnx[idx[i]];
    s1 += f.fny[idx[i]];
    s2 += std::fabs(f.fnz[idx[i]]);
  }
  const float fsz = static_cast<float>(sz);
  const float m0 = s0 / fsz, m1 = s1 / fsz, m2 = s2 / fsz;
  const float proj_norm = std::sqrt(m0 * m0 + m1 * m1);
  const float mean_norm = std::sqrt(m0 * m0 + (m1 * m1 + m2 * m2));
  const float grad = o_asinf(proj_norm / mean_norm);
  if (grad > th_angle) return false;

  // mean_2d is recomputed from data_2d: same sums, same order
  const float mean2_0 = m0, mean2_1 = m1;
  float cxx = 0.f, cxy = 0.f, cyy = 0.f;
  for (size_t k = 0; k < sz; ++k) {
#if HOPE_ORACLE_COV_ORDER == 1
    const size_t i = sz - 1 - k;  // order study: the covariance sums in reverse point order
#else
    const size_t i = k;
#endif
    const float tx = f.fnx[idx[i]] - mean2_0;
    const float ty = f.fny[idx[i]] - mean2_1;
    cxx += tx * tx;
    cxy += tx * ty;
    cyy += ty * ty;
  }
  const float denom = static_cast<float>(sz - 1);
  cxx /= denom;
  cxy /= denom;
  cyy /= denom;
  float ax, ay;
  principal_axis_2x2(cxx, cxy, cyy, ax, ay);

  float a0 = 0.f, a1 = 0.f, a2 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f;
  size_t na = 0, nb = 0;
  for (size_t i = 0; i < sz; ++i) {
    const float nx = f.fnx[idx[i]], ny = f.fny[idx[i]], nz = std::fabs(f.fnz[idx[i]]);
    const float tx = nx - mean2_0;
    const float ty = ny - mean2_1;
    if (ax * tx + ay * ty > 0) {
      a0 += nx;
      a1 += ny;
      a2 += nz;
      ++na;
    } else {
      b0 += nx;
      b1 += ny;
      b2 += nz;
      ++nb;
    }
  }
  if (na == 0 || nb == 0) return false;
  const float fa = static_cast<float>(na), fb = static_cast<float>(nb);
  const float p1[3] = {a0 / fa, a1 / fa, a2 / fa};
  const float p2[3] = {b0 / fb, b1 / fb, b2 / fb};
  const float mu = p2[0] * p1[0] + (p2[1] * p1[1] + p2[2] * p1[2]);
  const float n1 = std::sqrt(sqnorm3(p1));
  const float n2 = std::sqrt(sqnorm3(p2));
  const float rad_mu = o_acosf(mu / (n1 * n2));
  return rad_mu <= th_angle;
}

/* ------------------------------------------------------------------------------------
 * The per-frame pipeline after A3: plane_segment.cpp:134-159 (offline) / :755-764 (RT).
 * ---------------------------------------------------------------------------------- */
void run_frame_after_transform(hope_ctx* c, Frame& f, double& ref_ms)
{
  f.counts.n_raw = (int64_t)f.tx.size();
  f.counts.status = HOPE_OK;
  if (c->input_is_downsampled) {
    // the input is VoxelGrid's output already: only the grid is derived (from the forced bbox)
    const float inv[3] = {1.0f / c->cfg.th_xy, 1.0f / c->cfg.th_xy, 1.0f / c->cfg.th_z};
    for (int d = 0; d < 3; ++d) {
      f.counts.min_b[d] = static_cast<int>(std::floor(c->forced_bbox[d] * inv[d]));
      f.counts.grid[d] = static_cast<int>(std::floor(c->forced_bbox[3 + d] * inv[d])) - f.counts.min_b[d] + 1;
    }
    f.vx = f.tx;
    f.vy = f.ty;
    f.vz = f.tz;
  } else {
    voxel_grid(c, f);
  }
  f.counts.n_vox = (int64_t)f.vx.size();
  if (f.vx.empty()) {
    f.counts.status = HOPE_EMPTY;
    return;
  }
  if (c->stop_after_voxel) return;
  KdTree tree;
  const bool use_tree = !c->use_bruteforce;
  if (use_tree) tree.init(f.vx.data(), f.vy.data(), f.vz.data(), (int)f.vx.size());
  estimate_normals(c, f, use_tree ? &tree : nullptr);

  // A7  getCloudByNorm + getCloudByInliers x2, utilities.cpp:629-645,677-701
  for (int i = 0; i < (int)f.vx.size(); ++i)
    if (std::fabs(f.nz[i]) > c->th_norm) f.fit_idx.push_back(i);
  f.counts.n_fit = (int64_t)f.fit_idx.size();
  if (f.fit_idx.empty()) {
    f.counts.status = HOPE_NO_NORMAL_FIT;
    return;
  }
  for (int i : f.fit_idx) {
    f.fx.push_back(f.vx[i]);
    f.fy.push_back(f.vy[i]);
    f.fz.push_back(f.vz[i]);
    f.fnx.push_back(f.nx[i]);
    f.fny.push_back(f.ny[i]);
    f.fnz.push_back(f.nz[i]);
  }

  const auto t0 = std::chrono::steady_clock::now();  // hst_.start(), plane_segment.cpp:157
  KdTree tree2;
  if (use_tree) tree2.init(f.fx.data(), f.fy.data(), f.fz.data(), (int)f.fx.size());
  z_growing(c, f, use_tree ? &tree2 : nullptr);

  // A10 getMeanZofEachCluster (plane_segment.cpp:286-322) + getCloudZInfo
  // (utilities.cpp:751-769); A11 getPlane -> gaussianImageAnalysis; A12 setFeatures.
  int n_planes = 0;
  f.max_plane.valid = 0;
  f.max_plane.n_points = 0;
  f.max_plane.z = -1000.0f;
  f.max_plane.n_members = 0;
  f.max_members.clear();
  for (size_t ci = 0; ci < f.clusters.size(); ++ci) {
    Cluster& cl = f.clusters[ci];
    float mid = 0.0f;
    size_t ct = 0;
    for (int p : cl.indices) {
      if (std::isfinite(f.fz[p])) {
        mid += f.fz[p];
        ct++;
      }
    }
    cl.z_mean = mid / ct;
    // setFeatures: pcl::getMinMax3D
    float mnx = FLT_MAX, mny = FLT_MAX, mxx = -FLT_MAX, mxy = -FLT_MAX;
    for (int p : cl.indices) {
      mnx = std::min(mnx, f.fx[p]);
      mny = std::min(mny, f.fy[p]);
      mxx = std::max(mxx, f.fx[p]);
      mxy = std::max(mxy, f.fy[p]);
    }
    cl.min_x = mnx;
    cl.min_y = mny;
    cl.max_x = mxx;
    cl.max_y = mxy;
    cl.pass = normal_analysis(f, cl.indices, c->th_angle) ? 1 : 0;
    if (c->cfg.mode == HOPE_MODE_OFFLINE) {
      if (cl.pass) ++n_planes;  // plane_points_.push_back, plane_segment.cpp:368-373
    } else {
      // PlaneSegmentRT::getPlane, plane_segment.cpp:930-962.  max_plane_cloud_ is kept as
      // the list of clusters whose points it concatenates (combineCloud appends the old
      // max plane AFTER the new cluster, utilities.cpp:1717-1727).
      const float z_in = cl.z_mean;
      if (z_in > c->cfg.min_height && z_in < c->cfg.max_height) {
        if (!cl.pass) continue;
        ++n_planes;
        std::vector<int> members(1, (int)ci);
        int count = (int)cl.indices.size();
        if (c->cfg.aggressive_merge && f.max_plane.valid) {
          if (std::fabs(f.max_plane.z - z_in) <= c->cfg.th_z) {
            members.insert(members.end(), f.max_members.begin(), f.max_members.end());
            count += f.max_plane.n_points;
          }
        }
        if (count > f.max_plane.n_points) {
          f.max_plane.valid = 1;
          f.max_plane.n_points = count;
          f.max_plane.z = z_in;
          f.max_members = members;  // plane_segment.cpp:943-955 concatenates without limit
          f.max_plane.n_members = (int)members.size();
          for (size_t m = 0; m < members.size() && m < 16; ++m) f.max_plane.members[m] = members[m];
        }
      }
    }
  }
  f.counts.n_planes = n_planes;
  const auto t1 = std::chrono::steady_clock::now();  // hst_.stop()
  ref_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
}

}  // namespace

/* =====================================================================================
 * C ABI (same signatures as include/hope_gpu.h, prefix hope_oracle_)
 * =================================================================================== */
#define ORACLE_API extern "C" __attribute__((visibility("default")))

ORACLE_API int hope_oracle_config_default(hope_config* cfg)
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

ORACLE_API int hope_oracle_create(const hope_config* cfg, hope_ctx** out)
{
  if (!cfg || !out) return HOPE_E_INVALID;
  if (!(cfg->th_xy > 0.f) || !(cfg->th_z > 0.f) || cfg->knn < 1 || cfg->knn > 32) return HOPE_E_INVALID;
  hope_ctx* c = new hope_ctx();
  c->cfg = *cfg;
  derive_thresholds(c);
  c->pose = Pose{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 1.f}};
  *out = c;
  return HOPE_OK;
}

ORACLE_API void hope_oracle_destroy(hope_ctx* ctx) { delete ctx; }

ORACLE_API const char* hope_oracle_last_error(const hope_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

ORACLE_API int hope_oracle_set_pose(hope_ctx* ctx, const float t[3], const float q[4])
{
  if (!ctx || !t || !q) return HOPE_E_INVALID;
  for (int i = 0; i < 3; ++i) ctx->pose.t[i] = t[i];
  for (int i = 0; i < 4; ++i) ctx->pose.q[i] = q[i];
  ctx->frame_poses.clear();
  return HOPE_OK;
}

// transform.cpp:108-130: the pose of the roll/pitch/yaw overload of doTransform.  The translation is the file's
// camera-to-base constants (transform.cpp:4-6: 0, 0, 1 m); the rotation is tf2::Quaternion::setRPY (geometry2,
// tf2/LinearMath/Quaternion.h -- a ROS dependency that is not part of the reference tree; its published formula is
// restated here in tf2Scalar = double, with libm's cos/sin as tf2Cos/tf2Sin are), then y is negated (:118) and the
// four doubles are narrowed to float by Eigen::Quaternion<float>(w, x, y, z) (:130).
ORACLE_API int hope_oracle_set_pose_rpy(hope_ctx* ctx, float roll, float pitch, float yaw)
{
  if (!ctx) return HOPE_E_INVALID;
  const double half_yaw = (double)yaw * 0.5, half_pitch = (double)pitch * 0.5, half_roll = (double)roll * 0.5;
  const double cy = std::cos(half_yaw), sy = std::sin(half_yaw);
  const double cp = std::cos(half_pitch), sp = std::sin(half_pitch);
  const double cr = std::cos(half_roll), sr = std::sin(half_roll);
  const double qx = sr * cp * cy - cr * sp * sy;
  const double qy = cr * sp * cy + sr * cp * sy;
  const double qz = cr * cp * sy - sr * sp * cy;
  const double qw = cr * cp * cy + sr * sp * sy;
  const float t[3] = {0.f, 0.f, 1.0f};
  const float q[4] = {(float)qx, (float)(-qy), (float)qz, (float)qw};
  return hope_oracle_set_pose(ctx, t, q);
}

ORACLE_API int hope_oracle_set_frame_poses(hope_ctx* ctx, int n, const float* t, const float* q)
{
  if (!ctx || n < 0 || (n > 0 && (!t || !q))) return HOPE_E_INVALID;
  ctx->frame_poses.resize(n);
  for (int f = 0; f < n; ++f) {
    for (int i = 0; i < 3; ++i) ctx->frame_poses[f].t[i] = t[3 * f + i];
    for (int i = 0; i < 4; ++i) ctx->frame_poses[f].q[i] = q[4 * f + i];
  }
  return HOPE_OK;
}

ORACLE_API int hope_oracle_set_height_window(hope_ctx* ctx, float mn, float mx)
{
  if (!ctx) return HOPE_E_INVALID;
  ctx->cfg.min_height = mn;
  ctx->cfg.max_height = mx;
  return HOPE_OK;
}

/* test hook: 1 = O(n^2) neighbour search instead of the kd-tree */
ORACLE_API int hope_oracle_set_bruteforce(hope_ctx* ctx, int on)
{
  if (!ctx) return HOPE_E_INVALID;
  ctx->use_bruteforce = on;
  return HOPE_OK;
}

static int worst_status(const hope_ctx* ctx)
{
  int st = HOPE_OK;
  for (const Frame& f : ctx->frames) st = std::max(st, (int)f.counts.status);
  return st;
}

ORACLE_API int hope_oracle_segment_depth_u16(hope_ctx* ctx, const uint16_t* depth, int w, int h, int n_frames,
                                             int /*on_device*/)
{
  if (!ctx || !depth || w <= 0 || h <= 0 || n_frames <= 0) return HOPE_E_INVALID;
  if (!ctx->frame_poses.empty() && (int)ctx->frame_poses.size() != n_frames) {
    ctx->err = "frame pose count does not match n_frames";
    return HOPE_E_INVALID;
  }
  const auto t0 = std::chrono::steady_clock::now();
  ctx->ref_ms = 0;
  ctx->frames.clear();
  ctx->frames.resize(n_frames);
  const hope_config& cfg = ctx->cfg;
  for (int fi = 0; fi < n_frames; ++fi) {
    Frame& f = ctx->frames[fi];
    std::memset(&f.counts, 0, sizeof(f.counts));
    std::memset(&f.max_plane, 0, sizeof(f.max_plane));
    f.counts.n_in = (int64_t)w * h;
    const Affine a = make_affine(ctx->frame_poses.empty() ? ctx->pose : ctx->frame_poses[fi]);
    const uint16_t* d = depth + (size_t)fi * w * h;
    for (int i = 0; i < w * h; ++i) {
      /* A1 GetCloud::getColorCloud, get_cloud.cpp:220-233 (the reference hard-codes the
       * width 640 at :224-225; generalised to w) */
      const int col = i % w;
      const int row = i / w;
      float z = float(d[i]) / cfg.depth_scale;
      float x, y;
      if (z < cfg.max_depth && z > cfg.min_depth) {
        x = (col - cfg.cx) * z / cfg.fx;
        y = (row - cfg.cy) * z / cfg.fy;
      } else {
        x = y = z = kNaN;
      }
      /* A2 Utilities::getCloudByZ -> pcl::PassThrough, utilities.cpp:662-675 */
      if (!std::isfinite(x) || !std::isfinite(y) || !std::isfinite(z)) continue;
      if (z < cfg.min_depth || z > cfg.max_depth) continue;
      /* A3 */
      float ox, oy, oz;
      apply_affine(a, x, y, z, ox, oy, oz);
      f.tx.push_back(ox);
      f.ty.push_back(oy);
      f.tz.push_back(oz);
      f.src_index.push_back(i);
    }
    run_frame_after_transform(ctx, f, ctx->ref_ms);
  }
  ctx->whole_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return worst_status(ctx);
}

ORACLE_API int hope_oracle_segment_cloud_xyz(hope_ctx* ctx, const void* xyz, size_t n, size_t stride,
                                             int apply_filter_transform, int /*on_device*/)
{
  if (!ctx || (!xyz && n > 0) || stride < 12 || (stride % 4) != 0) return HOPE_E_INVALID;
  const auto t0 = std::chrono::steady_clock::now();
  ctx->ref_ms = 0;
  ctx->frames.clear();
  ctx->frames.resize(1);
  Frame& f = ctx->frames[0];
  std::memset(&f.counts, 0, sizeof(f.counts));
  std::memset(&f.max_plane, 0, sizeof(f.max_plane));
  f.counts.n_in = (int64_t)n;
  const hope_config& cfg = ctx->cfg;
  const Affine a = make_affine(ctx->pose);
  const char* base = static_cast<const char*>(xyz);
  for (size_t i = 0; i < n; ++i) {
    float p[3];
    std::memcpy(p, base + i * stride, 12);
    if (apply_filter_transform) {
      /* PassThrough on z in the sensor frame, then doTransform (plane_segment.cpp:791-795) */
      if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
      if (p[2] < cfg.min_depth || p[2] > cfg.max_depth) continue;
      float ox, oy, oz;
      apply_affine(a, p[0], p[1], p[2], ox, oy, oz);
      p[0] = ox;
      p[1] = oy;
      p[2] = oz;
    }
    /* POINT_CLOUD mode aliases the input (plane_segment.cpp:118-120); non-finite points
     * are skipped later by VoxelGrid, so they never reach any result: drop them here */
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    f.tx.push_back(p[0]);
    f.ty.push_back(p[1]);
    f.tz.push_back(p[2]);
    f.src_index.push_back((int)i);
  }
  run_frame_after_transform(ctx, f, ctx->ref_ms);
  ctx->whole_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return worst_status(ctx);
}

/* ---- staged entry points for one cloud across several ranks (include/hope_gpu.h) ------------ */
namespace {
// A2/A3 of one record; false if the point is dropped
bool cloud_point(const hope_ctx* ctx, const Affine& a, const char* base, size_t stride, size_t i, int apply, float p[3])
{
  std::memcpy(p, base + i * stride, 12);
  if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) return false;
  if (apply) {
    if (p[2] < ctx->cfg.min_depth || p[2] > ctx->cfg.max_depth) return false;
    float ox, oy, oz;
    apply_affine(a, p[0], p[1], p[2], ox, oy, oz);
    p[0] = ox;
    p[1] = oy;
    p[2] = oz;
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) return false;
  }
  return true;
}
}  // namespace

ORACLE_API int hope_oracle_grid_from_bbox(hope_ctx* ctx, const float bbox[6], int32_t min_b[3], int32_t div_b[3])
{
  if (!ctx || !bbox || !min_b || !div_b) return HOPE_E_INVALID;
  const float inv[3] = {1.0f / ctx->cfg.th_xy, 1.0f / ctx->cfg.th_xy, 1.0f / ctx->cfg.th_z};
  int64_t dxyz = 1;
  for (int d = 0; d < 3; ++d) {
    dxyz *= static_cast<int64_t>((bbox[3 + d] - bbox[d]) * inv[d]) + 1;
    min_b[d] = static_cast<int>(std::floor(bbox[d] * inv[d]));
    div_b[d] = static_cast<int>(std::floor(bbox[3 + d] * inv[d])) - min_b[d] + 1;
  }
  return dxyz > static_cast<int64_t>(std::numeric_limits<int32_t>::max()) ? HOPE_E_RANGE : HOPE_OK;
}

ORACLE_API int hope_oracle_shard_bbox(hope_ctx* ctx, const void* xyz, size_t n, size_t stride, int apply, int,
                                      float bbox[6], int64_t* n_raw)
{
  if (!ctx || !bbox || !n_raw || (!xyz && n > 0)) return HOPE_E_INVALID;
  const Affine a = make_affine(ctx->pose);
  for (int d = 0; d < 3; ++d) {
    bbox[d] = std::numeric_limits<float>::infinity();
    bbox[3 + d] = -std::numeric_limits<float>::infinity();
  }
  int64_t cnt = 0;
  for (size_t i = 0; i < n; ++i) {
    float p[3];
    if (!cloud_point(ctx, a, static_cast<const char*>(xyz), stride, i, apply, p)) continue;
    ++cnt;
    for (int d = 0; d < 3; ++d) {
      bbox[d] = std::min(bbox[d], p[d]);
      bbox[3 + d] = std::max(bbox[3 + d], p[d]);
    }
  }
  *n_raw = cnt;
  return HOPE_OK;
}

ORACLE_API int hope_oracle_shard_z_histogram(hope_ctx* ctx, const void* xyz, size_t n, size_t stride, int apply, int,
                                             const float bbox[6], int64_t* hist, int n_bins)
{
  if (!ctx || !bbox || !hist || n_bins <= 0) return HOPE_E_INVALID;
  const Affine a = make_affine(ctx->pose);
  const float inv_z = 1.0f / ctx->cfg.th_z;
  const int min_bz = static_cast<int>(std::floor(bbox[2] * inv_z));
  for (int b = 0; b < n_bins; ++b) hist[b] = 0;
  for (size_t i = 0; i < n; ++i) {
    float p[3];
    if (!cloud_point(ctx, a, static_cast<const char*>(xyz), stride, i, apply, p)) continue;
    const int k = static_cast<int>(std::floor(p[2] * inv_z) - static_cast<float>(min_bz));
    if (k >= 0 && k < n_bins) hist[k] += 1;
  }
  return HOPE_OK;
}

ORACLE_API int hope_oracle_shard_route(hope_ctx* ctx, const void* xyz, size_t n, size_t stride, int apply, int,
                                       const float bbox[6], const int32_t* layer_owner, int n_bins, int n_ranks,
                                       float* out_xyz, int, int64_t* counts)
{
  if (!ctx || !bbox || !layer_owner || n_bins <= 0 || n_ranks <= 0 || !counts) return HOPE_E_INVALID;
  const Affine a = make_affine(ctx->pose);
  const float inv_z = 1.0f / ctx->cfg.th_z;
  const int min_bz = static_cast<int>(std::floor(bbox[2] * inv_z));
  std::vector<std::vector<float>> per(n_ranks);
  for (size_t i = 0; i < n; ++i) {
    float p[3];
    if (!cloud_point(ctx, a, static_cast<const char*>(xyz), stride, i, apply, p)) continue;
    const int k = static_cast<int>(std::floor(p[2] * inv_z) - static_cast<float>(min_bz));
    if (k < 0 || k >= n_bins) continue;
    std::vector<float>& v = per[layer_owner[k]];
    v.insert(v.end(), p, p + 3);
  }
  size_t o = 0;
  for (int r = 0; r < n_ranks; ++r) {
    counts[r] = (int64_t)(per[r].size() / 3);
    if (!per[r].empty()) std::memcpy(out_xyz + o, per[r].data(), per[r].size() * sizeof(float));
    o += per[r].size();
  }
  return HOPE_OK;
}

static int oracle_cloud_staged(hope_ctx* ctx, const void* xyz, size_t n, size_t stride, const float bbox[6], int stop,
                               int adopt, int64_t n_raw_total)
{
  ctx->forced_bbox = bbox;
  ctx->stop_after_voxel = stop;
  ctx->input_is_downsampled = adopt;
  const int st = hope_oracle_segment_cloud_xyz(ctx, xyz, n, stride, 0, 0);
  ctx->forced_bbox = nullptr;
  ctx->stop_after_voxel = 0;
  ctx->input_is_downsampled = 0;
  if (st >= 0 && n_raw_total >= 0 && !ctx->frames.empty()) ctx->frames[0].counts.n_raw = n_raw_total;
  return st;
}

ORACLE_API int hope_oracle_downsample_cloud_bbox(hope_ctx* ctx, const void* xyz, size_t n, size_t stride, int,
                                                 const float bbox[6])
{
  if (!ctx || !bbox) return HOPE_E_INVALID;
  return oracle_cloud_staged(ctx, xyz, n, stride, bbox, 1, 0, -1);
}

ORACLE_API int hope_oracle_segment_downsampled(hope_ctx* ctx, const void* vox, size_t n_vox, size_t stride, int,
                                               const float bbox[6], int64_t n_raw_total)
{
  if (!ctx || !bbox) return HOPE_E_INVALID;
  return oracle_cloud_staged(ctx, vox, n_vox, stride, bbox, 0, 1, n_raw_total);
}

// The checker's form of the staged call: it computes the whole result itself (own range and
// stop_after need nothing special: every accessor is valid afterwards) and CHECKS the arrays the
// ranks have gathered against its own, so a misplaced or stale piece fails the call.
ORACLE_API int hope_oracle_segment_downsampled_staged(hope_ctx* ctx, const void* vox, size_t n_vox, size_t stride, int,
                                                      const float bbox[6], int64_t n_raw_total, int64_t own_first,
                                                      int64_t own_count, const float* normals_all,
                                                      const int32_t* knn_all, int stop_after)
{
  if (!ctx || !bbox || own_first < 0 || own_count < 0 || (size_t)(own_first + own_count) > n_vox || stop_after < 0 ||
      stop_after > 2)
    return HOPE_E_INVALID;
  // (k-NN rows that are all -1 mark the points of other z bands: the regions then grow inside the own band
  // only; every other row is checked against the oracle's own)
  ctx->band_knn = knn_all;
  ctx->knn_mismatch = -1;
  const int st = oracle_cloud_staged(ctx, vox, n_vox, stride, bbox, 0, 1, n_raw_total);
  ctx->band_knn = nullptr;
  if (st < 0 || ctx->frames.empty()) return st;
  const Frame& f = ctx->frames[0];
  if (normals_all) {
    for (size_t i = 0; i < f.nx.size(); ++i) {
      const float mine[4] = {f.nx[i], f.ny[i], f.nz[i], f.curv[i]};
      if (std::memcmp(mine, normals_all + 4 * i, sizeof(mine)) != 0) {
        ctx->err = "staged call: gathered normals differ from the oracle's at point " + std::to_string(i);
        return HOPE_E_INVALID;
      }
    }
  }
  if (knn_all && ctx->knn_mismatch >= 0) {
    ctx->err = "staged call: gathered k-NN rows differ from the oracle's at entry " + std::to_string(ctx->knn_mismatch);
    return HOPE_E_INVALID;
  }
  return st;
}

#define GET_FRAME(ctx, frame)                                            \
  if (!(ctx)) return HOPE_E_INVALID;                                     \
  if ((frame) < 0 || (frame) >= (int)(ctx)->frames.size()) return HOPE_E_RANGE; \
  Frame& f = (ctx)->frames[(frame)];

/* ------------------------------------------------------------------------------------
 * F1: Utilities::getClustersUponPlane (utilities.cpp:1363-1408).
 *  - getCloudZInfo of the contour (utilities.cpp:751-769): z_max over finite z, initial -1000;
 *  - getStraightRect2D (utilities.cpp:1538-1560): pcl::getMinMax3D of the contour, centre and extent in float;
 *  - per point of src_mono_cloud_: skip when `z < z_max + 0.01` (float promoted, double compare), skip when
 *    `fabs(x - center.x) > width * 0.5` (or y / height), keep when isInContour;
 *  - isInContour (utilities.cpp:1337-1360): sum over hull edges of |atan2(|pvi x pvj|, pvi . pvj)| in float
 *    (Eigen Vector3f with z = 0: cross = (0, 0, ax*by - ay*bx), norm = sqrt of its square, dot = ax*bx + ay*by),
 *    inside iff fabs(sum - 2*M_PI) < 0.01 in double.  atan2f through hope_detmath.h like every other
 *    transcendental of the path;
 *  - pcl::EuclideanClusterExtraction (PCL 1.8.1 segmentation/impl/extract_clusters.hpp): for each unprocessed
 *    point in index order a breadth-first flood over radiusSearch(tolerance) (FLANN: dist2 < float(tol*tol) with
 *    tol = (double)(float)0.01), size gate [min, max], indices sorted ascending, and finally
 *    std::sort(clusters.rbegin(), clusters.rend(), comparePointClusters) -- called literally below (libstdc++).
 * ---------------------------------------------------------------------------------- */
namespace {
bool is_in_contour(const float* cx, const float* cy, int nc, float px, float py)
{
  float angle_sum = 0;
  for (int i = 0; i < nc; ++i) {
    const int j = i + 1 < nc ? i + 1 : 0;
    const float ax = cx[i] - px, ay = cy[i] - py, bx = cx[j] - px, by = cy[j] - py;
    const float cz = ax * by - ay * bx;
    const float nrm = std::sqrt(cz * cz);
    const float dot = ax * bx + ay * by;
    const float angle = o_atan2f(nrm, dot);
    angle_sum += std::fabs(angle);
  }
  return std::fabs((double)angle_sum - 2 * M_PI) < 0.01;
}
struct UponCluster { std::vector<int> indices; };
bool compare_point_clusters(const UponCluster& a, const UponCluster& b) { return a.indices.size() < b.indices.size(); }
}  // namespace

ORACLE_API int hope_oracle_clusters_upon_plane(hope_ctx* ctx, int frame, const float* contour, int n_contour, float tolerance,
                                               int min_size, int max_size, int32_t* n_clusters, int64_t* n_upon)
{
  GET_FRAME(ctx, frame);
  if (!contour || n_contour < 0 || !n_clusters) return HOPE_E_INVALID;
  ctx->upon_clusters.clear();
  ctx->upon_x.clear(); ctx->upon_y.clear(); ctx->upon_z.clear(); ctx->upon_src.clear();
  *n_clusters = 0;
  if (n_upon) *n_upon = 0;
  std::vector<float> cx(n_contour), cy(n_contour);
  float z_max = -1000.0f;
  float mnx = FLT_MAX, mny = FLT_MAX, mxx = -FLT_MAX, mxy = -FLT_MAX;
  for (int i = 0; i < n_contour; ++i) {
    const float x = contour[3 * i], y = contour[3 * i + 1], z = contour[3 * i + 2];
    cx[i] = x; cy[i] = y;
    if (std::isfinite(z) && z > z_max) z_max = z;
    if (std::isfinite(x) && std::isfinite(y) && std::isfinite(z)) {
      mnx = std::min(mnx, x); mny = std::min(mny, y); mxx = std::max(mxx, x); mxy = std::max(mxy, y);
    }
  }
  const float center_x = (mxx + mnx) * 0.5f, center_y = (mxy + mny) * 0.5f;
  const float width = mxx - mnx, height = mxy - mny;
  for (size_t i = 0; i < f.tx.size(); ++i) {
    if ((double)f.tz[i] < (double)z_max + 0.01) continue;
    if ((double)std::fabs(f.tx[i] - center_x) > (double)width * 0.5 || (double)std::fabs(f.ty[i] - center_y) > (double)height * 0.5) continue;
    if (!is_in_contour(cx.data(), cy.data(), n_contour, f.tx[i], f.ty[i])) continue;
    ctx->upon_x.push_back(f.tx[i]); ctx->upon_y.push_back(f.ty[i]); ctx->upon_z.push_back(f.tz[i]);
    ctx->upon_src.push_back(f.src_index[i]);
  }
  const int n = (int)ctx->upon_x.size();
  if (n_upon) *n_upon = n;
  if (n == 0) return HOPE_OK;   // initCompute fails on an empty cloud: no clusters
  KdTree tree;
  tree.init(ctx->upon_x.data(), ctx->upon_y.data(), ctx->upon_z.data(), n);
  const double tol = (double)tolerance;
  const float r2 = (float)(tol * tol);
  std::vector<char> processed(n, 0);
  std::vector<UponCluster> clusters;
  std::vector<KdTree::DI> nn;
  for (int i = 0; i < n; ++i) {
    if (processed[i]) continue;
    std::vector<int> seed_queue;
    size_t sq_idx = 0;
    seed_queue.push_back(i);
    processed[i] = 1;
    while (sq_idx < seed_queue.size()) {
      const int q = seed_queue[sq_idx];
      nn.clear();
      tree.radius_rec(0, ctx->upon_x[q], ctx->upon_y[q], ctx->upon_z[q], r2, nn);
      std::sort(nn.begin(), nn.end());
      for (const KdTree::DI& e : nn) {
        if (processed[e.second]) continue;
        seed_queue.push_back(e.second);
        processed[e.second] = 1;
      }
      ++sq_idx;
    }
    if ((long long)seed_queue.size() >= min_size && (long long)seed_queue.size() <= max_size) {
      UponCluster r;
      r.indices = seed_queue;
      std::sort(r.indices.begin(), r.indices.end());
      r.indices.erase(std::unique(r.indices.begin(), r.indices.end()), r.indices.end());
      clusters.push_back(r);
    }
  }
  std::sort(clusters.rbegin(), clusters.rend(), compare_point_clusters);
  for (UponCluster& c : clusters) ctx->upon_clusters.push_back(std::move(c.indices));
  *n_clusters = (int32_t)ctx->upon_clusters.size();
  return HOPE_OK;
}

ORACLE_API int hope_oracle_get_upon_cluster_sizes(hope_ctx* ctx, int32_t* sizes, int cap)
{
  if (!ctx || !sizes) return HOPE_E_INVALID;
  int n = 0;
  for (; n < (int)ctx->upon_clusters.size() && n < cap; ++n) sizes[n] = (int32_t)ctx->upon_clusters[n].size();
  return n;
}

ORACLE_API int hope_oracle_get_upon_cluster(hope_ctx* ctx, int cluster, float* xyz, int32_t* src_index, int cap)
{
  if (!ctx || !xyz) return HOPE_E_INVALID;
  if (cluster < 0 || cluster >= (int)ctx->upon_clusters.size()) { ctx->err = "cluster index out of range"; return HOPE_E_RANGE; }
  const std::vector<int>& idx = ctx->upon_clusters[cluster];
  if (cap < (int)idx.size()) { ctx->err = "buffer too small"; return HOPE_E_RANGE; }
  for (size_t k = 0; k < idx.size(); ++k) {
    xyz[3 * k] = ctx->upon_x[idx[k]]; xyz[3 * k + 1] = ctx->upon_y[idx[k]]; xyz[3 * k + 2] = ctx->upon_z[idx[k]];
    if (src_index) src_index[k] = ctx->upon_src[idx[k]];
  }
  return (int)idx.size();
}

ORACLE_API int hope_oracle_num_frames(const hope_ctx* ctx) { return ctx ? (int)ctx->frames.size() : HOPE_E_INVALID; }

ORACLE_API int hope_oracle_frame_counts(hope_ctx* ctx, int frame, hope_counts* out)
{
  GET_FRAME(ctx, frame);
  if (!out) return HOPE_E_INVALID;
  *out = f.counts;
  return HOPE_OK;
}

ORACLE_API int hope_oracle_get_transformed(hope_ctx* ctx, int frame, float* xyz, int32_t* src)
{
  GET_FRAME(ctx, frame);
  for (size_t i = 0; i < f.tx.size(); ++i) {
    if (xyz) {
      xyz[3 * i] = f.tx[i];
      xyz[3 * i + 1] = f.ty[i];
      xyz[3 * i + 2] = f.tz[i];
    }
    if (src) src[i] = f.src_index[i];
  }
  return HOPE_OK;
}

ORACLE_API int hope_oracle_get_downsampled(hope_ctx* ctx, int frame, float* xyz)
{
  GET_FRAME(ctx, frame);
  if (!xyz) return HOPE_E_INVALID;
  for (size_t i = 0; i < f.vx.size(); ++i) {
    xyz[3 * i] = f.vx[i];
    xyz[3 * i + 1] = f.vy[i];
    xyz[3 * i + 2] = f.vz[i];
  }
  return HOPE_OK;
}

ORACLE_API int hope_oracle_get_normals(hope_ctx* ctx, int frame, float* o)
{
  GET_FRAME(ctx, frame);
  if (!o) return HOPE_E_INVALID;
  for (size_t i = 0; i < f.nx.size(); ++i) {
    o[4 * i] = f.nx[i];
    o[4 * i + 1] = f.ny[i];
    o[4 * i + 2] = f.nz[i];
    o[4 * i + 3] = f.curv[i];
  }
  return HOPE_OK;
}

ORACLE_API int hope_oracle_get_downsampled_to(hope_ctx* ctx, int frame, float* xyz, int)
{
  return hope_oracle_get_downsampled(ctx, frame, xyz);
}

ORACLE_API int hope_oracle_get_normals_rows(hope_ctx* ctx, int frame, int64_t first, int64_t count, float* o, int)
{
  GET_FRAME(ctx, frame);
  if (first < 0 || count < 0 || (size_t)(first + count) > f.nx.size()) return HOPE_E_RANGE;
  if (count > 0 && !o) return HOPE_E_INVALID;
  for (int64_t i = 0; i < count; ++i) {
    o[4 * i] = f.nx[first + i];
    o[4 * i + 1] = f.ny[first + i];
    o[4 * i + 2] = f.nz[first + i];
    o[4 * i + 3] = f.curv[first + i];
  }
  return HOPE_OK;
}

ORACLE_API int hope_oracle_get_knn_rows(hope_ctx* ctx, int frame, int64_t first, int64_t count, int32_t* o, int)
{
  GET_FRAME(ctx, frame);
  const int64_t K = ctx->cfg.knn;
  if (first < 0 || count < 0 || (size_t)((first + count) * K) > f.knn.size()) return HOPE_E_RANGE;
  if (count > 0 && !o) return HOPE_E_INVALID;
  std::copy(f.knn.begin() + first * K, f.knn.begin() + (first + count) * K, o);
  return HOPE_OK;
}

// Edges growRegion would follow (validatePoint, z_growing.cpp:335-364: |dz| < th_z) from the fit points
// [fit_first, fit_first + fit_count) to fit points outside that range (multi-GPU: are the z bands independent?)
ORACLE_API int hope_oracle_count_band_crossings(hope_ctx* ctx, int frame, int64_t fit_first, int64_t fit_count,
                                                int64_t* n_cross)
{
  GET_FRAME(ctx, frame);
  const int64_t K = ctx->cfg.knn, n_fit = (int64_t)f.fz.size();
  if (!n_cross || fit_first < 0 || fit_count < 0 || fit_first + fit_count > n_fit) return HOPE_E_INVALID;
  *n_cross = 0;
  if (fit_count == 0) return HOPE_OK;
  if ((int64_t)f.knn.size() < (fit_first + fit_count) * K) return HOPE_E_INVALID;
  for (int64_t g = fit_first; g < fit_first + fit_count; ++g)
    for (int64_t j = 0; j < K; ++j) {
      const int v = f.knn[(size_t)(g * K + j)];
      if (v < 0 || (v >= fit_first && v < fit_first + fit_count)) continue;
      if (std::fabs(f.fz[g] - f.fz[v]) < ctx->cfg.th_z) ++*n_cross;
    }
  return HOPE_OK;
}

ORACLE_API int hope_oracle_get_fit_indices(hope_ctx* ctx, int frame, int32_t* idx)
{
  GET_FRAME(ctx, frame);
  if (!idx) return HOPE_E_INVALID;
  std::copy(f.fit_idx.begin(), f.fit_idx.end(), idx);
  return HOPE_OK;
}

ORACLE_API int hope_oracle_get_knn(hope_ctx* ctx, int frame, int32_t* nbr)
{
  GET_FRAME(ctx, frame);
  if (!nbr) return HOPE_E_INVALID;
  std::copy(f.knn.begin(), f.knn.end(), nbr);
  return HOPE_OK;
}

ORACLE_API int hope_oracle_get_labels(hope_ctx* ctx, int frame, int32_t* labels)
{
  GET_FRAME(ctx, frame);
  if (!labels) return HOPE_E_INVALID;
  std::copy(f.labels.begin(), f.labels.end(), labels);
  return HOPE_OK;
}

ORACLE_API int hope_oracle_get_clusters(hope_ctx* ctx, int frame, hope_cluster* out, int cap)
{
  GET_FRAME(ctx, frame);
  if (!out && cap > 0) return HOPE_E_INVALID;
  const int n = std::min<int>(cap, (int)f.clusters.size());
  for (int i = 0; i < n; ++i) {
    const Cluster& cl = f.clusters[i];
    out[i].size = (int)cl.indices.size();
    out[i].pass = cl.pass;
    out[i].z_mean = cl.z_mean;
    out[i].min_x = cl.min_x;
    out[i].min_y = cl.min_y;
    out[i].max_x = cl.max_x;
    out[i].max_y = cl.max_y;
    out[i].seed_rank = cl.seed_rank;
  }
  return n;
}

ORACLE_API int hope_oracle_get_cluster_indices(hope_ctx* ctx, int frame, int cluster, int32_t* idx, int cap)
{
  GET_FRAME(ctx, frame);
  if (cluster < 0 || cluster >= (int)f.clusters.size()) return HOPE_E_RANGE;
  const std::vector<int>& v = f.clusters[cluster].indices;
  if ((int)v.size() > cap || !idx) return HOPE_E_RANGE;
  std::copy(v.begin(), v.end(), idx);
  return (int)v.size();
}

// F2: the reference hulls EVERY point of a plane (utilities.cpp:1531-1536, plane_segment.cpp:391-406), so here
// every member is a candidate; the device library's pre-filter has no counterpart in the reference.
ORACLE_API int hope_oracle_get_cluster_hull_candidates(hope_ctx* ctx, int frame, int cluster, int32_t* pos, int cap)
{
  GET_FRAME(ctx, frame);
  if (cluster < 0 || cluster >= (int)f.clusters.size()) return HOPE_E_RANGE;
  if (cap < 0 || (cap > 0 && !pos)) return HOPE_E_INVALID;
  const int n = (int)f.clusters[cluster].indices.size();
  for (int i = 0; i < n && i < cap; ++i) pos[i] = i;
  return n;
}

ORACLE_API int hope_oracle_get_max_plane(hope_ctx* ctx, int frame, hope_max_plane* out)
{
  GET_FRAME(ctx, frame);
  if (!out) return HOPE_E_INVALID;
  *out = f.max_plane;
  return HOPE_OK;
}

ORACLE_API int hope_oracle_get_max_plane_members(hope_ctx* ctx, int frame, int32_t* out, int cap)
{
  GET_FRAME(ctx, frame);
  if (!out && cap > 0) return HOPE_E_INVALID;
  const int n = std::min<int>(cap < 0 ? 0 : cap, (int)f.max_members.size());
  for (int i = 0; i < n; ++i) out[i] = f.max_members[i];
  return (int)f.max_members.size();
}

ORACLE_API int hope_oracle_last_timing(hope_ctx* ctx, double* whole_ms, double* ref_ms)
{
  if (!ctx) return HOPE_E_INVALID;
  if (whole_ms) *whole_ms = ctx->whole_ms;
  if (ref_ms) *ref_ms = ctx->ref_ms;
  return HOPE_OK;
}

/* test hook: the seed permutation std::sort leaves on n all-equal keys
 * (z_growing.cpp:235-246) */
ORACLE_API int hope_oracle_seed_order(int n, int32_t* out)
{
  if (n < 0 || (n > 0 && !out)) return HOPE_E_INVALID;
  std::vector<std::pair<float, int>> pr(n, std::pair<float, int>());
  for (int i = 0; i < n; ++i) pr[i].second = i;
  std::sort(pr.begin(), pr.end(), compare_pair);
  for (int i = 0; i < n; ++i) out[i] = pr[i].second;
  return HOPE_OK;
}

/* test hooks for include/hope_detmath.h against glibc */
ORACLE_API void hope_oracle_detmath(int which, int n, const float* a, const float* b, float* det, float* libm)
{
  for (int i = 0; i < n; ++i) {
    switch (which) {
      case 0: det[i] = hope_det_atan2f(a[i], b[i]); libm[i] = std::atan2(a[i], b[i]); break;
      case 1: det[i] = hope_det_cosf(a[i]); libm[i] = std::cos(a[i]); break;
      case 2: det[i] = hope_det_sinf(a[i]); libm[i] = std::sin(a[i]); break;
      case 3: det[i] = hope_det_asinf(a[i]); libm[i] = std::asin(a[i]); break;
      default: det[i] = hope_det_acosf(a[i]); libm[i] = std::acos(a[i]); break;
    }
  }
}
