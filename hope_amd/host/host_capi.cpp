// C entry points over the host-side pieces that need no device (format readers, hull, plane IDs), so the
// Python tests and tools can call them with ctypes.  Plain pointers and sizes only.
#include <cstring>
#include <string>

#include "formats.hpp"
#include "hull2d.hpp"
#include "plane_id.hpp"

#define HOST_API extern "C" __attribute__((visibility("default")))

namespace {
thread_local std::string g_err;
int fail(const std::string& e) { g_err = e; return -1; }
}  // namespace

HOST_API const char* hope_host_last_error(void) { return g_err.c_str(); }

// cv::imread(path, -1) for PNG.  First call with out == NULL to get the geometry; sample size is bit_depth/8.
HOST_API int hope_host_read_png(const char* path, void* out, size_t cap_bytes, int* width, int* height, int* channels, int* bit_depth)
{
  hope::Image im; std::string err;
  if (!hope::readPng(path, im, err)) return fail(err);
  *width = im.width; *height = im.height; *channels = im.channels; *bit_depth = im.bit_depth;
  const size_t need = im.bit_depth == 16 ? im.u16.size() * 2 : im.u8.size();
  if (!out) return 0;
  if (cap_bytes < need) return fail("buffer too small");
  std::memcpy(out, im.bit_depth == 16 ? (const void*)im.u16.data() : (const void*)im.u8.data(), need);
  return 0;
}

// kind: "ply" or "pcd".  Returns the point count; fills up to cap points of {x,y,z} floats and {r,g,b} bytes.
HOST_API long long hope_host_read_cloud(const char* path, const char* kind, float* xyz, unsigned char* rgb, size_t cap)
{
  hope::PointCloud c; std::string err;
  const bool ok = std::strcmp(kind, "ply") == 0 ? hope::readPly(path, c, err) : std::strcmp(kind, "pcd") == 0 ? hope::readPcd(path, c, err)
                                                                                                              : (err = "Unrecognized file format.", false);
  if (!ok) return fail(err);
  for (size_t i = 0; i < c.size() && i < cap; ++i) {
    if (xyz) { xyz[3 * i] = c.points[i].x; xyz[3 * i + 1] = c.points[i].y; xyz[3 * i + 2] = c.points[i].z; }
    if (rgb) { rgb[3 * i] = c.points[i].r; rgb[3 * i + 1] = c.points[i].g; rgb[3 * i + 2] = c.points[i].b; }
  }
  return (long long)c.size();
}

// TUM all.txt: returns the number of entries; for the first cap of them the 7 pose floats (tx ty tz qx qy qz qw).
HOST_API int hope_host_read_tum_list(const char* path, float* pose7, int cap)
{
  std::vector<hope::TumEntry> l; std::string err;
  if (!hope::readTumList(path, l, err)) return fail(err);
  for (int i = 0; i < (int)l.size() && i < cap; ++i) { std::memcpy(pose7 + 7 * i, l[i].t, 12); std::memcpy(pose7 + 7 * i + 3, l[i].q, 16); }
  return (int)l.size();
}

// Utilities::computeHull on n points of stride_bytes: hull vertex indices in output order, polygon area.
HOST_API int hope_host_hull(const void* xyz, size_t n, size_t stride_bytes, int* idx_out, int cap, double* area)
{
  hope::PointCloudMono c, h;
  c.points.resize(n);
  for (size_t i = 0; i < n; ++i) { const float* p = (const float*)((const char*)xyz + i * stride_bytes); c.points[i] = hope::PointXYZ(p[0], p[1], p[2]); }
  std::vector<int> idx;
  const double a = hope::computeHull(c, h, &idx);
  if (area) *area = a;
  for (int i = 0; i < (int)idx.size() && i < cap; ++i) idx_out[i] = idx[i];
  return (int)idx.size();
}

// Utilities::matchID on row-major float matrices (n x dim).
HOST_API int hope_host_match_id(const float* global, int n_global, const float* local, int n_local, int dim, const int* ids_in, int* ids_out)
{
  std::vector<std::vector<float>> g(n_global, std::vector<float>(dim)), l(n_local, std::vector<float>(dim));
  for (int i = 0; i < n_global; ++i) for (int c = 0; c < dim; ++c) g[i][c] = global[i * dim + c];
  for (int i = 0; i < n_local; ++i) for (int c = 0; c < dim; ++c) l[i][c] = local[i * dim + c];
  std::vector<int> in(ids_in, ids_in + n_global), out;
  hope::matchID(g, l, in, out, dim);
  for (int i = 0; i < n_local; ++i) ids_out[i] = out[i];
  return 0;
}

// sensor_msgs/PointCloud2 -> packed xyz; field table as parallel arrays.
HOST_API int hope_host_from_pointcloud2(const unsigned char* data, size_t n_bytes, unsigned width, unsigned height, unsigned point_step,
                                        unsigned row_step, int n_fields, const char* const* names, const unsigned* offsets,
                                        const unsigned char* datatypes, int is_bigendian, float* xyz_out)
{
  std::vector<hope::PointField> f;
  for (int i = 0; i < n_fields; ++i) f.push_back({names[i], offsets[i], datatypes[i], 1});
  hope::PointCloudMono c; std::string err;
  if (!hope::fromPointCloud2(data, n_bytes, width, height, point_step, row_step, f, is_bigendian != 0, false, c, err)) return fail(err);
  for (size_t i = 0; i < c.size(); ++i) { xyz_out[3 * i] = c.points[i].x; xyz_out[3 * i + 1] = c.points[i].y; xyz_out[3 * i + 2] = c.points[i].z; }
  return 0;
}
