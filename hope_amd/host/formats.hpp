// On-disk / on-wire formats either side of the path, without OpenCV or
// PCL: what hope_node reads with cv::imread (hope_node.cpp:204-205,234-235), phaseInput (:64-98),
// pcl::io::loadPCDFile / pcl::PLYReader (:176-185) and what pcl::fromROSMsg unpacks
// (plane_segment.cpp:787-789).  Every reader returns false and fills `err` instead of throwing.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "hope_cloud.hpp"

namespace hope {

struct Image {
  int width = 0, height = 0, channels = 0, bit_depth = 0;   // channels as stored in the file (1,2,3,4)
  std::vector<uint8_t> u8;     // bit_depth <= 8 (palette and low-bit grey expanded to 8 bits), row-major, interleaved
  std::vector<uint16_t> u16;   // bit_depth == 16, host byte order
};

// PNG: colour types 0/2/3/4/6, bit depths 1..16, filters 0-4, Adam7 interlace not supported (TUM has none).
bool readPng(const std::string& path, Image& out, std::string& err);
bool decodePng(const uint8_t* data, size_t n, Image& out, std::string& err);
// cv::imread(path, -1) of a TUM depth image: CV_16UC1 -> w*h uint16
bool readDepthPng(const std::string& path, std::vector<uint16_t>& depth, int& w, int& h, std::string& err);

// TUM association list "all.txt": `stamp rgb_path depth_path tx ty tz qx qy qz qw` per line
// (README.md:62-78, phaseInput hope_node.cpp:64-98).
struct TumEntry { float stamp; std::string rgb, depth; float t[3]; float q[4]; };
bool readTumList(const std::string& path, std::vector<TumEntry>& out, std::string& err);

// PLY (ascii, binary_little_endian, binary_big_endian) and PCD (ascii, binary, binary_compressed) with
// x y z and optional colour (PLY red/green/blue[/alpha] or diffuse_*; PCD rgb / rgba packed field).
bool readPly(const std::string& path, PointCloud& out, std::string& err);
bool readPcd(const std::string& path, PointCloud& out, std::string& err);
bool writePcdBinary(const std::string& path, const PointCloud& cloud, std::string& err);

// sensor_msgs/PointCloud2 payload -> PointCloudMono, the job of pcl::fromROSMsg in cloudCallback.
struct PointField { std::string name; uint32_t offset; uint8_t datatype; uint32_t count; };  // datatype: 7 = FLOAT32, 8 = FLOAT64
bool fromPointCloud2(const uint8_t* data, size_t n_bytes, uint32_t width, uint32_t height, uint32_t point_step,
                     uint32_t row_step, const std::vector<PointField>& fields, bool is_bigendian, bool is_dense,
                     PointCloudMono& out, std::string& err);

}  // namespace hope
