// hope_node: the reference's benchmark executable (src/hope_node.cpp) over the MI355X library.
//   hope_node {TUM_DATASET_FOLDER}                                        TUM list  (all.txt, README.md:62-87)
//   hope_node {POINT_CLOUD_FILE} {ply|pcd}                                point cloud file (README.md:100)
//   hope_node {ROLL} {PITCH} {YAW}                                        synthetic object (needs ROS; refused here)
//   hope_node {PREFIX} {RGB} {DEPTH} {TX} {TY} {TZ} {QX} {QY} {QZ} {QW}   single TUM pair (README.md:51)
// The argument COUNT picks the mode as upstream documents it (hope_node.cpp:49-62); upstream's 1/2/3-argument
// branches read argv one slot too far (:118,126-127,132-134), which is not reproduced.  There is no viewer:
// the planes are printed, and when the environment variable HOPE_NODE_DUMP names a file a JSON record per
// frame is appended to it (tests read that).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <fstream>
#include <future>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "formats.hpp"
#include "plane_id.hpp"
#include "plane_segment.hpp"
#ifdef HOPE_NODE_MULTI_GPU
#include "rccl_comm.hpp"
#endif

using namespace hope;

namespace {

void printHelp()
{
  std::cout << "hope_node VARIABLES\n"
               "VARIABLES number could be 1, 2, 3, or 10:\n"
               "1 VARIABLE, test a sequence on TUM dataset, {TUM_DATASET_FOLDER}\n"
               "Example: ./hope_node /path/to/TUM/rgbd_dataset_freiburg1_desk\n\n"
               "2 VARIABLES, test on point cloud file, {POINT_CLOUD_FILE} {TYPE}, TYPE could be ply or pcd\n"
               "Example: ./hope_node ~/loft.ply ply\n\n"
               "3 VARIABLES, test with the synthetic object, set the pose of the object with {ROLL} {PITCH} {YAW}\n"
               "Example: ./hope_node 0 0 0\n\n"
               "10 VARIABLES, test on single TUM pair, {TUM_DATASET_FOLDER} {RGB_IMG} {DEPTH_IMG} {TX} {TY} {TZ} {QX} {QY} {QZ} {QW}\n"
               "Example: ./hope_node ~/TUM/rgbd_dataset_freiburg1_desk/ rgb/1305031459.259760.png depth/1305031459.274941.png "
               "-0.2171 -0.0799 1.3959 -0.8445 -0.0451 0.0954 0.5251"
            << std::endl;
}

void report(PlaneSegment& hope, const std::string& name, std::ofstream* dump)
{
  const hope_counts& c = hope.last_counts();
  std::printf("[%s] raw %lld  down-sampled %lld  normal-fit %lld  clusters %d  planes %zu\n", name.c_str(), (long long)c.n_raw,
              (long long)c.n_vox, (long long)c.n_fit, c.n_clusters, hope.plane_coeff_.size());
  for (size_t i = 0; i < hope.plane_coeff_.size(); ++i) {
    const std::vector<float>& f = hope.plane_coeff_[i];
    std::printf("  plane id %3d  z %.4f  x [%.3f, %.3f]  y [%.3f, %.3f]  points %zu", hope.plane_ids()[i], f[0], f[1], f[3], f[2], f[4],
                hope.plane_points_[i]->size());
    if (hope.cal_hull_) std::printf("  hull %zu vertices, area %.4f m^2", hope.plane_hull_[i]->size(), hope.plane_area_[i]);
    std::printf("\n");
  }
  if (dump) {
    *dump << "{\"frame\": \"" << name << "\", \"n_raw\": " << c.n_raw << ", \"n_vox\": " << c.n_vox << ", \"n_fit\": " << c.n_fit
          << ", \"n_clusters\": " << c.n_clusters << ", \"status\": " << hope.last_status() << ", \"planes\": [";
    char buf[256];
    for (size_t i = 0; i < hope.plane_coeff_.size(); ++i) {
      const std::vector<float>& f = hope.plane_coeff_[i];
      std::snprintf(buf, sizeof buf, "%s{\"id\": %d, \"coeff\": [%.9g, %.9g, %.9g, %.9g, %.9g], \"points\": %zu, \"hull\": %zu, \"area\": %.9g}",
                    i ? ", " : "", hope.plane_ids()[i], f[0], f[1], f[2], f[3], f[4], hope.plane_points_[i]->size(),
                    hope.cal_hull_ ? hope.plane_hull_[i]->size() : (size_t)0, hope.cal_hull_ ? hope.plane_area_[i] : 0.f);
      *dump << buf;
    }
    *dump << "]}" << std::endl;
  }
}

}  // namespace

int main(int argc, char** argv)
{
  data_type type;
  std::string path_prefix, path_rgb, path_depth, path_list_all, path_cloud, cloud_type;
  float roll = 0.f, pitch = 0.f, yaw = 0.f;
  float tx = 0.f, ty = 0.f, tz = 0.f, qx = 0.f, qy = 0.f, qz = 0.f, qw = 0.f;   // file-scope zeros upstream (hope_node.cpp:47)

  if (argc == 2) {
    type = TUM_LIST;
    path_prefix = argv[1];
    path_list_all = path_prefix + "/all.txt";
    std::cout << "Using the image list from TUM RGB-D SLAM dataset." << std::endl;
  } else if (argc == 3) {
    type = POINT_CLOUD;
    path_cloud = argv[1];
    cloud_type = argv[2];
    std::cout << "Using point cloud file " << path_cloud << std::endl;
  } else if (argc == 4) {
    type = SYN;
    roll = (float)std::atof(argv[1]); pitch = (float)std::atof(argv[2]); yaw = (float)std::atof(argv[3]);
    std::cout << "Using synthetic object." << std::endl;
  } else if (argc == 11) {
    type = TUM_SINGLE;
    int a = 1;
    path_prefix = argv[a++];
    path_rgb = path_prefix + argv[a++];       // plain concatenation, hope_node.cpp:141-142
    path_depth = path_prefix + argv[a++];
    tx = (float)std::atof(argv[a++]); ty = (float)std::atof(argv[a++]); tz = (float)std::atof(argv[a++]);
    qx = (float)std::atof(argv[a++]); qy = (float)std::atof(argv[a++]); qz = (float)std::atof(argv[a++]); qw = (float)std::atof(argv[a++]);
    std::cout << "Using single pair of rgb and depth image of TUM RGB-D SLAM dataset." << std::endl;
  } else {
    printHelp();
    return -1;
  }

  const float xy_resolution = 0.05f, z_resolution = 0.02f;   // hope_node.cpp:160-161
  std::cout << "Using threshold: xy@" << xy_resolution << " z@" << z_resolution << std::endl;

  std::ofstream dump_file;
  std::ofstream* dump = nullptr;
  if (const char* d = std::getenv("HOPE_NODE_DUMP")) { dump_file.open(d, std::ios::app); if (dump_file) dump = &dump_file; }

  try {
    PlaneSegment hope(type, xy_resolution, z_resolution);
    hope.cal_hull_ = std::getenv("HOPE_NODE_HULL") != nullptr;
    std::string err;

    if (type == SYN) {
      hope.setRPY(roll, pitch, yaw);                                                      // hope_node.cpp:169
      // Upstream the synthetic object's cloud arrives on a ROS topic (plane_segment.cpp:604-622).  Without ROS the
      // message is a .pcd/.ply file named by HOPE_NODE_SYN_CLOUD, handed to the same cloudCallback.
      const char* syn = std::getenv("HOPE_NODE_SYN_CLOUD");
      PointCloud::Ptr none(new PointCloud);
      if (!syn) {
        hope.getHorizontalPlanes(none);          // refuses: no cloud source
        return 2;
      }
      PointCloud::Ptr msg(new PointCloud);
      const std::string path = syn;
      const bool is_ply = path.size() > 4 && path.compare(path.size() - 4, 4, ".ply") == 0;
      if (!(is_ply ? readPly(path, *msg, err) : readPcd(path, *msg, err))) {
        std::cerr << "Couldn't read " << path << ". (" << err << ")" << std::endl;
        return -1;
      }
      hope.cloudCallback(msg);
      hope.getHorizontalPlanes(none);
      if (hope.last_status() < 0) return 1;
      report(hope, path, dump);
      std::printf("[TIMER]: total: %.3f ms (whole call %.3f ms)\n", hope.last_timer_window_ms(), hope.last_ms());
      return 0;
    }
    if (type == POINT_CLOUD) {
      PointCloud::Ptr src_cloud(new PointCloud);
      if (cloud_type == "pcd") {
        if (!readPcd(path_cloud, *src_cloud, err)) { std::cerr << "Couldn't read pcd file. (" << err << ")" << std::endl; return -1; }
      } else if (cloud_type == "ply") {
        if (!readPly(path_cloud, *src_cloud, err)) { std::cerr << "Couldn't read ply file. (" << err << ")" << std::endl; return -1; }
      } else { std::cerr << "Unrecognized file format." << std::endl; return -1; }
#ifdef HOPE_NODE_MULTI_GPU
      // HOPE_NODE_GPUS=N: the cloud in N contiguous ranges, one thread, ctx and RCCL communicator per GPU
      // (rccl_comm.cpp / multi_gpu.cpp; exchanges as in DESIGN.md section 7).  No counterpart upstream.
      const char* on0 = std::getenv("HOPE_NODE_RANKS_ON_GPU0");   // (one-GPU boxes: N ranks as threads on device 0, no RCCL)
      const char* g = on0 ? on0 : std::getenv("HOPE_NODE_GPUS");
      if (g) {
        const int n_gpus = std::atoi(g);
        hope_config cfg;
        hope_config_default(&cfg);
        cfg.th_xy = xy_resolution; cfg.th_z = z_resolution; cfg.mode = HOPE_MODE_OFFLINE;
        std::vector<float> xyz(3 * src_cloud->size());
        for (size_t i = 0; i < src_cloud->size(); ++i) {
          xyz[3 * i] = src_cloud->points[i].x; xyz[3 * i + 1] = src_cloud->points[i].y; xyz[3 * i + 2] = src_cloud->points[i].z;
        }
        MultiResult m;
        const int st = on0 ? run_cloud_as_ranks_on_gpu0(n_gpus, cfg, xyz.data(), src_cloud->size(), 0, m, err)
                           : run_cloud_on_gpus(n_gpus, cfg, xyz.data(), src_cloud->size(), 0, nullptr, nullptr, m, err);   // POINT_CLOUD: no filter, no pose (:118-120)
        if (st < 0) { std::cerr << "HoPE multi-GPU: " << err << std::endl; return 1; }
        std::vector<std::vector<float>> coeff;
        std::vector<int> points;
        for (const hope_cluster& k : m.clusters)
          if (k.pass == 1) { coeff.push_back({k.z_mean, k.min_x, k.min_y, k.max_x, k.max_y}); points.push_back(k.size); }
        PlaneIdTracker ids;
        ids.setID(coeff);
        std::printf("[%s] %d GPUs (%s)  raw %lld  down-sampled %lld  normal-fit %lld  clusters %zu  planes %zu\n", path_cloud.c_str(), n_gpus,
                    m.path.c_str(), (long long)m.n_raw_total, (long long)m.n_vox, (long long)m.n_fit, m.clusters.size(), coeff.size());
        for (size_t i = 0; i < coeff.size(); ++i)
          std::printf("  plane id %3d  z %.4f  x [%.3f, %.3f]  y [%.3f, %.3f]  points %d\n", ids.global_id_temp_[i], coeff[i][0], coeff[i][1],
                      coeff[i][3], coeff[i][2], coeff[i][4], points[i]);
        if (dump) {
          *dump << "{\"frame\": \"" << path_cloud << "\", \"gpus\": " << n_gpus << ", \"path\": \"" << m.path << "\", \"n_raw\": " << m.n_raw_total
                << ", \"n_vox\": " << m.n_vox << ", \"n_fit\": " << m.n_fit << ", \"n_clusters\": " << m.clusters.size() << ", \"status\": " << st
                << ", \"planes\": [";
          char buf[256];
          for (size_t i = 0; i < coeff.size(); ++i) {
            std::snprintf(buf, sizeof buf, "%s{\"id\": %d, \"coeff\": [%.9g, %.9g, %.9g, %.9g, %.9g], \"points\": %d, \"hull\": 0, \"area\": 0}",
                          i ? ", " : "", ids.global_id_temp_[i], coeff[i][0], coeff[i][1], coeff[i][2], coeff[i][3], coeff[i][4], points[i]);
            *dump << buf;
          }
          *dump << "]}" << std::endl;
        }
        return 0;
      }
#endif
      hope.setQ(qx, qy, qz, qw);
      hope.getHorizontalPlanes(src_cloud);
      if (hope.last_status() < 0) return 1;
      report(hope, path_cloud, dump);
      std::printf("[TIMER]: total: %.3f ms (whole call %.3f ms)\n", hope.last_timer_window_ms(), hope.last_ms());
      return 0;
    }
    if (type == TUM_SINGLE) {
      hope.setQ(qx, qy, qz, qw);
      hope.setT(tx, ty, tz);
      Image rgb;
      if (!readPng(path_rgb, rgb, err)) std::cerr << "warning: rgb image not read (" << err << "); colours only matter to the viewer" << std::endl;
      std::vector<uint16_t> depth; int w = 0, h = 0;
      if (!readDepthPng(path_depth, depth, w, h, err)) { std::cerr << err << std::endl; return -1; }
      hope.getHorizontalPlanes(depth.data(), w, h);
      if (hope.last_status() < 0) return 1;
      report(hope, path_depth, dump);
      std::printf("[TIMER]: total: %.3f ms (whole call %.3f ms)\n", hope.last_timer_window_ms(), hope.last_ms());
      return 0;
    }
    // TUM_LIST
    std::vector<TumEntry> list;
    if (!readTumList(path_list_all, list, err)) { std::cout << err << std::endl; return 0; }   // upstream prints and goes on with nothing
    if (list.empty()) return 0;
    // phaseInput keeps ONE translation for the whole list, the last line's (the globals tx_ ty_ tz_ are
    // overwritten per line, hope_node.cpp:88-90, and read after the loop, :231); the quaternion is per frame.
    const TumEntry& last = list.back();
    const size_t kBatch = 256;                 // frames per device call; 157 MB of depth
    // The device needs ~12 us per frame, inflating a 16-bit PNG ~2 ms: the images of a batch are decoded by all
    // host threads at once, and the next batch is decoded while the device works on the current one.
    struct Batch {
      std::vector<uint16_t> depth;
      std::vector<float> t, q;
      std::vector<std::string> names, messages;
      int w = 0, h = 0;
    };
    const unsigned n_workers = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    auto load_batch = [&](size_t b0, size_t b1) {
      const size_t n = b1 - b0;
      std::vector<std::vector<uint16_t>> img(n);
      std::vector<int> wi(n, 0), hi(n, 0);
      std::vector<std::string> errs(n);
      std::vector<char> ok(n, 0);
      std::atomic<size_t> next{0};
      std::vector<std::thread> pool;
      for (unsigned k = 0; k < std::min<size_t>(n_workers, n); ++k)
        pool.emplace_back([&] {
          for (size_t i = next++; i < n; i = next++)
            ok[i] = readDepthPng(path_prefix + "/" + list[b0 + i].depth, img[i], wi[i], hi[i], errs[i]) ? 1 : 0;
        });
      for (std::thread& th : pool) th.join();
      Batch b;
      for (size_t i = 0; i < n; ++i) {   // (in list order, with upstream's messages)
        if (!ok[i]) { b.messages.push_back(errs[i]); continue; }
        if (b.w == 0) { b.w = wi[i]; b.h = hi[i]; }
        if (wi[i] != b.w || hi[i] != b.h) { b.messages.push_back(list[b0 + i].depth + ": image size differs from the batch, skipped"); continue; }
        b.depth.insert(b.depth.end(), img[i].begin(), img[i].end());
        b.t.insert(b.t.end(), last.t, last.t + 3);
        b.q.insert(b.q.end(), list[b0 + i].q, list[b0 + i].q + 4);
        b.names.push_back(list[b0 + i].rgb);
      }
      return b;
    };
    double total_ms = 0;
    size_t done = 0;
    std::future<Batch> ahead = std::async(std::launch::async, load_batch, (size_t)0, std::min(list.size(), kBatch));
    for (size_t b0 = 0; b0 < list.size(); b0 += kBatch) {
      Batch b = ahead.get();
      const size_t n0 = b0 + kBatch;
      if (n0 < list.size()) ahead = std::async(std::launch::async, load_batch, n0, std::min(list.size(), n0 + kBatch));
      for (const std::string& m : b.messages) std::cerr << m << std::endl;
      if (b.names.empty()) continue;
      hope.getHorizontalPlanesBatch(b.depth.data(), b.w, b.h, (int)b.names.size(), b.t.data(), b.q.data(), [&](int f) {
        std::cout << "Processing: " << b.names[f] << std::endl;
        report(hope, b.names[f], dump);
      });
      if (hope.last_status() < 0) return 1;
      total_ms += hope.last_ms();
      done += b.names.size();
    }
    std::printf("[TIMER]: %zu frames, %.3f ms in the library, %.4f ms per frame\n", done, total_ms, done ? total_ms / done : 0.0);
  } catch (const std::exception& e) {
    std::cerr << "hope_node: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
