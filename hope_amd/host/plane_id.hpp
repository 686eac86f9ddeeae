// Temporal plane-ID tracking: PlaneSegment::setID
// (plane_segment.cpp:496-517) over Utilities::matchID / matNormalize / searchAvailableID / isInVector
// (utilities.cpp:930-1001,1027-1036,1063-1159).  Host-side and tiny.
// The reference matches with cv::FlannBasedMatcher (randomised kd-trees; OpenCV is not in this image).
// Restated as the exact nearest neighbour in L2 over the normalised 5-feature rows, ties to the lower
// train index: what FLANN returns whenever its search is exhaustive, which it is for the tens of
// rows a frame has.  PARITY UNPINNED beyond that.
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

namespace hope {

// utilities.cpp:975-1001
inline void searchAvailableID(const std::vector<int>& id_used, std::vector<int>& id_ava, size_t limit = 1024)
{
  for (size_t i = 0; i < limit; ++i) {
    bool used = false;
    for (int n : id_used) if ((int)i == n) { used = true; break; }
    if (!used) id_ava.push_back((int)i);
  }
}

// utilities.cpp:930-972: per column, z-score over the rows of BOTH matrices (population sd), float
// accumulators, pow() evaluated in double like the reference's pow(float, int).
inline void matNormalize(const std::vector<std::vector<float>>& query_in, const std::vector<std::vector<float>>& train_in, int dim,
                         std::vector<std::vector<float>>& query_out, std::vector<std::vector<float>>& train_out)
{
  query_out.assign(query_in.size(), std::vector<float>(dim));
  train_out.assign(train_in.size(), std::vector<float>(dim));
  const float rows = float(query_in.size() + train_in.size());
  for (int c = 0; c < dim; ++c) {
    float acc = 0;
    for (auto& r : query_in) acc += r[c];
    for (auto& r : train_in) acc += r[c];
    const float r_mean = acc / rows;
    acc = 0;
    for (auto& r : query_in) acc = (float)((double)acc + std::pow((double)(r[c] - r_mean), 2.0));
    for (auto& r : train_in) acc = (float)((double)acc + std::pow((double)(r[c] - r_mean), 2.0));
    const float sd = std::sqrt(acc / rows);
    for (size_t r = 0; r < query_in.size(); ++r) query_out[r][c] = (query_in[r][c] - r_mean) / sd;
    for (size_t r = 0; r < train_in.size(); ++r) train_out[r][c] = (train_in[r][c] - r_mean) / sd;
  }
}

// utilities.cpp:1063-1159.  global/local: feature rows {z, min_x, min_y, max_x, max_y}; in: ids of the
// global rows; out: ids for the local rows (matched id, or the smallest unused id).
inline void matchID(const std::vector<std::vector<float>>& global, const std::vector<std::vector<float>>& local,
                    const std::vector<int>& in, std::vector<int>& out, int feature_dim)
{
  std::vector<int> id_ava;
  searchAvailableID(in, id_ava);
  out.resize(local.size());
  std::vector<std::vector<float>> query, train;
  matNormalize(local, global, feature_dim, query, train);
  struct DMatch { int queryIdx, trainIdx; float distance; };
  std::vector<DMatch> matches;
  for (size_t q = 0; q < query.size(); ++q) {
    int best = -1; float best_d = 0.f;
    for (size_t t = 0; t < train.size(); ++t) {
      float d = 0.f;
      for (int c = 0; c < feature_dim; ++c) { const float e = query[q][c] - train[t][c]; d += e * e; }
      if (best < 0 || d < best_d) { best = (int)t; best_d = d; }
    }
    if (best >= 0) matches.push_back({(int)q, best, std::sqrt(best_d)});
  }
  std::vector<float> g_dist_temp(global.size(), 10000);
  std::vector<int> match_for_g(global.size(), -1);
  for (size_t i = 0; i < global.size(); ++i)
    for (auto& m : matches)
      if ((int)i == m.trainIdx && m.distance < g_dist_temp[i]) { g_dist_temp[i] = m.distance; match_for_g[i] = m.queryIdx; }
  for (size_t j = 0; j < local.size(); ++j) {
    int pos = -1;
    for (size_t i = 0; i < match_for_g.size(); ++i) if ((int)j == match_for_g[i]) { pos = (int)i; break; }
    if (pos >= 0) out[j] = in[pos];
    else { out[j] = id_ava.empty() ? -1 : id_ava[0]; if (!id_ava.empty()) id_ava.erase(id_ava.begin()); }
  }
}

// PlaneSegment::setID state (plane_segment.cpp:496-517)
struct PlaneIdTracker {
  std::vector<std::vector<float>> global_coeff_temp_;
  std::vector<int> global_id_temp_;
  void setID(const std::vector<std::vector<float>>& plane_coeff)
  {
    if (global_id_temp_.empty()) {
      for (size_t i = 0; i < plane_coeff.size(); ++i) { global_id_temp_.push_back((int)i); global_coeff_temp_.push_back(plane_coeff[i]); }
    } else {
      std::vector<int> local_id_temp;
      matchID(global_coeff_temp_, plane_coeff, global_id_temp_, local_id_temp, 5);
      global_coeff_temp_.clear(); global_id_temp_.clear();
      for (size_t i = 0; i < plane_coeff.size(); ++i) { global_id_temp_.push_back(local_id_temp[i]); global_coeff_temp_.push_back(plane_coeff[i]); }
    }
  }
};

}  // namespace hope
