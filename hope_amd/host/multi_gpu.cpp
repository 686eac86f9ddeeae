// One cloud across several ranks: see multi_gpu.hpp.  Step numbers follow hope_amd/sharding.py and DESIGN.md section 7.
#include "multi_gpu.hpp"

#include <algorithm>
#include <cstring>
#include <numeric>

namespace hope {

std::vector<int32_t> band_partition(const std::vector<int64_t>& hist, int world)
{
  const int n = (int)hist.size();
  std::vector<int32_t> owner((size_t)n, 0);
  int64_t total = 0;
  for (int64_t h : hist) total += h;
  if (total == 0 || world == 1 || n == 0) return owner;
  std::vector<int64_t> upto((size_t)n);  // points in layers 0..k
  int64_t run = 0;
  for (int k = 0; k < n; ++k) { run += hist[k]; upto[k] = run; }
  // cut k = between layers k and k + 1; legal when the layer on either side is empty: ZGrowing only links
  // |dz| < th_z (z_growing.cpp:335-364) and a layer is th_z tall, so an empty layer separates clusters
  std::vector<int> legal;
  for (int k = 0; k + 1 < n; ++k) if (hist[k] == 0 || hist[k + 1] == 0) legal.push_back(k);
  if (!legal.empty()) {
    int prev = -1;
    for (int r = 1; r < world; ++r) {
      const int64_t target = total * r / world;
      size_t first = 0;
      while (first < legal.size() && legal[first] < prev) ++first;
      if (first == legal.size()) first = legal.size() - 1;
      int best = legal[first];
      int64_t bd = std::llabs(upto[best] - target);
      for (size_t i = first + 1; i < legal.size(); ++i) {
        const int64_t d = std::llabs(upto[legal[i]] - target);
        if (d < bd) { bd = d; best = legal[i]; }
      }
      const int k = std::max(best, prev);
      for (int l = k + 1; l < n; ++l) ++owner[l];
      prev = k;
    }
    return owner;
  }
  for (int k = 0; k < n; ++k) {
    const int64_t before = upto[k] - hist[k];
    owner[k] = (int32_t)std::min<int64_t>(before * world / total, world - 1);
  }
  return owner;
}

namespace {

struct Buf {  // a buffer of the backend, released on scope exit
  Comm& c;
  void* p = nullptr;
  Buf(Comm& comm, size_t bytes) : c(comm), p(comm.alloc(bytes)) {}
  ~Buf() { if (p) c.release(p); }
  Buf(const Buf&) = delete;
  Buf& operator=(const Buf&) = delete;
  template <class T> T* as() const { return static_cast<T*>(p); }
};

void whole_result(hope_ctx* ctx, MultiResult& out)
{
  hope_counts c{};
  hope_frame_counts(ctx, 0, &c);
  out.n_segments = c.n_segments;
  out.clusters.resize((size_t)c.n_clusters);
  if (c.n_clusters > 0) hope_get_clusters(ctx, 0, out.clusters.data(), c.n_clusters);
  out.labels.resize((size_t)c.n_fit);
  if (c.n_fit > 0) hope_get_labels(ctx, 0, out.labels.data());
  out.n_planes = 0;
  for (const hope_cluster& k : out.clusters) out.n_planes += k.pass == 1;
}

// One rank's progress through the collective call.  A library step that fails on ONE rank must not leave the others
// waiting in the next collective: the failing rank remembers its status, skips the library steps that follow and
// keeps taking part in the exchanges (with empty pieces) until the next agreement -- every small exchange of sizes
// carries the ranks' statuses, so all ranks learn of the failure there and leave together.
struct Rank {
  hope_ctx* ctx;
  Comm& comm;
  std::string& err;
  int fail_rank, fail_step;
  int st = 0;    // 0, or the first negative status of this rank
  int soft = 0;  // the last non-negative status (HOPE_EMPTY, HOPE_NO_NORMAL_FIT)
  int step = 0;

  // run a library step unless this rank has already failed
  template <class F>
  bool run(F&& f)
  {
    ++step;
    if (st < 0) return false;
    int r = (comm.rank() == fail_rank && step == fail_step) ? HOPE_E_NOMEM : f();
    if (r < 0) {
      st = r;
      const char* msg = (comm.rank() == fail_rank && step == fail_step) ? "injected failure" : hope_last_error(ctx);
      err = "step " + std::to_string(step) + ": " + (msg ? msg : "");
      return false;
    }
    soft = r;
    return true;
  }

  // all-gather of `mine` (n values per rank; slot 0 is overwritten with this rank's status).  Returns false when
  // some rank has failed: then every rank returns `st` (its own status, or HOPE_E_PEER for the others).
  bool agree(std::vector<int64_t>& mine, std::vector<int64_t>& all)
  {
    const int world = comm.world();
    const size_t n = mine.size();
    mine[0] = st;
    all.assign(n * (size_t)world, 0);
    Buf d_m(comm, sizeof(int64_t) * n), d_all(comm, sizeof(int64_t) * n * (size_t)world);
    comm.put(d_m.p, mine.data(), sizeof(int64_t) * n);
    comm.all_gather_v(d_m.p, d_all.p, std::vector<size_t>((size_t)world, sizeof(int64_t) * n));
    comm.get(all.data(), d_all.p, sizeof(int64_t) * all.size());
    for (int r = 0; r < world; ++r)
      if (all[(size_t)r * n] < 0) {
        if (st >= 0) {
          st = HOPE_E_PEER;
          err = "rank " + std::to_string(r) + " failed with status " + std::to_string((long long)all[(size_t)r * n]);
        }
        return false;
      }
    return true;
  }
};

}  // namespace

int segment_cloud_multi(hope_ctx* ctx, Comm& comm, const float* xyz_host, size_t n, int apply, const hope_config& cfg,
                        MultiResult& out, std::string& err, bool allow_bands, int fail_rank, int fail_step)
{
  const int world = comm.world(), rank = comm.rank(), dev = comm.on_device();
  out = MultiResult();
  Rank R{ctx, comm, err, fail_rank, fail_step};
  std::vector<int64_t> mine, all;
  // the rank's share, where the backend's collectives and the library read it
  Buf share(comm, std::max<size_t>(n, 1) * 12);
  comm.put(share.p, xyz_host, n * 12);
  // 1. bounding box and survivor count: every rank's (status, count, box), reduced by every rank the same way
  float bb[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int64_t n_raw = 0;
  const bool have_box = R.run([&] { return hope_shard_bbox(ctx, share.p, n, 12, apply, dev, bb, &n_raw); });
  mine.assign(8, 0);
  mine[1] = have_box ? n_raw : 0;
  for (int i = 0; i < 6; ++i) {
    uint32_t u;
    std::memcpy(&u, &bb[i], 4);
    mine[2 + i] = (int64_t)u;
  }
  if (!R.agree(mine, all)) return R.st;
  for (int r = 0; r < world; ++r) {
    out.n_raw_total += all[(size_t)r * 8 + 1];
    for (int i = 0; i < 6; ++i) {
      const uint32_t u = (uint32_t)all[(size_t)r * 8 + 2 + i];
      float f;
      std::memcpy(&f, &u, 4);
      // (MIN of the lower corners, MAX of the upper ones, as the all-reduce of [-min, max] by MAX did: a rank
      //  without survivors reports +inf / -inf)
      if (r == 0) bb[i] = f;
      else if (i < 3) bb[i] = -std::max(-bb[i], -f);
      else bb[i] = std::max(bb[i], f);
    }
  }
  if (out.n_raw_total == 0) {
    const int st = hope_segment_cloud_xyz(ctx, xyz_host, 0, 12, 0, 0);
    out.path = "empty";
    if (st < 0) err = hope_last_error(ctx);
    return st;  // (every rank takes this branch)
  }
  int32_t min_b[3], div_b[3] = {0, 0, 1};
  // (the same box on every rank: this step fails on all of them or on none)
  if (!R.run([&] { return hope_grid_from_bbox(ctx, bb, min_b, div_b); })) return R.st;
  const int n_bins = div_b[2];
  // 2. z-layer histogram -> bands; one more bin counts the ranks that have failed
  std::vector<int64_t> hist((size_t)n_bins + 1, 0);
  R.run([&] { return hope_shard_z_histogram(ctx, share.p, n, 12, apply, dev, bb, hist.data(), n_bins); });
  if (R.st < 0) std::fill(hist.begin(), hist.end(), 0);
  hist[(size_t)n_bins] = R.st < 0 ? 1 : 0;
  {
    Buf d_hist(comm, sizeof(int64_t) * hist.size());
    comm.put(d_hist.p, hist.data(), sizeof(int64_t) * hist.size());
    comm.all_reduce_sum_i64(d_hist.as<int64_t>(), hist.size());
    comm.get(hist.data(), d_hist.p, sizeof(int64_t) * hist.size());
  }
  if (hist[(size_t)n_bins] != 0) {
    if (R.st >= 0) {
      R.st = HOPE_E_PEER;
      err = "a rank failed in the z-histogram step";
    }
    return R.st;
  }
  hist.pop_back();
  out.layer_owner = band_partition(hist, world);
  // 3. route the survivors to their band owners
  Buf routed(comm, std::max<int64_t>(n_raw, 1) * 12);
  std::vector<int64_t> send((size_t)world, 0), recv((size_t)world, 0);
  R.run([&] {
    return hope_shard_route(ctx, share.p, n, 12, apply, dev, bb, out.layer_owner.data(), n_bins, world, routed.as<float>(), dev,
                            send.data());
  });
  mine.assign((size_t)world + 1, 0);
  for (int d = 0; d < world; ++d) mine[1 + d] = send[d];
  if (!R.agree(mine, all)) return R.st;
  for (int s = 0; s < world; ++s) recv[s] = all[(size_t)s * (world + 1) + 1 + rank];
  out.sent = send;
  out.received = recv;
  const int64_t n_mine = std::accumulate(recv.begin(), recv.end(), (int64_t)0);
  Buf mine_pts(comm, std::max<int64_t>(n_mine, 1) * 12);
  {
    std::vector<int64_t> sf(send), rf(recv);
    for (int64_t& v : sf) v *= 3;
    for (int64_t& v : rf) v *= 3;
    comm.all_to_all_v(routed.as<float>(), sf, mine_pts.as<float>(), rf);
  }
  // 4. VoxelGrid on the band with the global grid; gather the down-sampled cloud
  hope_counts cnt{};
  if (R.run([&] { return hope_downsample_cloud_bbox(ctx, mine_pts.p, (size_t)n_mine, 12, dev, bb); }))
    hope_frame_counts(ctx, 0, &cnt);
  Buf vox_own(comm, std::max<int64_t>(cnt.n_vox, 1) * 12);
  if (cnt.n_vox > 0) R.run([&] { return hope_get_downsampled_to(ctx, 0, vox_own.as<float>(), dev); });
  mine.assign(2, 0);
  mine[1] = cnt.n_vox;
  if (!R.agree(mine, all)) return R.st;
  std::vector<int64_t> vox_sizes((size_t)world, 0);
  for (int r = 0; r < world; ++r) vox_sizes[r] = all[(size_t)r * 2 + 1];
  const int64_t n_vox = std::accumulate(vox_sizes.begin(), vox_sizes.end(), (int64_t)0);
  out.n_vox = n_vox;
  if (n_vox == 0) {
    const int st = hope_segment_cloud_xyz(ctx, xyz_host, 0, 12, 0, 0);
    out.path = "empty";
    if (st < 0) err = hope_last_error(ctx);
    return st;
  }
  Buf vox_all(comm, (size_t)n_vox * 12);
  {
    std::vector<size_t> bytes((size_t)world);
    for (int r = 0; r < world; ++r) bytes[r] = (size_t)vox_sizes[r] * 12;
    comm.all_gather_v(vox_own.p, vox_all.p, bytes);
  }
  // 5. A6..A12 on the whole down-sampled cloud; A6 and A8 shared by z band (a contiguous index range, no halo)
  if (world == 1) {
    if (!R.run([&] { return hope_segment_downsampled(ctx, vox_all.p, (size_t)n_vox, 12, dev, bb, out.n_raw_total); }))
      return R.st;
    out.path = "one rank";
    hope_frame_counts(ctx, 0, &cnt);
    out.n_fit = cnt.n_fit;
    whole_result(ctx, out);
    return R.soft;
  }
  int64_t first = 0;
  for (int r = 0; r < rank; ++r) first += vox_sizes[r];
  const int64_t count = vox_sizes[rank];
  R.run([&] {
    return hope_segment_downsampled_staged(ctx, vox_all.p, (size_t)n_vox, 12, dev, bb, out.n_raw_total, first, count, nullptr,
                                           nullptr, 1);
  });
  Buf normals_all(comm, (size_t)n_vox * 16);
  {
    Buf own(comm, std::max<int64_t>(count, 1) * 16);
    if (count > 0) R.run([&] { return hope_get_normals_rows(ctx, 0, first, count, own.as<float>(), dev); });
    std::vector<size_t> bytes((size_t)world);
    for (int r = 0; r < world; ++r) bytes[r] = (size_t)vox_sizes[r] * 16;
    comm.all_gather_v(own.p, normals_all.p, bytes);  // (a failed rank's piece is not used: the agreement below ends the call)
  }
  int64_t lo = 0, hi = 0;
  if (R.run([&] {
        return hope_segment_downsampled_staged(ctx, vox_all.p, (size_t)n_vox, 12, dev, bb, out.n_raw_total, first, count,
                                               normals_all.as<float>(), nullptr, 2);
      })) {
    hope_frame_counts(ctx, 0, &cnt);
    std::vector<int32_t> fit_idx((size_t)cnt.n_fit);
    if (cnt.n_fit > 0) R.run([&] { return hope_get_fit_indices(ctx, 0, fit_idx.data()); });
    if (R.st >= 0) {
      lo = std::lower_bound(fit_idx.begin(), fit_idx.end(), (int32_t)first) - fit_idx.begin();
      hi = std::lower_bound(fit_idx.begin(), fit_idx.end(), (int32_t)(first + count)) - fit_idx.begin();
    }
  }
  const int K = cfg.knn;
  Buf own_k(comm, std::max<int64_t>(hi - lo, 1) * K * 4);
  if (hi > lo) R.run([&] { return hope_get_knn_rows(ctx, 0, lo, hi - lo, own_k.as<int32_t>(), dev); });
  // 6. A9..A12 by z band when no edge ZGrowing would follow leaves a band (every rank checks its own rows).  The
  //    size filter is what drops the other bands' points (rows without neighbours: singletons), so a minimum
  //    cluster size of 1 takes the replicated path.
  int64_t crossing = (allow_bands && cfg.min_cluster > 1) ? 0 : 1;
  if (crossing == 0 && hi > lo) R.run([&] { return hope_count_band_crossings(ctx, 0, lo, hi - lo, &crossing); });
  mine.assign(3, 0);
  mine[1] = hi - lo;
  mine[2] = crossing;
  if (!R.agree(mine, all)) return R.st;
  std::vector<int64_t> fit_sizes((size_t)world, 0);
  crossing = 0;
  for (int r = 0; r < world; ++r) {
    fit_sizes[r] = all[(size_t)r * 3 + 1];
    crossing += all[(size_t)r * 3 + 2];
  }
  const int64_t n_fit = std::accumulate(fit_sizes.begin(), fit_sizes.end(), (int64_t)0);
  out.n_fit = n_fit;
  Buf knn_in(comm, (size_t)std::max<int64_t>(n_fit, 1) * K * 4);
  if (crossing == 0) {
    // the own band only: the other bands' points come as rows without neighbours (singletons, dropped by the size
    // filter); seed ranks are those of the whole fit cloud, so the bands' records merge by seed rank
    comm.fill_ff(knn_in.p, (size_t)std::max<int64_t>(n_fit, 1) * K * 4);
    comm.copy(knn_in.as<int32_t>() + lo * K, own_k.p, (size_t)(hi - lo) * K * 4);
  } else {
    std::vector<size_t> bytes((size_t)world);
    for (int r = 0; r < world; ++r) bytes[r] = (size_t)fit_sizes[r] * K * 4;
    if (n_fit > 0) comm.all_gather_v(own_k.p, knn_in.p, bytes);
  }
  R.run([&] {
    return hope_segment_downsampled_staged(ctx, vox_all.p, (size_t)n_vox, 12, dev, bb, out.n_raw_total, first, count,
                                           normals_all.as<float>(), knn_in.as<int32_t>(), 0);
  });
  const int final_status = R.soft;
  MultiResult own;
  if (R.st >= 0) whole_result(ctx, own);
  const int64_t seg_own = (int64_t)own.n_segments - (crossing == 0 ? n_fit - (hi - lo) : 0);
  mine.assign(3, 0);
  mine[1] = (int64_t)own.clusters.size();
  mine[2] = seg_own;
  if (!R.agree(mine, all)) return R.st;
  if (crossing != 0) {
    out.path = "replicated";
    out.n_segments = own.n_segments;
    out.n_planes = own.n_planes;
    out.clusters = std::move(own.clusters);
    out.labels = std::move(own.labels);
    return final_status;
  }
  out.path = "z-bands";
  // gather (clusters, own labels) of every band and merge by seed rank
  int64_t n_cl = 0, n_seg = 0;
  std::vector<size_t> cl_bytes((size_t)world), lab_bytes((size_t)world);
  std::vector<int64_t> cl_first((size_t)world + 1, 0);
  for (int r = 0; r < world; ++r) {
    const int64_t ncl_r = all[(size_t)r * 3 + 1];
    cl_first[r + 1] = cl_first[r] + ncl_r;
    n_cl += ncl_r;
    n_seg += all[(size_t)r * 3 + 2];
    cl_bytes[r] = (size_t)ncl_r * sizeof(hope_cluster);
    lab_bytes[r] = (size_t)fit_sizes[r] * 4;
  }
  std::vector<hope_cluster> recs((size_t)n_cl);
  std::vector<int32_t> lab_all((size_t)n_fit);
  {
    Buf d_own(comm, std::max<size_t>(cl_bytes[rank], 1)), d_all(comm, std::max<size_t>((size_t)n_cl * sizeof(hope_cluster), 1));
    comm.put(d_own.p, own.clusters.data(), cl_bytes[rank]);
    comm.all_gather_v(d_own.p, d_all.p, cl_bytes);
    comm.get(recs.data(), d_all.p, (size_t)n_cl * sizeof(hope_cluster));
    Buf l_own(comm, std::max<size_t>(lab_bytes[rank], 1)), l_all(comm, std::max<size_t>((size_t)n_fit * 4, 1));
    comm.put(l_own.p, own.labels.data() + lo, lab_bytes[rank]);
    comm.all_gather_v(l_own.p, l_all.p, lab_bytes);
    comm.get(lab_all.data(), l_all.p, (size_t)n_fit * 4);
  }
  std::vector<int64_t> order((size_t)n_cl);
  std::iota(order.begin(), order.end(), (int64_t)0);
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return recs[a].seed_rank < recs[b].seed_rank; });
  std::vector<int32_t> new_id((size_t)n_cl);
  out.clusters.resize((size_t)n_cl);
  for (int64_t i = 0; i < n_cl; ++i) {
    new_id[order[i]] = (int32_t)i;
    out.clusters[i] = recs[order[i]];
    if (i > 0 && out.clusters[i].seed_rank == out.clusters[i - 1].seed_rank) {
      err = "two z bands report a cluster with the same seed";  // (the same records on every rank: all return here)
      return HOPE_E_INVALID;
    }
  }
  out.labels.assign((size_t)n_fit, -1);
  int64_t at = 0;
  for (int r = 0; r < world; ++r)
    for (int64_t i = 0; i < fit_sizes[r]; ++i, ++at) {
      const int32_t l = lab_all[at];
      out.labels[at] = l >= 0 ? new_id[cl_first[r] + l] : -1;
    }
  out.n_segments = (int)n_seg;
  out.n_planes = 0;
  for (const hope_cluster& k : out.clusters) out.n_planes += k.pass == 1;
  return final_status;
}

}  // namespace hope
