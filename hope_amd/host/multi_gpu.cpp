// One cloud across several ranks: see multi_gpu.hpp.  Step numbers follow hope_amd/sharding.py and DESIGN.md section 7.
#include "multi_gpu.hpp"

#include <algorithm>
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

int fail(hope_ctx* ctx, int st, std::string& err)
{
  err = hope_last_error(ctx);
  return st;
}

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

}  // namespace

int segment_cloud_multi(hope_ctx* ctx, Comm& comm, const float* xyz_host, size_t n, int apply, const hope_config& cfg,
                        MultiResult& out, std::string& err, bool allow_bands)
{
  const int world = comm.world(), rank = comm.rank(), dev = comm.on_device();
  out = MultiResult();
  // the rank's share, where the backend's collectives and the library read it
  Buf share(comm, std::max<size_t>(n, 1) * 12);
  comm.put(share.p, xyz_host, n * 12);
  int st;
  // 1. bounding box (MIN/MAX as one MAX over [-min, max]) and survivor count
  float bb[6];
  int64_t n_raw = 0;
  if ((st = hope_shard_bbox(ctx, share.p, n, 12, apply, dev, bb, &n_raw)) < 0) return fail(ctx, st, err);
  {
    const float mm[6] = {-bb[0], -bb[1], -bb[2], bb[3], bb[4], bb[5]};
    Buf d_mm(comm, sizeof mm), d_cnt(comm, sizeof(int64_t));
    comm.put(d_mm.p, mm, sizeof mm);
    comm.put(d_cnt.p, &n_raw, sizeof n_raw);
    comm.all_reduce_max_f32(d_mm.as<float>(), 6);
    comm.all_reduce_sum_i64(d_cnt.as<int64_t>(), 1);
    float r[6];
    comm.get(r, d_mm.p, sizeof r);
    comm.get(&out.n_raw_total, d_cnt.p, sizeof(int64_t));
    for (int i = 0; i < 3; ++i) { bb[i] = -r[i]; bb[3 + i] = r[3 + i]; }
  }
  if (out.n_raw_total == 0) {
    st = hope_segment_cloud_xyz(ctx, xyz_host, 0, 12, 0, 0);
    out.path = "empty";
    return st < 0 ? fail(ctx, st, err) : st;
  }
  int32_t min_b[3], div_b[3];
  if ((st = hope_grid_from_bbox(ctx, bb, min_b, div_b)) < 0) return fail(ctx, st, err);
  const int n_bins = div_b[2];
  // 2. z-layer histogram -> bands
  std::vector<int64_t> hist((size_t)n_bins, 0);
  if ((st = hope_shard_z_histogram(ctx, share.p, n, 12, apply, dev, bb, hist.data(), n_bins)) < 0) return fail(ctx, st, err);
  {
    Buf d_hist(comm, sizeof(int64_t) * (size_t)n_bins);
    comm.put(d_hist.p, hist.data(), sizeof(int64_t) * (size_t)n_bins);
    comm.all_reduce_sum_i64(d_hist.as<int64_t>(), (size_t)n_bins);
    comm.get(hist.data(), d_hist.p, sizeof(int64_t) * (size_t)n_bins);
  }
  out.layer_owner = band_partition(hist, world);
  // 3. route the survivors to their band owners
  Buf routed(comm, std::max<int64_t>(n_raw, 1) * 12);
  std::vector<int64_t> send((size_t)world, 0), recv((size_t)world, 0);
  if ((st = hope_shard_route(ctx, share.p, n, 12, apply, dev, bb, out.layer_owner.data(), n_bins, world, routed.as<float>(),
                             dev, send.data())) < 0)
    return fail(ctx, st, err);
  {
    std::vector<int64_t> all((size_t)world * world);
    Buf d_send(comm, sizeof(int64_t) * (size_t)world), d_all(comm, sizeof(int64_t) * (size_t)world * world);
    comm.put(d_send.p, send.data(), sizeof(int64_t) * (size_t)world);
    comm.all_gather_v(d_send.p, d_all.p, std::vector<size_t>((size_t)world, sizeof(int64_t) * (size_t)world));
    comm.get(all.data(), d_all.p, sizeof(int64_t) * all.size());
    for (int s = 0; s < world; ++s) recv[s] = all[(size_t)s * world + rank];
  }
  out.sent = send;
  out.received = recv;
  const int64_t n_mine = std::accumulate(recv.begin(), recv.end(), (int64_t)0);
  Buf mine(comm, std::max<int64_t>(n_mine, 1) * 12);
  {
    std::vector<int64_t> sf(send), rf(recv);
    for (int64_t& v : sf) v *= 3;
    for (int64_t& v : rf) v *= 3;
    comm.all_to_all_v(routed.as<float>(), sf, mine.as<float>(), rf);
  }
  // 4. VoxelGrid on the band with the global grid; gather the down-sampled cloud
  if ((st = hope_downsample_cloud_bbox(ctx, mine.p, (size_t)n_mine, 12, dev, bb)) < 0) return fail(ctx, st, err);
  hope_counts cnt{};
  hope_frame_counts(ctx, 0, &cnt);
  std::vector<int64_t> vox_sizes((size_t)world, 0);
  {
    Buf d_s(comm, sizeof(int64_t) * (size_t)world);
    vox_sizes[rank] = cnt.n_vox;
    comm.put(d_s.p, vox_sizes.data(), sizeof(int64_t) * (size_t)world);
    comm.all_reduce_sum_i64(d_s.as<int64_t>(), (size_t)world);
    comm.get(vox_sizes.data(), d_s.p, sizeof(int64_t) * (size_t)world);
  }
  const int64_t n_vox = std::accumulate(vox_sizes.begin(), vox_sizes.end(), (int64_t)0);
  out.n_vox = n_vox;
  if (n_vox == 0) {
    st = hope_segment_cloud_xyz(ctx, xyz_host, 0, 12, 0, 0);
    out.path = "empty";
    return st < 0 ? fail(ctx, st, err) : st;
  }
  Buf vox_all(comm, (size_t)n_vox * 12);
  {
    Buf vox_own(comm, std::max<int64_t>(cnt.n_vox, 1) * 12);
    if (cnt.n_vox > 0 && (st = hope_get_downsampled_to(ctx, 0, vox_own.as<float>(), dev)) < 0) return fail(ctx, st, err);
    std::vector<size_t> bytes((size_t)world);
    for (int r = 0; r < world; ++r) bytes[r] = (size_t)vox_sizes[r] * 12;
    comm.all_gather_v(vox_own.p, vox_all.p, bytes);
  }
  // 5. A6..A12 on the whole down-sampled cloud; A6 and A8 shared by z band (a contiguous index range, no halo)
  if (world == 1) {
    st = hope_segment_downsampled(ctx, vox_all.p, (size_t)n_vox, 12, dev, bb, out.n_raw_total);
    if (st < 0) return fail(ctx, st, err);
    out.path = "one rank";
    hope_frame_counts(ctx, 0, &cnt);
    out.n_fit = cnt.n_fit;
    whole_result(ctx, out);
    return st;
  }
  int64_t first = 0;
  for (int r = 0; r < rank; ++r) first += vox_sizes[r];
  const int64_t count = vox_sizes[rank];
  if ((st = hope_segment_downsampled_staged(ctx, vox_all.p, (size_t)n_vox, 12, dev, bb, out.n_raw_total, first, count, nullptr,
                                            nullptr, 1)) < 0)
    return fail(ctx, st, err);
  Buf normals_all(comm, (size_t)n_vox * 16);
  {
    Buf own(comm, std::max<int64_t>(count, 1) * 16);
    if (count > 0 && (st = hope_get_normals_rows(ctx, 0, first, count, own.as<float>(), dev)) < 0) return fail(ctx, st, err);
    std::vector<size_t> bytes((size_t)world);
    for (int r = 0; r < world; ++r) bytes[r] = (size_t)vox_sizes[r] * 16;
    comm.all_gather_v(own.p, normals_all.p, bytes);
  }
  if ((st = hope_segment_downsampled_staged(ctx, vox_all.p, (size_t)n_vox, 12, dev, bb, out.n_raw_total, first, count,
                                            normals_all.as<float>(), nullptr, 2)) < 0)
    return fail(ctx, st, err);
  hope_frame_counts(ctx, 0, &cnt);
  std::vector<int32_t> fit_idx((size_t)cnt.n_fit);
  if (cnt.n_fit > 0 && (st = hope_get_fit_indices(ctx, 0, fit_idx.data())) < 0) return fail(ctx, st, err);
  const int64_t lo = std::lower_bound(fit_idx.begin(), fit_idx.end(), (int32_t)first) - fit_idx.begin();
  const int64_t hi = std::lower_bound(fit_idx.begin(), fit_idx.end(), (int32_t)(first + count)) - fit_idx.begin();
  std::vector<int64_t> fit_sizes((size_t)world, 0);
  {
    Buf d_s(comm, sizeof(int64_t) * (size_t)world);
    fit_sizes[rank] = hi - lo;
    comm.put(d_s.p, fit_sizes.data(), sizeof(int64_t) * (size_t)world);
    comm.all_reduce_sum_i64(d_s.as<int64_t>(), (size_t)world);
    comm.get(fit_sizes.data(), d_s.p, sizeof(int64_t) * (size_t)world);
  }
  const int64_t n_fit = std::accumulate(fit_sizes.begin(), fit_sizes.end(), (int64_t)0);
  out.n_fit = n_fit;
  const int K = cfg.knn;
  Buf own_k(comm, std::max<int64_t>(hi - lo, 1) * K * 4);
  if (hi > lo && (st = hope_get_knn_rows(ctx, 0, lo, hi - lo, own_k.as<int32_t>(), dev)) < 0) return fail(ctx, st, err);
  // 6. A9..A12 by z band when no edge ZGrowing would follow leaves a band (every rank checks its own rows)
  int64_t crossing = allow_bands ? 0 : 1;
  if (allow_bands && hi > lo && (st = hope_count_band_crossings(ctx, 0, lo, hi - lo, &crossing)) < 0) return fail(ctx, st, err);
  {
    Buf d_c(comm, sizeof(int64_t));
    comm.put(d_c.p, &crossing, sizeof crossing);
    comm.all_reduce_sum_i64(d_c.as<int64_t>(), 1);
    comm.get(&crossing, d_c.p, sizeof crossing);
  }
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
  st = hope_segment_downsampled_staged(ctx, vox_all.p, (size_t)n_vox, 12, dev, bb, out.n_raw_total, first, count,
                                       normals_all.as<float>(), knn_in.as<int32_t>(), 0);
  if (st < 0) return fail(ctx, st, err);
  if (crossing != 0) {
    out.path = "replicated";
    whole_result(ctx, out);
    return st;
  }
  out.path = "z-bands";
  // gather (clusters, own labels, own segments) of every band and merge by seed rank
  MultiResult own;
  whole_result(ctx, own);
  const int64_t seg_own = cfg.min_cluster > 1 ? (int64_t)own.n_segments - (n_fit - (hi - lo)) : (int64_t)own.n_segments;
  std::vector<int64_t> meta((size_t)world * 2, 0), my_meta = {(int64_t)own.clusters.size(), seg_own};
  {
    Buf d_m(comm, sizeof(int64_t) * 2), d_all(comm, sizeof(int64_t) * 2 * (size_t)world);
    comm.put(d_m.p, my_meta.data(), sizeof(int64_t) * 2);
    comm.all_gather_v(d_m.p, d_all.p, std::vector<size_t>((size_t)world, sizeof(int64_t) * 2));
    comm.get(meta.data(), d_all.p, sizeof(int64_t) * meta.size());
  }
  int64_t n_cl = 0, n_seg = 0;
  std::vector<size_t> cl_bytes((size_t)world), lab_bytes((size_t)world);
  std::vector<int64_t> cl_first((size_t)world + 1, 0);
  for (int r = 0; r < world; ++r) {
    cl_first[r + 1] = cl_first[r] + meta[2 * r];
    n_cl += meta[2 * r];
    n_seg += meta[2 * r + 1];
    cl_bytes[r] = (size_t)meta[2 * r] * sizeof(hope_cluster);
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
      err = "two z bands report a cluster with the same seed";
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
  return st;
}

}  // namespace hope
