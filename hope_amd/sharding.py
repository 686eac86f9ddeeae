"""Multi-GPU use of the path: frames are independent extraction problems, so a batch is sharded
by contiguous frame range (one process per GPU) and only the small plane records are gathered on
rank 0.  There is no data-path collective ("Frames: yes, trivially").


A single large cloud (configs 3/4) is sharded by contiguous point range; the raw-point work
(A2, A3, bounding box, z-layer histogram, VoxelGrid) runs on every rank's share and the three
exchanges the path really needs sit between the stages: a MIN/MAX all-reduce of the bounding box
(6 floats), a SUM all-reduce of the int64 z-layer histogram (a few hundred bins), one all-to-all
that sends every surviving point to the rank owning its z band, and one all-gather of the
down-sampled cloud (~N_raw/10..40 points).  A6..A12 then run on the gathered down-sampled cloud, the
two neighbour searches (A6 normals, A8 k-NN) shared between the ranks by z band with an all-gather of
the normals and one of the k-NN rows (hope_segment_downsampled_staged); see segment_cloud_sharded.  Every step is bit-identical to the single-process path."""
import numpy as np


def frame_partition(n_frames, world, rank):
    """Contiguous, balanced [lo, hi) frame range of `rank` (first n_frames % world ranks get one more)."""
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def segment_depth_sharded(ctx, depth, t, q, rank, world):
    """Runs this rank's share of a [n, h, w] uint16 batch through `ctx` (any implementation of the
    C ABI) and returns (lo, hi, [per-frame dict])."""
    n = depth.shape[0]
    lo, hi = frame_partition(n, world, rank)
    out = []
    if hi > lo:
        ctx.set_frame_poses(t[lo:hi], q[lo:hi])
        ctx.segment_depth(depth[lo:hi])
        for f in range(hi - lo):
            c = ctx.counts(f)
            out.append({"frame": lo + f, "n_raw": int(c.n_raw), "n_vox": int(c.n_vox), "n_fit": int(c.n_fit),
                        "n_planes": int(c.n_planes), "clusters": ctx.clusters(f).copy()})
    return lo, hi, out


def gather_results(local, dist, dst=0):
    """Gathers the per-frame records of every rank on `dst`, ordered by frame."""
    world = dist.get_world_size()
    if dist.get_rank() == dst:
        parts = [None] * world
        dist.gather_object(local, parts, dst=dst)
        merged = [r for p in parts for r in p]
        assert [r["frame"] for r in merged] == sorted(r["frame"] for r in merged)
        return merged
    dist.gather_object(local, None, dst=dst)
    return None


def records_equal(a, b):
    return (a["frame"] == b["frame"] and a["n_raw"] == b["n_raw"] and a["n_vox"] == b["n_vox"] and
            a["n_fit"] == b["n_fit"] and a["n_planes"] == b["n_planes"] and
            np.asarray(a["clusters"]).tobytes() == np.asarray(b["clusters"]).tobytes())


# ---- one cloud across several ranks --------------------------------------------------------------
def band_partition(hist, world):
    """Owner rank of every z layer: contiguous bands holding about equal numbers of points (hist: all-reduced
    survivors per layer).  Deterministic, so every rank derives the same map.

    ZGrowing only links neighbours with |dz| < th_z (z_growing.cpp:335-364) and a layer is th_z tall, so an
    EMPTY layer separates clusters exactly: bands cut there are independent extraction problems for A9..A12.
    A cut is legal when the layer on either side of it is empty; among the legal cuts the one nearest the
    equal-population target is taken (several ranks may get the same cut: a rank then owns nothing).  Without
    any empty layer the cuts fall at equal population and the caller has to check for edges across them."""
    hist = np.asarray(hist, dtype=np.int64)
    total = int(hist.sum())
    n = hist.shape[0]
    owner = np.zeros(n, np.int32)
    if total == 0 or world == 1 or n == 0:
        return owner
    upto = np.cumsum(hist)                    # points in layers 0..k
    # cut k = between layers k and k + 1
    legal = [k for k in range(n - 1) if hist[k] == 0 or hist[k + 1] == 0]
    if legal:
        cuts, prev = [], -1
        lg = np.asarray(legal)
        for r in range(1, world):
            target = total * r // world
            cand = lg[lg >= prev]
            if cand.size == 0:
                cand = lg[-1:]
            k = int(cand[np.argmin(np.abs(upto[cand] - target))])
            k = max(k, prev)
            cuts.append(k)
            prev = k
        for k in cuts:
            owner[k + 1:] += 1
        return owner
    before = upto - hist                      # points in the layers below
    owner[:] = np.minimum((before * world) // total, world - 1).astype(np.int32)
    return owner


def merge_band_results(parts, n_fit):
    """parts: per rank (lo, hi, clusters of the own band [CLUSTER_DTYPE, with the GLOBAL seed rank], labels of
    the own fit points [lo, hi) as indices into those clusters, segments among the own points), in rank order.
    Returns the whole cloud's result: clusters in the reference's order (ascending seed rank,
    z_growing.cpp:229-282), labels of all fit points as indices into that list."""
    from hope_amd.capi import CLUSTER_DTYPE
    recs = np.concatenate([np.asarray(p[2], dtype=CLUSTER_DTYPE) for p in parts]) if parts else np.zeros(0, CLUSTER_DTYPE)
    src = np.concatenate([np.full(len(p[2]), r, np.int64) for r, p in enumerate(parts)]) if parts else np.zeros(0, np.int64)
    order = np.argsort(recs["seed_rank"], kind="stable")
    assert len(set(recs["seed_rank"].tolist())) == len(recs)      # a seed belongs to exactly one band
    new_id = np.empty(len(recs), np.int64)
    new_id[order] = np.arange(len(recs))
    first = np.cumsum([0] + [len(p[2]) for p in parts])
    labels = np.full(n_fit, -1, np.int32)
    for r, (lo, hi, cl, lab, _) in enumerate(parts):
        lab = np.asarray(lab, np.int64)
        out = np.where(lab >= 0, new_id[first[r] + np.maximum(lab, 0)] if len(cl) else -1, -1)
        labels[lo:hi] = out.astype(np.int32)
    clusters = recs[order]
    return {"clusters": clusters, "labels": labels, "n_segments": int(sum(p[4] for p in parts)),
            "n_clusters": int(len(clusters)), "n_planes": int((clusters["pass"] == 1).sum())}


def whole_result(ctx):
    """The same dictionary from a ctx that holds the whole cloud's result (replicated path, one rank)."""
    c = ctx.counts(0)
    cl = ctx.clusters(0)
    return {"clusters": cl, "labels": ctx.labels(0), "n_segments": int(c.n_segments), "n_clusters": int(c.n_clusters),
            "n_planes": int((cl["pass"] == 1).sum()) if len(cl) else 0}


def _ptr(t):
    return t.data_ptr(), t.is_cuda


def _sync(t):
    """The library runs on a non-blocking stream of its own and requires a device input to be complete
    before the call (include/hope_gpu.h).  torch's collectives, torch.cat and slice copies only order
    torch's stream and never block the host, so every tensor they produce is synchronised here before
    its pointer goes into the library.  No-op for host tensors (gloo)."""
    if t.is_cuda:
        import torch
        torch.cuda.current_stream(t.device).synchronize()


def _all_gather_rows(mine, sizes, dist):
    """All-gather of row blocks of different lengths (padded to the longest), concatenated in rank order."""
    import torch
    pad = max(max(sizes), 1)
    buf = torch.zeros((pad,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    buf[:mine.shape[0]] = mine
    parts = [torch.empty_like(buf) for _ in sizes]
    dist.all_gather(parts, buf)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)]).contiguous()


HOST_SYNCS = 0          # exchanges whose result the host waited for, in the last call of segment_cloud_sharded
_HIST_CAP = 8192        # z layers a rank's histogram may span in the fused first exchange


def _exchange(values, dist, dev):
    """All-gather of a short int64 vector per rank -> numpy [world, len(values)].  The ONE place where the host waits
    for a collective: sizes shape the next allocation or the next collective's counts, so they have to come back."""
    import torch
    global HOST_SYNCS
    mine = torch.as_tensor(np.asarray(values, dtype=np.int64), device=dev)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    HOST_SYNCS += 1
    return torch.stack(parts).cpu().numpy()


def _f32_bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32).astype(np.int64)


def _bits_f32(a):
    return np.asarray(a, dtype=np.int64).astype(np.uint32).view(np.float32)


def segment_cloud_sharded(ctx, xyz_local, dist, apply_filter_transform=False, allow_bands=True):
    """xyz_local: this rank's contiguous share of the cloud, float32 torch tensor [n, 3] (CPU with
    gloo, cuda with nccl = RCCL).  Ranks hold the points in rank order (rank 0 the lowest indices).
    A6 and A8 are shared by z band.  A9..A12 run per z band too when the bands are independent (cuts at empty
    z layers; checked on the k-NN rows): every rank then grows and verifies the clusters of its own band only and
    rank 0 merges the records by seed rank -- info["path"] == "z-bands", info["result"] on rank 0 (None elsewhere).
    Otherwise (edges cross the cuts) the k-NN rows are all-gathered and A9..A12 run on every rank on the whole fit
    cloud -- info["path"] == "replicated", info["result"] on every rank, and `ctx` holds the whole result exactly
    as if one process had called segment_cloud.  info["result"] = merge_band_results' dictionary (clusters in the
    reference's order, labels of all fit points, n_segments, n_clusters, n_planes); A1..A8 accessors of `ctx` are
    those of the whole cloud on both paths.

    The host waits for five exchanges (HOST_SYNCS; info["host_syncs"]), each a short vector of sizes per rank that the
    next step cannot be shaped without: (1) survivors + bounding box + the rank's OWN z-layer histogram (binned on
    its own box: layers are floor(z / th_z), the same on every rank, so the histograms add up after a shift -- the
    global box and the global histogram come out of one exchange), (2) the routed counts of the all-to-all, (3) the
    sizes of the down-sampled pieces, (4) the fit points per band + the band-crossing check, (5) the clusters per
    band.  Payloads never touch the host.
    Returns a dict of the exchanged sizes, the path taken and the result."""
    import torch
    global HOST_SYNCS
    HOST_SYNCS = 0
    world, rank = dist.get_world_size(), dist.get_rank()
    xyz_local = xyz_local.contiguous()
    dev = xyz_local.device
    n = xyz_local.shape[0]
    ptr, on_dev = _ptr(xyz_local)
    _sync(xyz_local)
    # 1. survivors, bounding box and z-layer histogram in ONE exchange: every rank bins its points on its own box
    bbox, n_raw = ctx.shard_bbox(ptr, n, 12, apply_filter_transform, on_dev)
    own_min_bz, own_bins, own_hist = 0, 0, np.zeros(0, np.int64)
    if n_raw > 0:
        mn, dv, _ = ctx.grid_from_bbox(bbox)
        own_min_bz, own_bins = int(mn[2]), int(dv[2])
        if own_bins <= _HIST_CAP:
            own_hist = ctx.shard_z_histogram(ptr, n, 12, apply_filter_transform, on_dev, bbox, own_bins)
    fused = own_bins <= _HIST_CAP
    head = np.concatenate([[n_raw, own_min_bz, own_bins if fused else -1], _f32_bits(bbox)])
    body = np.zeros(_HIST_CAP, np.int64)
    body[:own_hist.shape[0]] = own_hist
    got = _exchange(np.concatenate([head, body]), dist, dev)
    n_raw_total = int(got[:, 0].sum())
    if n_raw_total == 0:
        ctx.segment_cloud(np.zeros((0, 3), np.float32))
        return {"n_raw_total": 0, "n_vox": 0, "sent": [0] * world, "received": [0] * world, "path": "empty",
                "result": whole_result(ctx), "host_syncs": HOST_SYNCS}
    live = got[:, 0] > 0
    boxes = np.stack([_bits_f32(got[r, 3:9]) for r in range(world)])
    gbbox = np.concatenate([boxes[live, :3].min(0), boxes[live, 3:].max(0)]).astype(np.float32)
    gmin_b, div_b, st = ctx.grid_from_bbox(gbbox)
    n_bins = int(div_b[2])
    if (got[live, 2] >= 0).all():
        hist = np.zeros(n_bins, np.int64)
        for r in np.nonzero(live)[0]:
            off, nb = int(got[r, 1]) - int(gmin_b[2]), int(got[r, 2])
            hist[off:off + nb] += got[r, 9:9 + nb]
    else:
        # a rank whose own box spans more layers than the fused exchange carries: the histogram on the global box,
        # summed by an all-reduce of its own (one more wait)
        h = torch.from_numpy(ctx.shard_z_histogram(ptr, n, 12, apply_filter_transform, on_dev, gbbox, n_bins)).to(dev)
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        HOST_SYNCS += 1
        hist = h.cpu().numpy()
    owner = band_partition(hist, world)
    # 2. route the survivors to their band owners
    routed = torch.empty((max(n_raw, 1), 3), dtype=torch.float32, device=dev)
    counts = ctx.shard_route(ptr, n, 12, apply_filter_transform, on_dev, gbbox, owner, world, routed.data_ptr(),
                             routed.is_cuda)
    sent_by = _exchange(counts, dist, dev)                      # [source, destination]
    send_l, recv_l = [int(v) for v in counts], [int(v) for v in sent_by[:, rank]]
    mine = torch.empty((max(sum(recv_l), 1), 3), dtype=torch.float32, device=dev)
    _sync(routed)
    dist.all_to_all_single(mine[:sum(recv_l)].view(-1), routed[:n_raw].view(-1), [3 * v for v in recv_l],
                           [3 * v for v in send_l])
    # 3. VoxelGrid on the band with the global grid; gather the down-sampled cloud
    _sync(mine)
    ctx.downsample_cloud_bbox(mine.data_ptr(), sum(recv_l), 12, mine.is_cuda, gbbox)
    vox_local = torch.empty((int(ctx.counts(0).n_vox), 3), dtype=torch.float32, device=dev)
    if vox_local.shape[0]:
        ctx.downsampled_to(vox_local.data_ptr(), vox_local.is_cuda)
    sizes_l = [int(v) for v in _exchange([vox_local.shape[0]], dist, dev)[:, 0]]
    vox_all = _all_gather_rows(vox_local, sizes_l, dist)
    # 4. A6..A12 on the whole down-sampled cloud; the two neighbour searches (A6, A8) are shared by z band:
    #    every rank holds all points, so its own band (a contiguous index range) needs no halo
    n_vox = int(vox_all.shape[0])
    _sync(vox_all)
    if n_vox == 0:
        ctx.segment_cloud(np.zeros((0, 3), np.float32))
        return {"n_raw_total": n_raw_total, "n_vox": 0, "sent": send_l, "received": recv_l, "bands": owner,
                "path": "empty", "result": whole_result(ctx), "host_syncs": HOST_SYNCS}
    if world == 1:
        ctx.segment_downsampled(vox_all.data_ptr(), n_vox, 12, vox_all.is_cuda, gbbox, n_raw_total)
        return {"n_raw_total": n_raw_total, "n_vox": n_vox, "sent": send_l, "received": recv_l, "bands": owner,
                "path": "one rank", "result": whole_result(ctx), "host_syncs": HOST_SYNCS}
    first, count = sum(sizes_l[:rank]), sizes_l[rank]
    args = (vox_all.data_ptr(), n_vox, 12, vox_all.is_cuda, gbbox, n_raw_total, first, count)
    ctx.segment_downsampled_staged(*args, stop_after=1)
    own_n = torch.empty((count, 4), dtype=torch.float32, device=dev)   # written by the library, host or device
    ctx.normals_rows(first, count, out_ptr=own_n.data_ptr(), out_on_device=own_n.is_cuda)
    normals_all = _all_gather_rows(own_n, sizes_l, dist)
    _sync(normals_all)
    ctx.segment_downsampled_staged(*args, normals_ptr=normals_all.data_ptr(), stop_after=2)
    fit_idx = ctx.fit_indices(0)
    lo, hi = (int(v) for v in np.searchsorted(fit_idx, [first, first + count]))
    K = ctx.cfg.knn
    own_k = torch.empty((hi - lo, K), dtype=torch.int32, device=dev)
    if hi > lo:
        ctx.knn_rows(lo, hi - lo, out_ptr=own_k.data_ptr(), out_on_device=own_k.is_cuda)
    # 5. A9..A12 by z band when the bands are independent: no valid edge (|dz| < th_z, validatePoint
    #    z_growing.cpp:335-364) leaves a band.  True by construction when the cuts sit at empty layers; checked
    #    on the k-NN rows themselves (every rank its own rows: an edge into a band is an edge out of another).
    #    The size filter is what drops the other bands' points (rows without neighbours: singletons), so a minimum
    #    cluster size of 1 takes the replicated path.
    crossing = 0 if (allow_bands and ctx.cfg.min_cluster > 1) else 1
    if crossing == 0 and hi > lo:
        crossing = int(ctx.count_band_crossings(lo, hi - lo) > 0)
    got = _exchange([hi - lo, crossing], dist, dev)
    fit_sizes_l = [int(v) for v in got[:, 0]]
    n_fit = sum(fit_sizes_l)
    base = {"n_raw_total": n_raw_total, "n_vox": n_vox, "n_fit": n_fit, "sent": send_l, "received": recv_l,
            "bands": owner, "own_vox": sizes_l, "own_fit": fit_sizes_l}
    if int(got[:, 1].sum()) == 0:
        # every band is closed under the neighbour relation: regions grow, and clusters are verified, inside the
        # own band only; the points of other bands come as rows without neighbours (singletons, dropped by the
        # size filter).  Seed ranks are those of the whole fit cloud, so rank 0 merges by seed rank.
        knn_part = torch.full((max(n_fit, 1), K), -1, dtype=torch.int32, device=dev)
        knn_part[lo:hi] = own_k
        _sync(knn_part)
        ctx.segment_downsampled_staged(*args, normals_ptr=normals_all.data_ptr(), knn_ptr=knn_part.data_ptr(), stop_after=0)
        c = ctx.counts(0)
        cl = np.ascontiguousarray(ctx.clusters(0))
        lab = np.ascontiguousarray(ctx.labels(0)[lo:hi], dtype=np.int32)
        seg_own = int(c.n_segments) - (n_fit - (hi - lo))
        meta = _exchange([len(cl), seg_own], dist, dev)
        # the records and the own labels as fixed-layout rows (no pickling)
        rec_bytes = cl.dtype.itemsize
        cl_t = torch.from_numpy(cl.view(np.uint8).reshape(len(cl), rec_bytes).copy()).to(dev)
        cl_all = _all_gather_rows(cl_t, [int(v) for v in meta[:, 0]], dist)
        lab_all = _all_gather_rows(torch.from_numpy(lab.reshape(-1, 1).copy()).to(dev), fit_sizes_l, dist)
        merged = None
        if rank == 0:
            from hope_amd.capi import CLUSTER_DTYPE
            recs = cl_all.cpu().numpy().reshape(-1).view(CLUSTER_DTYPE)
            labs = lab_all.cpu().numpy().reshape(-1)
            parts, c0, f0 = [], 0, 0
            for r in range(world):
                ncl, nf = int(meta[r, 0]), fit_sizes_l[r]
                parts.append((f0, f0 + nf, recs[c0:c0 + ncl], labs[f0:f0 + nf], int(meta[r, 1])))
                c0, f0 = c0 + ncl, f0 + nf
            merged = merge_band_results(parts, n_fit)
        base.update({"path": "z-bands", "result": merged, "host_syncs": HOST_SYNCS})
        return base
    # fallback: edges cross the band cuts (no empty layer to cut at): all-gather the k-NN rows and run A9..A12
    # on the whole fit cloud on every rank
    if n_fit == 0:
        knn_all = torch.zeros((1, K), dtype=torch.int32, device=dev)
    else:
        knn_all = _all_gather_rows(own_k, fit_sizes_l, dist)
    _sync(knn_all)
    ctx.segment_downsampled_staged(*args, normals_ptr=normals_all.data_ptr(), knn_ptr=knn_all.data_ptr(), stop_after=0)
    base.update({"path": "replicated", "result": whole_result(ctx), "host_syncs": HOST_SYNCS})
    return base
