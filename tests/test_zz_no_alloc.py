"""The no-allocation entry point hope_shard_z_histogram_static (no counterpart in the reference; for callers
that cannot hipMalloc: graph capture, tight serving loops): its bins equal hope_shard_z_histogram's, bit for bit,
on two calls in a row, the second with a smaller input -- the static scratch array keeps the first call's
contents, so whatever must start at zero has to be cleared by the call itself."""
import numpy as np
import pytest
import torch  # noqa: F401  (before the library is loaded: the process then holds one HIP runtime, torch's)

from hope_amd.capi import HopeContext, HopeError

pytestmark = pytest.mark.gpu


def _cloud(n, seed):
    from hope_amd import synth
    xyz = synth.make_multilevel_cloud(n, first_index=seed * 1000003, device="cpu").numpy()
    xyz[::997, 2] = np.nan
    return xyz


def test_static_z_histogram_equals_the_allocating_one(hip_lib):
    ctx = HopeContext(hip_lib)
    ctx.set_pose(np.float32([0.2, -0.1, 0.4]), np.float32([0.01, -0.02, 0.3, 0.95]))
    for call, (n, apply) in enumerate(((300000, False), (70000, True))):   # the second call: smaller, other settings
        xyz = _cloud(n, call)
        if apply:
            xyz = xyz * np.float32(0.2) + np.float32([0, 0, 0.4])           # sensor z into the 0.3 .. 8 m window
        dev = torch.from_numpy(xyz).cuda()
        torch.cuda.synchronize()
        bbox, n_raw = ctx.shard_bbox(dev.data_ptr(), n, 12, apply, True)
        assert 0 < n_raw <= n
        _, div_b, st = ctx.grid_from_bbox(bbox)
        n_bins = int(div_b[2])
        assert st == 0 and 10 < n_bins <= 8192
        want = ctx.shard_z_histogram(dev.data_ptr(), n, 12, apply, True, bbox, n_bins)
        assert want.sum() == n_raw
        free_before = torch.cuda.mem_get_info()[0]
        got_host = ctx.shard_z_histogram_static(dev.data_ptr(), n, 12, apply, bbox, n_bins)
        assert torch.cuda.mem_get_info()[0] == free_before      # (hipMemGetInfo: the call took no device memory)
        out = torch.full((n_bins + 8,), -7, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ctx.shard_z_histogram_static(dev.data_ptr(), n, 12, apply, bbox, n_bins, out_ptr=out.data_ptr())
        got_dev = out.cpu().numpy()
        assert np.array_equal(got_host, want)
        assert np.array_equal(got_dev[:n_bins], want) and np.all(got_dev[n_bins:] == -7)
    # sizes beyond the static buffer are refused with a clear error, and nothing is written
    with pytest.raises(HopeError, match="8192"):
        ctx.shard_z_histogram_static(dev.data_ptr(), n, 12, apply, bbox, 8193)
    # an empty input gives empty bins (the bins of the call before must not show through)
    empty = ctx.shard_z_histogram_static(dev.data_ptr(), 0, 12, apply, bbox, n_bins)
    assert empty.sum() == 0
