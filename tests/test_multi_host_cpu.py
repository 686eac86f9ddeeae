"""The C++ multi-GPU host (hope_amd/host/multi_gpu.cpp: the orchestration `hope_node <file> ply` runs over RCCL when
HOPE_NODE_GPUS=N) on 2 and 3 ranks, with threads and host buffers as the collective backend and the CPU oracle as the
library (tests/multi_driver.cpp).  Result == one oracle call on the whole cloud, on the z-band path (scene with empty
layers), the replicated fallback (ramp: no empty layer; bands switched off) and with the sensor-frame range filter."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from test_sharding_cpu import _cloud, _layer_cloud, _ramp_cloud

from hope_amd.capi import HopeContext

DRIVER = os.path.join(ROOT, "tests", "_build", "multi_driver_oracle")


@pytest.fixture(scope="module")
def driver(oracle_lib):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests"), "_build/multi_driver_oracle"])
    return DRIVER


def run(driver, xyz, tmp_path, world, apply=0, allow=1):
    f = tmp_path / "cloud.f32"
    np.ascontiguousarray(xyz, np.float32).tofile(f)
    r = subprocess.run([driver, str(f), str(world), str(apply), str(allow)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


def expect(oracle_lib, xyz, apply=False):
    c = HopeContext(oracle_lib)
    c.segment_cloud(xyz, apply_filter_transform=apply)
    k = c.counts(0)
    cl = c.clusters(0)
    h = 1469598103934665603
    for l in c.labels(0).astype(np.int64):
        h = ((h ^ (int(l) & 0xffffffff)) * 1099511628211) & 0xffffffffffffffff
    return k, cl, "%x" % h


def check(got, want):
    k, cl, fnv = want
    assert (got["n_raw"], got["n_vox"], got["n_fit"], got["n_segments"]) == (k.n_raw, k.n_vox, k.n_fit, k.n_segments)
    assert len(got["clusters"]) == len(cl) and got["n_planes"] == int((cl["pass"] == 1).sum())
    for g, w in zip(got["clusters"], cl):
        assert (g[0], g[1], g[7]) == (int(w["size"]), int(w["pass"]), int(w["seed_rank"]))
        assert np.array_equal(np.float32(g[2:7]), np.float32([w["z_mean"], w["min_x"], w["min_y"], w["max_x"], w["max_y"]]))
    assert got["n_labels"] == k.n_fit and got["labels_fnv"] == fnv


@pytest.mark.parametrize("world", [2, 3])
def test_layered_scene_takes_the_z_band_path(driver, oracle_lib, tmp_path, world):
    xyz = _layer_cloud()
    got = run(driver, xyz, tmp_path, world)
    assert got["path"] == "z-bands" and sum(got["received"]) > 0
    check(got, expect(oracle_lib, xyz))
    # the same scene with the bands switched off: the replicated path, the same result
    rep = run(driver, xyz, tmp_path, world, allow=0)
    assert rep["path"] == "replicated"
    check(rep, expect(oracle_lib, xyz))


def test_ramp_falls_back_and_the_range_filter_is_applied(driver, oracle_lib, tmp_path):
    xyz = _ramp_cloud()
    got = run(driver, xyz, tmp_path, 2)
    assert got["path"] == "replicated"
    check(got, expect(oracle_lib, xyz))
    xyz = _cloud() * np.float32(0.25) + np.array([0, 0, 0.2], np.float32)
    got = run(driver, xyz, tmp_path, 3, apply=1)
    check(got, expect(oracle_lib, xyz, apply=True))
    one = run(driver, xyz, tmp_path, 1, apply=1)
    assert one["path"] == "one rank"
    check(one, expect(oracle_lib, xyz, apply=True))


def test_a_rank_that_fails_does_not_leave_the_others_waiting(driver, tmp_path):
    """A library step that fails on ONE rank (fault injection: step k of rank 1 reports HOPE_E_NOMEM), or a rank whose
    backend throws: every rank comes back -- the failing one with its own status, the others with HOPE_E_PEER (-5)
    at the next exchange of sizes, or through the aborted thread group -- instead of waiting in a collective."""
    xyz = _layer_cloud()
    path = os.path.join(str(tmp_path), "cloud.f32")
    xyz.astype(np.float32).tofile(path)
    for step in (1, 3, 4, 5, 7, 9, 12):
        r = subprocess.run([driver, path, "3", "0", "1", "1", str(step)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 3, (step, r.returncode, r.stderr)
        # (the others report HOPE_E_PEER, or an error of their own when the failed rank's empty piece of an exchange
        #  reached a library step first: the oracle checks the normals it is handed)
        assert "rank 1: -3" in r.stderr and "rank 0: -" in r.stderr and "rank 2: -" in r.stderr, (step, r.stderr)
        if step in (1, 3, 4, 5):
            assert "rank 0: -5" in r.stderr and "rank 2: -5" in r.stderr, (step, r.stderr)
    r = subprocess.run([driver, path, "3", "0", "1", "2", "0", "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "rank 2: -101 injected exception" in r.stderr, r.stderr
    assert "rank 0:" in r.stderr and "rank 1:" in r.stderr
