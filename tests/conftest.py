import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the library reads its HOPE_DEBUG_* test knobs (forced paths) only when this is set
os.environ["HOPE_DEBUG"] = "1"

ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "libhope_oracle.so")
ORACLE_LIBM_LIB = os.path.join(ROOT, "oracle", "_build", "libhope_oracle_libm.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_oracle():
    if not (os.path.exists(ORACLE_LIB) and os.path.exists(ORACLE_LIBM_LIB)):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (checker).  Only tests load it."""
    from hope_amd.capi import HopeLib
    _ensure_oracle()
    return HopeLib(ORACLE_LIB, "hope_oracle_")


@pytest.fixture(scope="session")
def oracle_libm_lib():
    from hope_amd.capi import HopeLib
    _ensure_oracle()
    return HopeLib(ORACLE_LIBM_LIB, "hope_oracle_")


@pytest.fixture(scope="session")
def hip_lib():
    """The product: hope_amd/csrc/libhope_hip.so, built beforehand (never inside a test)."""
    from hope_amd.capi import HopeLib
    return HopeLib()


@pytest.fixture(scope="session")
def tum_example():
    g = np.load(os.path.join(GOLDEN, "tum_example_depth.npz"))
    return g["depth"], g["t"], g["q"]


def synth_frames(n, first=0):
    """n synthetic config-2 frames as numpy (generated on the CPU so both sides see the same bytes)."""
    from hope_amd import synth
    d, t, q = synth.make_depth_frames(n, first_frame=first, device="cpu")
    return d.numpy().view(np.uint16), t.numpy(), q.numpy()


def patch_cloud(n_patches=24, per_patch=3000, seed=3):
    """n_patches separate horizontal patches (0.6 m squares, 1.5 m apart) at z = 1 m +- 2 mm noise, in the
    sensor frame of an identity pose: one ZGrowing cluster each, all within th_z of one height, so the
    RT winner merges every one of them (plane_segment.cpp:943-955 has no limit on that)."""
    rng = np.random.default_rng(seed)
    pts = []
    for k in range(n_patches):
        ox, oy = 1.5 * (k % 6) - 3.75, 1.5 * (k // 6) - 2.25
        xy = rng.uniform(0.0, 0.6, (per_patch, 2)).astype(np.float32) + np.array([ox, oy], np.float32)
        z = (1.0 + rng.normal(0.0, 0.002, per_patch)).astype(np.float32)
        pts.append(np.concatenate([xy, z[:, None]], 1))
    xyz = np.concatenate(pts).astype(np.float32)
    return xyz[rng.permutation(len(xyz))]
