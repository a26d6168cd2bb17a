import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def fixture_clouds():
    """The reference's test/cloud1..4.bin (KITTI-layout float32 x,y,z,intensity), committed as
    tests/golden/cloud*.bin.npz so the GPU box (which has no /root/reference) can read them."""
    out = []
    for i in range(1, 5):
        d = np.load(os.path.join(GOLDEN, "cloud%d.npz" % i))
        out.append(d["xyzi"].astype(np.float32))
    return out


@pytest.fixture(scope="session")
def gpu_ctx():
    import slam3d_amd
    ctx = slam3d_amd.Context(0)   # raises BackendError when the HIP extension / device is missing
    yield ctx
    ctx.close()


def transform_delta(A, B):
    d = np.linalg.inv(A) @ B
    return float(np.linalg.norm(d[:3, 3])), float(np.arccos(np.clip((np.trace(d[:3, :3]) - 1) / 2, -1, 1)))
