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


def rotation_angle(R):
    """Angle of a (nearly) rotation matrix, well-conditioned near the identity.  The results under test are
    float-rounded 4x4 matrices (PCL's Matrix4f final transformation, widened without re-orthonormalisation,
    PointCloudSensor.cpp:80-81): their trace is off by ~1e-8, and arccos((trace - 1) / 2) turns THAT into up to
    ~2e-4 rad of pure measurement noise.  The skew part carries the angle linearly: atan2(|skew| / 2, (tr - 1) / 2)."""
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2.0
    return float(np.arctan2(np.linalg.norm(w), (np.trace(R) - 1.0) / 2.0))


def transform_delta(A, B):
    d = np.linalg.inv(A) @ B
    return float(np.linalg.norm(d[:3, 3])), rotation_angle(d[:3, :3])
