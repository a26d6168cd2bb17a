"""GPU parity of fillGroundPlane (PointCloudSensor.cpp:362-388) through the C ABI: the RANSAC plane (inlier
counting on the device, kPlaneBatch hypotheses per pass) and the ring points against the CPU oracle.
Counts are integers and the plane comes from the same float expressions: the bar is BIT-EXACT."""
import hashlib
import json
import os

import numpy as np
import pytest

import slam3d_amd
from conftest import GOLDEN
from test_plane_oracle import plane_scene, write_test_plys

pytestmark = pytest.mark.gpu


def same_fit(fit, oracle_fit):
    ok, co, ninl, it = oracle_fit
    return (fit["found"] == ok and np.array_equal(fit["coefficients"], co) and fit["n_inliers"] == ninl
            and fit["iterations"] == it)


def test_fit_plane_bit_exact(gpu_ctx, oracle_mod):
    for seed in range(4):
        c = plane_scene(seed=seed, n_plane=20000 + 777 * seed, n_out=3000 * seed + 5)
        fit = gpu_ctx.fit_plane(c)
        assert same_fit(fit, oracle_mod.fit_plane_ransac(c)), seed
        assert fit["hypotheses_scored"] >= fit["iterations"]
    # other thresholds / iteration caps / probabilities, packed xyz input
    c = np.ascontiguousarray(plane_scene(seed=9, sigma=0.02)[:, :3])
    for thr, cap, prob in ((0.05, 1000, 0.99), (0.002, 40, 0.99), (0.01, 0, 0.5), (0.01, 1000, 0.999999)):
        assert same_fit(gpu_ctx.fit_plane(c, thr, cap, prob), oracle_mod.fit_plane_ransac(c, thr, cap, prob)), thr


def test_fit_plane_on_the_synthetic_street(gpu_ctx, oracle_mod):
    c = slam3d_amd.make_scene_cloud(100000, 5)
    fit = gpu_ctx.fit_plane(c)
    assert same_fit(fit, oracle_mod.fit_plane_ransac(c))
    # the ground of the scene is z = -1.7
    s = np.sign(fit["coefficients"][2])
    assert abs(abs(fit["coefficients"][2]) - 1) < 1e-3 and abs(s * fit["coefficients"][3] - 1.7) < 0.02


def test_fit_plane_degenerate(gpu_ctx, oracle_mod):
    assert not gpu_ctx.fit_plane(np.zeros((2, 3), np.float32))["found"]
    assert not gpu_ctx.fit_plane(np.zeros((0, 3), np.float32))["found"]
    line = np.outer(np.arange(1, 50, dtype=np.float32), [1, 2, 4]).astype(np.float32)
    assert not gpu_ctx.fit_plane(line)["found"]
    with pytest.raises(ValueError):
        gpu_ctx.fill_ground_plane(line, 2.0, 0.1)          # S3D_STATUS_TOO_FEW_POINTS: no plane
    # three points: exactly one hypothesis, all three are inliers
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0.5]], np.float32)
    assert same_fit(gpu_ctx.fit_plane(tri), oracle_mod.fit_plane_ransac(tri))


def test_fill_ground_plane_golden(gpu_ctx, oracle_mod, fixture_clouds):
    g = json.load(open(os.path.join(GOLDEN, "plane_golden.json")))
    for case in g["cases"]:
        raw = fixture_clouds[case["cloud"] - 1]
        cloud = raw if case["input"] == "raw" else gpu_ctx.voxel_downsample(raw, 0.2)
        fit = gpu_ctx.fit_plane(cloud)
        assert [float(x).hex() for x in fit["coefficients"]] == case["coefficients_hex"]
        assert (fit["n_inliers"], fit["iterations"]) == (case["n_inliers"], case["iterations"])
        filled = gpu_ctx.fill_ground_plane(cloud, case["radius"], case["map_resolution"])
        ring = filled[len(cloud):]
        assert np.array_equal(filled[:len(cloud)], cloud[:, :3]) and len(ring) == case["n_ring"]
        assert hashlib.sha256(np.ascontiguousarray(ring, np.float32).tobytes()).hexdigest() == case["ring_sha256"]


def test_fill_ground_plane_large_map(gpu_ctx, oracle_mod):
    # a 4 M-point map: 1000 hypotheses in 32 passes; count cross-checked with numpy on the returned plane
    rng = np.random.default_rng(2)
    c = np.vstack([slam3d_amd.make_scene_cloud(1000000, s) + np.float32([40 * s, 0, 0]) for s in range(4)])
    c = c[rng.permutation(len(c))]
    fit = gpu_ctx.fit_plane(c)
    co = fit["coefficients"]
    v = (co[0] * c[:, 0] + co[1] * c[:, 1]) + (co[2] * c[:, 2] + co[3])
    assert fit["found"] and fit["n_inliers"] == int(np.count_nonzero(np.abs(v) < np.float32(0.01)))


def test_cpp_mirror_fill_ground_plane_and_load_ply(gpu_ctx, oracle_mod, fixture_clouds, tmp_path):
    """slam3d::PointCloudSensor::fillGroundPlane / loadPLY through the C++ mirror (cpp/example_ground_plane.cpp)."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "cpp", "example_ground_plane")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "cpp")])
    scan = fixture_clouds[1]
    scan.astype(np.float32).tofile(tmp_path / "scan.bin")
    plys = write_test_plys(str(tmp_path))
    for ply in plys:
        out = subprocess.check_output([exe, str(tmp_path / "scan.bin"), "5.0", str(tmp_path / "ring.bin"), str(tmp_path / ply)],
                                      stderr=subprocess.DEVNULL).decode().splitlines()
        want = oracle_mod.fill_ground_plane(scan, 5.0, 0.1)     # the mirror's default map resolution is 0.1 (:180)
        assert out[0] == "FILLED %d %d" % (len(scan), len(want))
        ring = np.fromfile(tmp_path / "ring.bin", np.float32).reshape(-1, 3)
        assert np.array_equal(ring, want[len(scan):])
        assert out[1] == "PLY 3 0.5 -1 2 1"
    # a file that is not a PLY: "Could not load initial map." and no measurement
    r = subprocess.run([exe, str(tmp_path / "scan.bin"), "1.0", str(tmp_path / "ring.bin"), str(tmp_path / "bad.ply")],
                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    assert r.returncode == 1 and r.stdout.decode().splitlines()[-1] == "PLY failed"
