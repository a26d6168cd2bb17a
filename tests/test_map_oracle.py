"""Patch accumulation and map building: the oracle against independent numpy /
scipy restatements, and the device per-thread code (CPU emulation) against the oracle.  No GPU needed."""
import ctypes as C

import numpy as np
import pytest
from scipy.spatial import cKDTree

from test_emu_core import emu, fp, dp  # noqa: F401  (fixture)


def rigid(rng, t_scale=5.0):
    a = rng.normal(size=3)
    a /= np.linalg.norm(a)
    th = rng.uniform(-np.pi, np.pi)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    T[:3, 3] = rng.normal(size=3) * t_scale
    return T


def test_transform_is_double_product_rounded_once(oracle_mod, fixture_clouds):
    rng = np.random.default_rng(1)
    T = rigid(rng)
    p = fixture_clouds[0][:5000, :3]
    got = oracle_mod.transform_cloud(p, T)
    x, y, z = (p[:, i].astype(np.float64) for i in range(3))
    want = np.stack([((x * T[r, 0] + y * T[r, 1]) + (z * T[r, 2] + T[r, 3])).astype(np.float32) for r in range(3)], 1)
    assert np.array_equal(got, want)
    # strided (pcl::PointXYZ) input gives the same points
    assert np.array_equal(oracle_mod.transform_cloud(fixture_clouds[0][:5000], T), got)


def test_accumulate_is_concatenation_in_vertex_order(oracle_mod, fixture_clouds):
    rng = np.random.default_rng(2)
    clouds = [c[:3000] for c in fixture_clouds[:3]]
    poses = [rigid(rng) for _ in clouds]
    acc = oracle_mod.accumulate_clouds(clouds, poses)
    want = np.concatenate([oracle_mod.transform_cloud(c, T) for c, T in zip(clouds, poses)])
    assert np.array_equal(acc, want)
    # createCombinedMeasurement: the float accumulated cloud through pose.inverse()
    frame = rigid(rng)
    inv = np.eye(4)
    inv[:3, :3] = frame[:3, :3].T
    inv[:3, 3] = -(frame[:3, :3].T @ frame[:3, 3])
    comb = oracle_mod.accumulate_clouds(clouds, poses, frame)
    ref = oracle_mod.transform_cloud(want, inv)
    assert np.abs(comb - ref).max() < 2e-6          # inverse formed with a different summation order: ulp level
    assert oracle_mod.accumulate_clouds([], []).shape == (0, 3)


@pytest.mark.parametrize("radius,min_nb", [(0.2, 3), (0.35, 8), (0.1, 1)])
def test_remove_outliers_matches_radius_count(oracle_mod, fixture_clouds, radius, min_nb):
    p = oracle_mod.voxel_downsample(fixture_clouds[1], 0.1)[0]
    got = oracle_mod.remove_outliers(p, radius, min_nb)
    # independent restatement: a point stays iff at least min_nb + 1 points (itself included) have float d2 <= r^2
    tree = cKDTree(p.astype(np.float64))
    keep = np.zeros(len(p), bool)
    for i, nb in enumerate(tree.query_ball_point(p.astype(np.float64), radius * 1.001)):
        q = p[nb]
        d = q - p[i]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]      # float32, FLANN L2_Simple order
        keep[i] = np.count_nonzero(d2.astype(np.float64) <= radius * radius) >= min_nb + 1
    assert 0 < keep.sum() < len(p)
    assert np.array_equal(got, p[keep])


def test_remove_outliers_degenerate_arguments(oracle_mod, fixture_clouds):
    p = fixture_clouds[0][:2000, :3]
    assert np.array_equal(oracle_mod.remove_outliers(p, 0.0, 3), p)       # PointCloudSensor.cpp:214
    assert np.array_equal(oracle_mod.remove_outliers(p, 0.2, 0), p)
    assert oracle_mod.remove_outliers(p[:0], 0.2, 3).shape == (0, 3)
    assert oracle_mod.remove_outliers(p[:3], 10.0, 3).shape == (0, 3)     # fewer than k points in the cloud: all go
    dup = np.repeat(p[:1], 5, 0)
    assert np.array_equal(oracle_mod.remove_outliers(dup, 0.01, 3), dup)   # duplicates count as neighbours


def test_build_map_is_the_three_stages(oracle_mod, fixture_clouds):
    rng = np.random.default_rng(3)
    clouds = [c[::4] for c in fixture_clouds]
    poses = [rigid(rng, 0.5) for _ in clouds]
    m = oracle_mod.build_map(clouds, poses, 0.2, 3, 0.1)
    acc = oracle_mod.accumulate_clouds(clouds, poses)
    want = oracle_mod.voxel_downsample(oracle_mod.remove_outliers(acc, 0.2, 3), 0.1)[0]
    assert np.array_equal(m, want) and 0 < len(m) < len(acc)


# ---- device per-thread code under g++ -----------------------------------------------------------------

def test_device_transform_bit_exact(emu, oracle_mod, fixture_clouds):  # noqa: F811
    rng = np.random.default_rng(4)
    p = np.ascontiguousarray(fixture_clouds[2][:20000, :3])
    for _ in range(3):
        T = rigid(rng, 50.0)
        out = np.empty_like(p)
        tf = oracle_mod.colmajor(T)
        emu.emu_transform(p.ctypes.data_as(fp), len(p), tf.ctypes.data_as(dp), out.ctypes.data_as(fp))
        assert np.array_equal(out, oracle_mod.transform_cloud(p, T))


@pytest.mark.parametrize("radius,min_nb,cpp", [(0.2, 3, 2), (0.2, 3, 16), (0.5, 10, 2), (0.05, 1, 2), (3.0, 200, 2)])
def test_device_radius_count_bit_exact(emu, oracle_mod, fixture_clouds, radius, min_nb, cpp):  # noqa: F811
    p = np.ascontiguousarray(oracle_mod.voxel_downsample(fixture_clouds[3], 0.1)[0])
    out = np.empty_like(p)
    emu.emu_remove_outliers.argtypes = [fp, C.c_int, C.c_double, C.c_uint, C.c_int, fp]
    m = emu.emu_remove_outliers(p.ctypes.data_as(fp), len(p), radius, min_nb, cpp, out.ctypes.data_as(fp))
    want = oracle_mod.remove_outliers(p, radius, min_nb)
    assert m == len(want) and np.array_equal(out[:m], want)


def test_map_golden_replay(oracle_mod, fixture_clouds):
    """tests/golden/map_golden.json (made by tests/golden/make_map_golden.py) replayed bit for bit."""
    import hashlib
    import json
    import os
    from conftest import GOLDEN
    g = json.load(open(os.path.join(GOLDEN, "map_golden.json")))
    poses = [np.array(p) for p in g["poses"]]

    def check(a, rec):
        assert len(a) == rec["n"]
        assert hashlib.sha256(np.ascontiguousarray(a, np.float32).tobytes()).hexdigest() == rec["sha256"]

    acc = oracle_mod.accumulate_clouds(fixture_clouds, poses)
    check(acc, g["accumulate"])
    check(oracle_mod.accumulate_clouds(fixture_clouds, poses, poses[1]), g["combined_frame1"])
    for key, rec in g["remove_outliers"].items():
        r, k = key.split("/")
        check(oracle_mod.remove_outliers(acc, float(r), int(k)), rec)
    for key, rec in g["build_map"].items():
        r, k, res = key.split("/")
        check(oracle_mod.build_map(fixture_clouds, poses, float(r), int(k), float(res)), rec)


def test_reference_map_building_case_empty_cloud(oracle_mod):
    """slam3d/sensor/pcl/PointCloudSensorTest.cpp:73-96: buildMap of one empty cloud must not fail."""
    m = oracle_mod.build_map([np.zeros((0, 4), np.float32)], [np.eye(4)])
    assert m.shape == (0, 3)
