"""GPU parity tests of the callers either side of the registration path: patch
accumulation (getAccumulatedCloud / createCombinedMeasurement), radius outlier removal and buildMap, through
the C ABI against the CPU oracle.  Everything here is float/integer work evaluated in the oracle's order:
the bar is BIT-EXACT."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, transform_delta
from test_map_oracle import rigid

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, np.float32).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(GOLDEN, "map_golden.json")))


def test_accumulate_bit_exact(gpu_ctx, oracle_mod, fixture_clouds):
    rng = np.random.default_rng(11)
    poses = [rigid(rng, 20.0) for _ in fixture_clouds]
    dev = [gpu_ctx.upload(c) for c in fixture_clouds]
    acc = gpu_ctx.accumulate(dev, poses)
    assert np.array_equal(acc.download(), oracle_mod.accumulate_clouds(fixture_clouds, poses))
    frame = rigid(rng, 20.0)
    comb = gpu_ctx.accumulate(dev, poses, frame)
    assert np.array_equal(comb.download(), oracle_mod.accumulate_clouds(fixture_clouds, poses, frame))
    # the same cloud may appear several times; packed xyz input; a single cloud
    p3 = np.ascontiguousarray(fixture_clouds[0][:, :3])
    d3 = gpu_ctx.upload(p3)
    twice = gpu_ctx.accumulate([d3, d3], poses[:2])
    assert np.array_equal(twice.download(), oracle_mod.accumulate_clouds([p3, p3], poses[:2]))
    assert np.array_equal(gpu_ctx.accumulate([d3], [np.eye(4)]).download(), p3)


def test_accumulate_edge_cases(gpu_ctx):
    assert gpu_ctx.accumulate([], []).n == 0
    e = gpu_ctx.upload(np.zeros((0, 3), np.float32))
    one = gpu_ctx.upload(np.array([[1, 2, 3]], np.float32))
    T = np.eye(4)
    T[:3, 3] = [1, 1, 1]
    out = gpu_ctx.accumulate([e, one, e], [np.eye(4), T, np.eye(4)])
    assert np.array_equal(out.download(), np.array([[2, 3, 4]], np.float32))


def test_accumulate_golden(gpu_ctx, fixture_clouds, golden):
    poses = [np.array(p) for p in golden["poses"]]
    dev = [gpu_ctx.upload(c) for c in fixture_clouds]
    a = gpu_ctx.accumulate(dev, poses).download()
    assert len(a) == golden["accumulate"]["n"] and sha(a) == golden["accumulate"]["sha256"]
    c = gpu_ctx.accumulate(dev, poses, poses[1]).download()
    assert sha(c) == golden["combined_frame1"]["sha256"]


@pytest.mark.parametrize("radius,min_nb", [(0.2, 3), (0.1, 2), (0.5, 20), (0.05, 1), (2.0, 300)])
def test_remove_outliers_bit_exact(gpu_ctx, oracle_mod, fixture_clouds, radius, min_nb):
    p = oracle_mod.voxel_downsample(fixture_clouds[2], 0.1)[0]
    want = oracle_mod.remove_outliers(p, radius, min_nb)
    got = gpu_ctx.remove_outliers(p, radius, min_nb)
    assert got.shape == want.shape and np.array_equal(got, want)
    # device-resident variant
    d = gpu_ctx.remove_outliers(gpu_ctx.upload(p), radius, min_nb)
    assert d.n == len(want) and np.array_equal(d.download(), want)


def test_remove_outliers_raw_scan_and_golden(gpu_ctx, oracle_mod, fixture_clouds, golden):
    poses = [np.array(p) for p in golden["poses"]]
    acc = oracle_mod.accumulate_clouds(fixture_clouds, poses)
    for key, rec in golden["remove_outliers"].items():
        r, k = key.split("/")
        got = gpu_ctx.remove_outliers(acc, float(r), int(k))
        assert len(got) == rec["n"] and sha(got) == rec["sha256"]


def test_remove_outliers_edge_cases(gpu_ctx, fixture_clouds):
    p = np.ascontiguousarray(fixture_clouds[0][:2000, :3])
    assert np.array_equal(gpu_ctx.remove_outliers(p, 0.0, 3), p)      # PointCloudSensor.cpp:214
    assert np.array_equal(gpu_ctx.remove_outliers(p, 0.2, 0), p)
    assert gpu_ctx.remove_outliers(p[:0], 0.2, 3).shape == (0, 3)
    assert gpu_ctx.remove_outliers(p[:3], 10.0, 3).shape == (0, 3)    # fewer than k points: all removed
    dup = np.repeat(p[:1], 5, 0)
    assert np.array_equal(gpu_ctx.remove_outliers(dup, 0.01, 3), dup)  # duplicates are neighbours
    far = np.array([[0, 0, 0], [100, 0, 0], [0, 100, 0], [0.05, 0, 0]], np.float32)
    assert np.array_equal(gpu_ctx.remove_outliers(far, 0.1, 1), far[[0, 3]])


def test_voxel_downsample_cloud_equals_host_entry(gpu_ctx, oracle_mod, fixture_clouds):
    c = fixture_clouds[1]
    want = oracle_mod.voxel_downsample(c, 0.15)[0]
    got = gpu_ctx.voxel_downsample_cloud(gpu_ctx.upload(c), 0.15)
    assert got.n == len(want) and np.array_equal(got.download(), want)


def test_build_map_bit_exact_and_golden(gpu_ctx, oracle_mod, fixture_clouds, golden):
    poses = [np.array(p) for p in golden["poses"]]
    dev = [gpu_ctx.upload(c) for c in fixture_clouds]
    for key, rec in golden["build_map"].items():
        r, k, res = key.split("/")
        m = gpu_ctx.build_map(dev, poses, float(r), int(k), float(res))
        got = m.download()
        assert len(got) == rec["n"] and sha(got) == rec["sha256"]
        prof = gpu_ctx.last_map_profile()
        assert prof["n_accumulated"] == golden["accumulate"]["n"] and prof["n_map"] == rec["n"]
    rng = np.random.default_rng(5)
    poses = [rigid(rng, 3.0) for _ in fixture_clouds]
    want = oracle_mod.build_map(fixture_clouds, poses, 0.25, 4, 0.2)
    assert np.array_equal(gpu_ctx.build_map(dev, poses, 0.25, 4, 0.2).download(), want)


def test_build_map_reference_case_empty_cloud(gpu_ctx):
    """slam3d/sensor/pcl/PointCloudSensorTest.cpp:73-96 (map_building): one vertex, empty cloud, no throw."""
    e = gpu_ctx.upload(np.zeros((0, 4), np.float32))
    assert gpu_ctx.build_map([e], [np.eye(4)]).n == 0
    assert gpu_ctx.build_map([], []).n == 0


def test_create_constraint_on_device_patches(gpu_ctx, oracle_mod, fixture_clouds):
    """ScanSensor::link (ScanSensor.cpp:143-166): two patches built on the device, registered without a host
    round trip == the host entry point fed with the oracle's patches."""
    import slam3d_amd as s3d
    rng = np.random.default_rng(9)
    small = [rigid(rng, 0.05) for _ in range(4)]
    for T in small:   # nearly-aligned scans: the patch is a slightly thickened scan
        T[:3, :3] = np.eye(3)
    dev = [gpu_ctx.upload(c) for c in fixture_clouds]
    src = gpu_ctx.accumulate(dev[:2], small[:2], small[0])
    tgt = gpu_ctx.accumulate(dev[2:], small[2:], small[2])
    src_h = oracle_mod.accumulate_clouds(fixture_clouds[:2], small[:2], small[0])
    tgt_h = oracle_mod.accumulate_clouds(fixture_clouds[2:], small[2:], small[2])
    assert np.array_equal(src.download(), src_h) and np.array_equal(tgt.download(), tgt_h)
    fine = s3d.default_params(registration_algorithm=s3d.ALG_ICP)
    coarse = s3d.default_params(registration_algorithm=s3d.ALG_ICP, point_cloud_density=0.5,
                                max_correspondence_distance=5.0)
    odo = np.eye(4)
    odo[0, 3] = 1.0
    st_d, rel_d, inf_d, info_d = gpu_ctx.create_constraint_clouds(src, np.eye(4), tgt, np.eye(4), odo, True, fine, coarse)
    st_h, rel_h, inf_h, info_h = gpu_ctx.create_constraint(src_h, np.eye(4), tgt_h, np.eye(4), odo, True, fine, coarse)
    assert st_d == st_h
    assert np.array_equal(rel_d, rel_h) and np.array_equal(inf_d, inf_h) and info_d == info_h
    # and against the oracle on the same patches (point-to-plane mode: 1e-4 m / 1e-4 rad bar)
    op = oracle_mod.default_params(registration_algorithm=oracle_mod.ALG_ICP)
    oc = oracle_mod.default_params(registration_algorithm=oracle_mod.ALG_ICP, point_cloud_density=0.5,
                                   max_correspondence_distance=5.0)
    st_o, rel_o, _, _ = oracle_mod.create_constraint(src_h, np.eye(4), tgt_h, np.eye(4), odo, True, op, oc)
    assert st_o == st_d
    if st_o == 0:
        dt, dr = transform_delta(rel_o, rel_d)
        assert dt < 1e-4 and dr < 1e-4


def test_map_of_ten_million_points_properties(gpu_ctx):
    """BASELINE-scale map (96 scans x 100k points: the wide-grid path with > 2^24 cells): size-independent
    properties — order-preserving subset, exact neighbour counts on a sample, one point per occupied voxel."""
    import slam3d_amd as s3d
    from scipy.spatial import cKDTree
    from multiprocessing.pool import ThreadPool
    n_scans = 96

    def scan(i):   # 98k surface points + 2k floating outliers above the scene
        rng = np.random.default_rng(7000 + i)
        air = rng.uniform([-40, -8, 5], [40, 8, 30], size=(2000, 3)).astype(np.float32)
        return np.concatenate([s3d.make_scene_cloud(98000, 5000 + i), air])

    with ThreadPool(8) as pool:
        clouds = pool.map(scan, range(n_scans))
    poses = []
    for i in range(n_scans):
        T = np.eye(4)
        T[:3, 3] = [0.8 * i, 0.3 * (i % 7), 0.0]
        poses.append(T)
    dev = [gpu_ctx.upload(c) for c in clouds]
    acc = gpu_ctx.accumulate(dev, poses)
    A = acc.download()
    assert len(A) == n_scans * 100000
    # accumulate: spot-check three scans against numpy in double
    for i in (0, 37, 95):
        c = clouds[i][:, :3].astype(np.float64)
        want = (c @ poses[i][:3, :3].T + poses[i][:3, 3]).astype(np.float32)
        assert np.abs(A[i * 100000:(i + 1) * 100000] - want).max() <= 4e-6
    radius, min_nb = 0.2, 3
    kept = gpu_ctx.remove_outliers(acc, radius, min_nb)
    K = kept.download()
    assert 0 < len(K) < len(A)
    # order-preserving subset: K == A[mask].  (Bit-identical twins share their verdict, so membership is a mask.)
    Av = np.ascontiguousarray(A).view(np.dtype((np.void, 12))).ravel()
    Kv = np.ascontiguousarray(K).view(np.dtype((np.void, 12))).ravel()
    mask = np.isin(Av, Kv)
    assert np.array_equal(A[mask], K)
    # exact verdict on a sample: float d2 in FLANN order, compared as PCL does
    tree = cKDTree(A.astype(np.float64))
    rng = np.random.default_rng(0)
    sample = rng.choice(len(A), 4000, replace=False)
    keep_ref = np.zeros(len(sample), bool)
    for s, (i, nb) in enumerate(zip(sample, tree.query_ball_point(A[sample].astype(np.float64), radius * 1.001))):
        d = A[nb] - A[i]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        keep_ref[s] = np.count_nonzero(d2.astype(np.float64) <= radius * radius) >= min_nb + 1
    assert np.array_equal(mask[sample], keep_ref)
    # map: one centroid per occupied voxel of the kept cloud
    m = gpu_ctx.build_map(dev, poses, radius, min_nb, 0.1)
    M = m.download()
    prof = gpu_ctx.last_map_profile()
    assert prof["n_accumulated"] == len(A) and prof["n_kept"] == len(K) and prof["n_map"] == len(M)
    inv = np.float32(1.0) / np.float32(0.1)
    ijk = np.floor(K * inv).astype(np.int64)
    n_vox = len(np.unique(ijk, axis=0))
    assert len(M) == n_vox
    mjk = np.floor(M * inv).astype(np.int64)
    assert len(np.unique(mjk, axis=0)) >= int(0.999 * n_vox)   # centroids stay in their voxel up to float rounding


def test_cpp_mirror_build_map_and_patch_link(gpu_ctx, oracle_mod, fixture_clouds, golden, tmp_path):
    """slam3d::PointCloudSensor::buildMap / getAccumulatedCloud / createCombinedMeasurement + createConstraint through
    the C++ mirror (cpp/example_build_map.cpp) == the golden map and the C ABI called from Python."""
    import subprocess
    import slam3d_amd as s3d
    from conftest import ROOT
    exe = os.path.join(ROOT, "cpp", "example_build_map")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "cpp")])
    poses = [np.array(p) for p in golden["poses"]]
    with open(tmp_path / "poses.txt", "w") as f:
        for T in poses:
            f.write(" ".join("%.17g" % v for v in T.reshape(-1)) + "\n")
    files = []
    for i, c in enumerate(fixture_clouds):
        fn = tmp_path / ("scan%d.bin" % i)
        c.astype(np.float32).tofile(fn)
        files.append(str(fn))
    out = subprocess.check_output([exe, str(tmp_path / "poses.txt"), *files], stderr=subprocess.DEVNULL).decode().splitlines()
    assert out[0].startswith("MAP ")
    n_map, sx, sy, sz = out[0].split()[1:]
    rec = golden["build_map"]["0.2/3/0.10"]   # the sensor's constructor defaults (PointCloudSensor.cpp:179-182)
    want = oracle_mod.build_map(fixture_clouds, poses, 0.2, 3, 0.1)
    assert int(n_map) == rec["n"] == len(want)
    assert np.allclose([float(sx), float(sy), float(sz)], want.astype(np.float64).sum(0), rtol=1e-9)
    assert out[1] == "ACCU %d" % golden["accumulate"]["n"]
    # the loop-closure link of the two device-resident patches == the same call from Python
    dev = [gpu_ctx.upload(c) for c in fixture_clouds]
    pa = gpu_ctx.accumulate(dev[:2], poses[:2], poses[0])
    pb = gpu_ctx.accumulate(dev[2:], poses[2:], poses[2])
    assert out[2] == "PATCH %d %d" % (pa.n, pb.n)
    fine = s3d.default_params(registration_algorithm=s3d.ALG_ICP)
    coarse = s3d.default_params(registration_algorithm=s3d.ALG_ICP, point_cloud_density=0.5,
                                max_correspondence_distance=5.0)
    guess = np.linalg.inv(poses[0]) @ poses[2]
    st, rel, _, _ = gpu_ctx.create_constraint_clouds(pa, np.eye(4), pb, np.eye(4), guess, True, fine, coarse)
    if st == 0:
        assert out[3] == "OK SE(3)"
        T_cpp = np.array([[float(x) for x in line.split()] for line in out[4:8]])
        assert np.allclose(T_cpp, rel, atol=1e-9)
    else:
        assert out[3].startswith("NoMatch")


def test_pose_graph_loop_closure_sweep(gpu_ctx):
    """a pose graph (drifted trajectory around the synthetic scene) -> linkToNeighbors candidates
    -> device-resident patches -> batched coarse + fine registration.  The batch must equal the one-by-one
    createConstraint on the same patches bit for bit, and recover the true relative poses."""
    import slam3d_amd as s3d
    from slam3d_amd.posegraph import LinkPolicy, PoseGraph, build_patch, register_links, sweep_candidates
    n = 24
    rng = np.random.default_rng(5)
    g = PoseGraph()
    truth = {}
    for i in range(n):   # two laps of a small circle inside the canyon: lap 2 revisits lap 1
        a = 4 * np.pi * i / n
        T = np.eye(4)
        T[:2, :2] = [[np.cos(0.2 * a), -np.sin(0.2 * a)], [np.sin(0.2 * a), np.cos(0.2 * a)]]
        T[:3, 3] = [3.0 * np.cos(a), 2.0 * np.sin(a), 0.0]
        truth[i] = T
        world = s3d.make_scene_cloud(30000, 900 + i).astype(np.float64)
        Ti = np.linalg.inv(T)
        local = (world @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
        drift = np.eye(4)
        drift[:3, 3] = rng.normal(0, 0.08, 3) * (i > 0)
        g.add_vertex(i, T @ drift, "velodyne", gpu_ctx.upload(local))
    for i in range(n - 1):
        g.add_edge(i, i + 1)
    pol = LinkPolicy(neighbor_radius=1.2, max_neighbor_links=2, min_loop_length=6, patch_building_range=1)
    pairs = sweep_candidates(g, pol)
    assert 4 <= len(pairs) <= 2 * n
    fine = s3d.default_params(registration_algorithm=s3d.ALG_ICP, point_cloud_density=0.1)
    coarse = s3d.default_params(registration_algorithm=s3d.ALG_ICP, point_cloud_density=0.4,
                                max_correspondence_distance=5.0)
    rec, status = register_links(gpu_ctx, g, pairs, pol, fine, coarse)
    assert sum(1 for st in status if st == 0) >= len(pairs) // 2
    errs = []
    for (s, t), r, st in zip(pairs[:6], rec, status):
        ps, _ = build_patch(gpu_ctx, g, s, pol)
        pt, _ = build_patch(gpu_ctx, g, t, pol)
        st1, rel, _, _ = gpu_ctx.create_constraint_clouds(ps, np.eye(4), pt, np.eye(4), g.get_transform(s, t), True,
                                                          fine, coarse)
        assert st1 == st
        if st == 0:
            assert np.array_equal(s3d.api.record_transform(r), rel)
            # the patches were assembled with the DRIFTED neighbour poses, so the truth is recovered only to the
            # drift level of the patch members; the source/target scans themselves dominate
            dt, dr = transform_delta(np.linalg.inv(truth[s]) @ truth[t], rel)
            dg, _ = transform_delta(np.linalg.inv(truth[s]) @ truth[t], g.get_transform(s, t))
            errs.append((dt, dg))
            assert dt < 0.25 and dr < 0.05
    errs = np.array(errs)
    assert len(errs) >= 2 and errs[:, 0].mean() < errs[:, 1].mean()   # the links correct the drifted graph estimate


def test_upload_any_stride(gpu_ctx):
    """s3d_cloud_upload takes records of >= 3 floats: packed xyz, PCL's 16-byte points, wider records (x, y, z first)."""
    import ctypes as C
    import slam3d_amd.api as api
    rng = np.random.default_rng(5)
    for stride in (3, 4, 5, 6, 8, 11):
        for n in (1, 255, 4097):
            rec = rng.normal(size=(n, stride)).astype(np.float32)
            h = C.c_void_p()
            st = gpu_ctx._L.s3d_cloud_upload(gpu_ctx._h, rec.ctypes.data_as(C.POINTER(C.c_float)), n, stride, C.byref(h))
            assert st == 0
            cloud = api.Cloud(gpu_ctx, h, n)
            assert np.array_equal(cloud.download(), rec[:, :3])
            cloud.release()
