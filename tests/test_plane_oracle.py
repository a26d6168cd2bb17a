"""fillGroundPlane (PointCloudSensor.cpp:362-388): the oracle's RANSAC plane fit and ring points against
independent numpy restatements, known answers and the committed golden vectors.  No GPU needed."""
import hashlib
import json
import math
import os

import numpy as np

from conftest import GOLDEN


def plane_scene(n_plane=20000, n_out=5000, seed=0, tilt=(0.02, -0.03), z0=-1.7, sigma=0.003):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(-10, 10, (n_plane, 2))
    z = z0 + tilt[0] * xy[:, 0] + tilt[1] * xy[:, 1] + sigma * rng.standard_normal(n_plane)
    out = rng.uniform(-10, 10, (n_out, 3))
    c = np.vstack([np.c_[xy, z], out]).astype(np.float32)
    return c[rng.permutation(len(c))]


def test_sampler_generator_is_mt19937(oracle_mod):
    # boost::mt19937(12345) == the classic init_genrand seeding == numpy's legacy RandomState(12345)
    want = np.random.RandomState(12345).randint(0, 2 ** 32, 2000, dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(oracle_mod.mt19937_outputs(12345, 2000), want)


def numpy_ransac(c, threshold=0.01, max_iterations=1000, probability=0.99):
    """independent restatement (numpy float32 arithmetic, numpy's MT19937 stream)"""
    p = np.ascontiguousarray(c[:, :3], np.float32)
    n = len(p)
    rs = np.random.RandomState(12345)
    perm = np.arange(n)
    thr = np.float32(threshold)
    iters, best, k, coeffs = 0, -1, 1.0, None
    f32 = np.float32
    with np.errstate(all="ignore"):
        while iters < k:
            for _ in range(1000):
                for i in range(3):
                    j = i + int((int(rs.randint(0, 2 ** 32, dtype=np.uint64)) >> 1) % (n - i))
                    perm[i], perm[j] = perm[j], perm[i]
                p0, p1, p2 = p[perm[0]], p[perm[1]], p[perm[2]]
                r = (p1 - p0) / (p2 - p0)
                if r[0] != r[1] or r[2] != r[1]:
                    break
            a, b = p1 - p0, p2 - p0
            mc = np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], f32)
            z = f32(f32(mc[0] * mc[0] + mc[2] * mc[2]) + f32(mc[1] * mc[1]))
            mc = (mc / np.sqrt(z)).astype(f32)
            d = f32(-f32(f32(mc[0] * p0[0] + mc[2] * p0[2]) + f32(mc[1] * p0[1])))
            v = (mc[0] * p[:, 0] + mc[1] * p[:, 1]) + (mc[2] * p[:, 2] + d)
            cnt = int(np.count_nonzero(np.abs(v) < thr))
            if cnt > best:
                best, coeffs = cnt, np.array([mc[0], mc[1], mc[2], d], f32)
                pn = min(max(1.0 - (best / n) ** 3, np.finfo(float).eps), 1 - np.finfo(float).eps)
                k = math.log(1 - probability) / math.log(pn)
            iters += 1
            if iters > max_iterations:
                break
    return coeffs, best, iters


def test_ransac_matches_numpy_restatement(oracle_mod):
    for seed in (0, 1):
        c = plane_scene(seed=seed)
        ok, co, ninl, it = oracle_mod.fit_plane_ransac(c)
        co2, ninl2, it2 = numpy_ransac(c)
        assert ok and np.array_equal(co, co2) and (ninl, it) == (ninl2, it2)


def test_ransac_known_answer(oracle_mod):
    c = plane_scene(tilt=(0.02, -0.03), z0=-1.7)
    ok, co, ninl, it = oracle_mod.fit_plane_ransac(c)
    n = np.array([-0.02, 0.03, 1.0])
    n /= np.linalg.norm(n)
    s = np.sign(co[2])
    assert ok and np.allclose(s * co[:3], n, atol=2e-3) and abs(s * co[3] - 1.7 * n[2]) < 0.02
    assert ninl > 0.8 * 20000 and it < 100     # an un-refined sample plane still catches most of the 3 mm-noise ground


def test_ransac_degenerate_inputs(oracle_mod):
    assert not oracle_mod.fit_plane_ransac(np.zeros((2, 3), np.float32))[0]          # fewer than 3 points
    line = np.outer(np.arange(1, 50, dtype=np.float32), [1, 2, 4]).astype(np.float32)
    assert not oracle_mod.fit_plane_ransac(line)[0]                                  # every sample collinear
    same = oracle_mod.fill_ground_plane(line, 2.0, 0.1)
    assert np.array_equal(same, line)                                                # nothing appended


def test_ring_points_lie_on_the_plane(oracle_mod):
    c = plane_scene(seed=3)
    ok, co, _, _ = oracle_mod.fit_plane_ransac(c)
    filled = oracle_mod.fill_ground_plane(c, 5.0, 0.1)
    ring = filled[len(c):].astype(np.float64)
    assert np.array_equal(filled[:len(c)], c[:, :3])
    # loops of :373-387: r = res, 2 res, ... <= radius (double accumulation), angle < 2 pi in steps of res / radius
    rings, r = 0, 0.1
    while r <= 5.0:
        rings, r = rings + 1, r + 0.1
    per, a = 0, 0.0
    while a < 2 * 3.141592654:          # the reference's own macro, #define PI 3.141592654 (PointCloudSensor.cpp:48, :376)
        per, a = per + 1, a + 0.1 / 5.0
    assert len(ring) == rings * per
    n, d = co[:3].astype(np.float64), float(co[3])
    assert np.abs(ring @ n + d).max() < 1e-5
    # each ring is a circle about the foot of the origin on ... the axis through the origin along the normal
    axial = ring @ n
    radial = np.linalg.norm(ring - np.outer(axial, n), axis=1).reshape(rings, per)
    assert np.abs(radial - radial[:, :1]).max() < 1e-5


def test_plane_golden_replay(oracle_mod, fixture_clouds):
    g = json.load(open(os.path.join(GOLDEN, "plane_golden.json")))
    for case in g["cases"]:
        raw = fixture_clouds[case["cloud"] - 1]
        cloud = raw if case["input"] == "raw" else oracle_mod.voxel_downsample(raw, 0.2)[0]
        ok, co, ninl, it = oracle_mod.fit_plane_ransac(cloud)
        assert ok == case["found"] and [float(x).hex() for x in co] == case["coefficients_hex"]
        assert (ninl, it) == (case["n_inliers"], case["iterations"])
        ring = oracle_mod.fill_ground_plane(cloud, case["radius"], case["map_resolution"])[len(cloud):]
        assert len(ring) == case["n_ring"]
        assert hashlib.sha256(np.ascontiguousarray(ring, np.float32).tobytes()).hexdigest() == case["ring_sha256"]


PLY_POINTS = np.float32([[-1.5, 2.25, 0.125], [3, 4, 5], [1e-3, -2e3, 7.5]])
PLY_CAMERA = [0.5, -1.0, 2.0, 0, 1, 0, -1, 0, 0, 0, 0, 1]      # view_p, then the x / y / z axis


def write_test_plys(d):
    """an ASCII and a binary_little_endian PLY with extra vertex properties, a face list and a camera element"""
    import struct
    header = ["ply", "format %s 1.0", "comment test", "element vertex 3", "property float x", "property float y",
              "property float z", "property uchar intensity", "element face 1", "property list uchar int vertex_indices",
              "element camera 1", "property float view_px", "property float view_py", "property float view_pz"]
    header += ["property float %s_axis%s" % (a, c) for a in "xyz" for c in "xyz"] + ["property int viewportx", "end_header"]
    with open(os.path.join(d, "ascii.ply"), "w") as f:
        f.write("\n".join(header) % "ascii" + "\n")
        for p in PLY_POINTS:
            f.write("%r %r %r 7\n" % tuple(float(v) for v in p))
        f.write("3 0 1 2\n" + " ".join(str(v) for v in PLY_CAMERA) + " 20\n")
    with open(os.path.join(d, "binary.ply"), "wb") as f:
        f.write(("\n".join(header) % "binary_little_endian" + "\n").encode())
        for p in PLY_POINTS:
            f.write(struct.pack("<fffB", *p, 7))
        f.write(struct.pack("<Biii", 3, 0, 1, 2) + struct.pack("<12fi", *PLY_CAMERA, 20))
    with open(os.path.join(d, "bad.ply"), "w") as f:
        f.write("not a ply\n")
    return ["ascii.ply", "binary.ply"]


def test_cpp_mirror_ply_reader(tmp_path):
    """readPLY of the C++ mirror (what loadPLY, PointCloudSensor.cpp:390-417, gets from pcl::PLYReader): host only."""
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "ply_probe")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "cpp"), "-o", exe,
                           os.path.join(ROOT, "tests", "cpp_probe", "ply_probe.cpp"),
                           os.path.join(ROOT, "cpp", "slam3d", "sensor", "pcl", "PointCloudSensor.cpp"),
                           "-L" + os.path.join(ROOT, "slam3d_amd", "lib"), "-lslam3d_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "slam3d_amd", "lib"), "-lpthread"])
    for name in write_test_plys(str(tmp_path)):
        out = subprocess.check_output([exe, str(tmp_path / name)]).decode().splitlines()
        assert out[0] == "0 3"
        got = np.float32([[float.fromhex(v) for v in ln.split()] for ln in out[1:4]])
        assert np.array_equal(got, PLY_POINTS)
        T = np.array([[float(v) for v in ln.split()] for ln in out[4:7]])
        assert np.array_equal(T[:, 3], PLY_CAMERA[:3]) and np.array_equal(T[:, :3].reshape(-1), PLY_CAMERA[3:])
    assert subprocess.check_output([exe, str(tmp_path / "bad.ply")]).decode().splitlines()[0] == "-1 0"
    assert subprocess.check_output([exe, str(tmp_path / "missing.ply")]).decode().splitlines()[0] == "-1 0"
