"""Host-side check of the DEVICE math (slam3d_amd/csrc/s3d_core.h) in a GPU-less container.

tests/emu/emu_pipeline.cpp runs the per-thread functions the HIP kernels call (voxel keys, grid
1-NN / k-NN ring search, covariance -> normal, Mahalanobis, the 73-term GICP quadratic form, the
BFGS controller, the point-to-plane solve) sequentially under g++.  It is a debugging aid for the
kernels' arithmetic, not a product back-end."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, transform_delta

EMU_DIR = os.path.join(ROOT, "tests", "emu")
fp, dp, ip = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int)


class EmuInfo(C.Structure):
    _fields_ = [("n_source_filtered", C.c_int), ("n_target_filtered", C.c_int), ("iterations", C.c_int),
                ("converged", C.c_int), ("correspondences", C.c_int), ("fitness", C.c_double),
                ("inner_total", C.c_int), ("evals_total", C.c_int)]


@pytest.fixture(scope="module")
def emu():
    so = os.path.join(EMU_DIR, "libs3d_emu.so")
    src = os.path.join(EMU_DIR, "emu_pipeline.cpp")
    core = os.path.join(ROOT, "slam3d_amd", "csrc", "s3d_core.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(core)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, src])
    L = C.CDLL(so)
    L.emu_set_perturb.argtypes = [C.c_double]
    return L


def emu_align(emu, oracle_mod, s, t, guess=np.eye(4), params=None, force=0, cpp=16):
    s = np.ascontiguousarray(s, np.float32)
    t = np.ascontiguousarray(t, np.float32)
    g = oracle_mod.colmajor(guess)
    res = np.empty(16)
    info = EmuInfo()
    params = params or oracle_mod.default_params()
    st = emu.emu_align(s.ctypes.data_as(fp), s.shape[0], s.shape[1], t.ctypes.data_as(fp), t.shape[0], t.shape[1],
                       g.ctypes.data_as(dp), C.byref(params), force, cpp, res.ctypes.data_as(dp), C.byref(info))
    return st, oracle_mod.from_colmajor(res), {k: getattr(info, k) for k, _ in EmuInfo._fields_}


def test_device_voxel_keys_bit_exact(emu, oracle_mod, fixture_clouds):
    for leaf in (0.1, 0.2, 1.0):
        ref, _ = oracle_mod.voxel_downsample(fixture_clouds[0], leaf)
        out = np.empty((len(fixture_clouds[0]), 3), np.float32)
        n = emu.emu_voxel(fixture_clouds[0].ctypes.data_as(fp), len(fixture_clouds[0]), 4, C.c_double(leaf),
                          out.ctypes.data_as(fp))
        assert n == len(ref) and np.array_equal(out[:n], ref)


@pytest.mark.parametrize("h0,cpp", [(0.4, 16), (0.1, 4), (2.0, 64)])
def test_device_grid_nn_is_exact(emu, oracle_mod, fixture_clouds, h0, cpp):
    v1, _ = oracle_mod.voxel_downsample(fixture_clouds[0], 0.2)
    v2, _ = oracle_mod.voxel_downsample(fixture_clouds[1], 0.2)
    rng = np.random.default_rng(1)
    far = rng.uniform(-150, 150, (500, 3)).astype(np.float32)      # queries outside the target's bbox
    q = np.ascontiguousarray(np.concatenate([v2[::3], far]))
    idx = np.empty(len(q), np.int32)
    d2 = np.empty(len(q), np.float32)
    emu.emu_nn(v1.ctypes.data_as(fp), len(v1), q.ctypes.data_as(fp), len(q), C.c_float(h0), cpp, C.c_float(2.5),
               idx.ctypes.data_as(ip), d2.ctypes.data_as(fp))
    oi, od = oracle_mod.nn_search(v1, q)
    m = od < 2.5 ** 2
    assert m.sum() > 9000
    assert np.array_equal(idx[m], oi[m]) and np.array_equal(d2[m], od[m])
    # beyond max_d the search may stop early, but it never reports a point closer than the gate wrongly
    assert np.all((idx[~m] == -1) | (d2[~m] >= 2.5 ** 2))


@pytest.mark.parametrize("h0,cpp,shift", [(0.4, 2, 0.0), (0.4, 2, 0.3), (0.4, 16, 0.1), (1.0, 64, 0.0)])
def test_device_scan27_nn_is_exact_when_it_answers(emu, oracle_mod, fixture_clouds, h0, cpp, shift):
    """grid_nn1_scan27 (the flat 27-cell scan of the second and third correspondence pass): every query it answers has
    the oracle's neighbour and float d2 (ties: lowest index); the lower bound it reports for the OTHER points is valid;
    far / outside queries are declined, never answered wrongly."""
    from scipy.spatial import cKDTree
    v1, _ = oracle_mod.voxel_downsample(fixture_clouds[0], 0.2)
    v2, _ = oracle_mod.voxel_downsample(fixture_clouds[1], 0.2)
    rng = np.random.default_rng(4)
    far = rng.uniform(-150, 150, (400, 3)).astype(np.float32)
    dup = np.repeat(v1[:200], 2, axis=0) + np.float32(0.01)        # pairs of queries with identical neighbours
    q = np.ascontiguousarray(np.concatenate([v2[::3] + np.float32(shift), far, dup, v1[:500]]))   # (v1 itself: d2 = 0)
    idx = np.empty(len(q), np.int32); d2 = np.empty(len(q), np.float32)
    ans = np.empty(len(q), np.int32); lb = np.empty(len(q), np.float32)
    tgt = np.ascontiguousarray(np.concatenate([v1, v1[:50]]))       # duplicate target points: ties of the distance
    oi, od = oracle_mod.nn_search(tgt, q)
    tree = cKDTree(tgt.astype(np.float64))
    # seeds: none / the true neighbour / the 5th neighbour (a stale one) / a random point of the cloud
    _, nb5 = tree.query(q.astype(np.float64), 5)
    seeds = {"none": None, "exact": oi.astype(np.int32), "stale": nb5[:, 4].astype(np.int32),
             "random": rng.integers(0, len(tgt), len(q)).astype(np.int32)}
    seeds["prescan"] = None      # no seed, the query's own row first (grid_nn1_scan27<PRESCAN>, the second pass)
    for kind, sd in seeds.items():
        emu.emu_nn_scan27(tgt.ctypes.data_as(fp), len(tgt), q.ctypes.data_as(fp), len(q), C.c_float(h0), cpp,
                          idx.ctypes.data_as(ip), d2.ctypes.data_as(fp), ans.ctypes.data_as(ip), lb.ctypes.data_as(fp),
                          sd.ctypes.data_as(ip) if sd is not None else None, 1 if kind == "prescan" else 0)
        a = ans == 1
        assert a.mean() > (0.2 if shift > 0.2 or h0 > 0.9 else 0.5), (kind, a.mean())
        assert np.array_equal(idx[a], oi[a]) and np.array_equal(d2[a], od[a]), kind
        # the bound on every other point: the second-nearest distance of a kd-tree query is not below it
        dd, _ = tree.query(q[a].astype(np.float64), 2)
        assert (dd[:, 1] >= lb[a] * (1 - 1e-5) - 1e-6).all(), kind
        assert not a[len(v2[::3]):len(v2[::3]) + 400].all(), kind     # the far queries are not all answerable


@pytest.mark.parametrize("fast", [0, 1])
@pytest.mark.parametrize("hint_kind", ["tiny", "exact", "huge", "random"])
def test_device_box_nn_is_exact_for_any_hint(emu, oracle_mod, fixture_clouds, hint_kind, fast):
    """grid_nn1_box (the kernel's hot variant) must return the exact NN whatever the radius hint - also in the
    first-pass form that keeps the best candidate as one packed key (fast)."""
    v1, _ = oracle_mod.voxel_downsample(fixture_clouds[0], 0.2)
    v2, _ = oracle_mod.voxel_downsample(fixture_clouds[1], 0.2)
    rng = np.random.default_rng(2)
    q = np.ascontiguousarray(np.concatenate([v2[::4], rng.uniform(-150, 150, (300, 3)).astype(np.float32)]))
    oi, od = oracle_mod.nn_search(v1, q)
    hint = {"tiny": np.full(len(q), 1e-4), "exact": np.sqrt(od), "huge": np.full(len(q), 50.0),
            "random": rng.uniform(0.0, 3.0, len(q))}[hint_kind].astype(np.float32)
    idx = np.empty(len(q), np.int32)
    d2 = np.empty(len(q), np.float32)
    pos_ok = C.c_int(0)
    emu.emu_nn_box(v1.ctypes.data_as(fp), len(v1), q.ctypes.data_as(fp), len(q), C.c_float(0.4), 16, C.c_float(2.5),
                   hint.ctypes.data_as(fp), idx.ctypes.data_as(ip), d2.ctypes.data_as(fp), fast, C.byref(pos_ok))
    assert pos_ok.value == len(q)          # the position names the same point as the index
    m = od < 2.5 ** 2
    assert np.array_equal(idx[m], oi[m]) and np.array_equal(d2[m], od[m])
    assert np.all((idx[~m] == -1) | (d2[~m] >= 2.5 ** 2))


@pytest.mark.parametrize("h0,cpp,k", [(0.6, 16, 20), (0.6, 2, 20), (0.3, 64, 7)])
def test_device_knn_normals_match_oracle(emu, oracle_mod, fixture_clouds, h0, cpp, k):
    v1, _ = oracle_mod.voxel_downsample(fixture_clouds[0], 0.3)
    nr = np.empty((len(v1), 3), np.float32)
    emu.emu_normals(v1.ctypes.data_as(fp), len(v1), k, C.c_float(h0), cpp, nr.ctypes.data_as(fp))
    _, on = oracle_mod.gicp_covariances(v1, k)
    dots = np.abs((nr.astype(np.float64) * on).sum(1))
    assert (dots < 1 - 1e-6).mean() < (1e-3 if k == 20 else 0.02)   # identical k-NN sets -> identical normals


@pytest.mark.parametrize("leaf,h0,cpp", [(0.3, 0.6, 2), (0.3, 0.6, 16), (0.2, 0.4, 2), (0.5, 0.3, 64)])
def test_device_knn_med3_answers_exactly_or_declines(emu, oracle_mod, fixture_clouds, leaf, h0, cpp):
    """The round-3 k-NN pre-pass (grid_knn_med3: truncated 32-bit keys, v_med3 insertion, segment table) must return
    the exact 20-neighbour SET of the 64-bit search for every point it answers - on coarse grids (long row ranges,
    many declines), fine grids (the shell, rings beyond it) and with exact duplicates (ties of the distance)."""
    v1, _ = oracle_mod.voxel_downsample(fixture_clouds[0], leaf)
    out = np.zeros(4, np.int64)
    emu.emu_knn3_check(v1.ctypes.data_as(fp), len(v1), C.c_float(h0), cpp, out.ctypes.data_as(C.POINTER(C.c_longlong)))
    assert out[0] == len(v1) and out[2] == 0, out
    assert out[1] < (0.9 if cpp == 64 else 0.6) * len(v1), out   # (a grid much coarser or finer than the 20-neighbour ball declines a lot)
    dup = np.ascontiguousarray(np.concatenate([v1[:3000], v1[:3000], v1[:1500]]))   # every distance a tie
    emu.emu_knn3_check(dup.ctypes.data_as(fp), len(dup), C.c_float(h0), cpp, out.ctypes.data_as(C.POINTER(C.c_longlong)))
    assert out[0] == len(dup) and out[2] == 0, out


def test_quadratic_form_equals_direct_sum(emu):
    rng = np.random.default_rng(0)
    m = 4000
    p = rng.uniform(-60, 60, (m, 3))
    q = p + rng.normal(0, 0.05, (m, 3)) + [0.3, -0.2, 0.05]
    A = rng.normal(size=(m, 3, 3))
    M = A @ A.transpose(0, 2, 1) + 0.1 * np.eye(3)
    M6 = np.ascontiguousarray(np.stack([M[:, 0, 0], M[:, 0, 1], M[:, 0, 2], M[:, 1, 1], M[:, 1, 2], M[:, 2, 2]], 1))
    acc = np.zeros(76)
    # expansion point of the form: a transform near the solution (row-major 3x4)
    x0 = np.array([0.28, -0.18, 0.04, 0.004, -0.012, 0.018])

    def rot(x):
        cph, sph, cth, sth, cps, sps = np.cos(x[3]), np.sin(x[3]), np.cos(x[4]), np.sin(x[4]), np.cos(x[5]), np.sin(x[5])
        return np.array([[cps * cth, cps * sth * sph - sps * cph, cps * sth * cph + sps * sph],
                         [sps * cth, sps * sth * sph + cps * cph, sps * sth * cph - cps * sph],
                         [-sth, cth * sph, cth * cph]])

    def direct(x):
        res = p @ rot(x).T + x[:3] - q
        return (res * np.einsum("nij,nj->ni", M, res)).sum() / m

    th0 = np.ascontiguousarray(np.hstack([rot(x0), x0[:3, None]]))
    emu.emu_gq_build(p.ctypes.data_as(dp), q.ctypes.data_as(dp), M6.ctypes.data_as(dp), m, th0.ctypes.data_as(dp),
                     acc.ctypes.data_as(dp))
    x = np.array([0.25, -0.15, 0.01, 0.005, -0.01, 0.02])
    f = C.c_double()
    g = np.zeros(6)
    emu.emu_gq_eval(acc.ctypes.data_as(dp), th0.ctypes.data_as(dp), x.ctypes.data_as(dp), C.byref(f),
                    g.ctypes.data_as(dp))
    assert abs(f.value - direct(x)) < 1e-12 * abs(f.value)      # expansion about the current transform: no cancellation
    gn = np.zeros(6)
    for i in range(6):
        h = 1e-6
        xp, xm = x.copy(), x.copy()
        xp[i] += h
        xm[i] -= h
        gn[i] = (direct(xp) - direct(xm)) / (2 * h)
    assert np.allclose(g, gn, rtol=1e-6, atol=1e-6)


def test_mahalanobis_from_normals_equals_full_covariances(emu):
    rng = np.random.default_rng(3)
    for _ in range(50):
        n1, n2 = rng.normal(size=3), rng.normal(size=3)
        n1 /= np.linalg.norm(n1)
        n2 /= np.linalg.norm(n2)
        A = rng.normal(size=(3, 3))
        R, _ = np.linalg.qr(A)
        C1 = np.eye(3) - (1 - 1e-3) * np.outer(n1, n1)      # == U diag(1,1,eps) U^T
        C2 = np.eye(3) - (1 - 1e-3) * np.outer(n2, n2)
        ref = np.linalg.inv(R @ C1 @ R.T + C2)
        S = R @ R.T
        S6 = np.array([S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]])
        n1r = R @ n1
        M6 = np.zeros(6)
        emu.emu_mahalanobis(S6.ctypes.data_as(dp), n1r.ctypes.data_as(dp), n2.ctypes.data_as(dp), C.c_double(1e-3),
                            M6.ctypes.data_as(dp))
        got = np.array([[M6[0], M6[1], M6[2]], [M6[1], M6[3], M6[4]], [M6[2], M6[4], M6[5]]])
        assert np.allclose(got, ref, rtol=1e-9, atol=1e-9)


def test_device_pipeline_point_to_plane_equals_oracle(emu, oracle_mod, fixture_clouds):
    p = oracle_mod.default_params(registration_algorithm=oracle_mod.ALG_ICP)
    for a, b in ((0, 1), (2, 3)):
        st, T, info = oracle_mod.align(fixture_clouds[a], fixture_clouds[b], params=p)
        st2, T2, info2 = emu_align(emu, oracle_mod, fixture_clouds[a], fixture_clouds[b], params=p)
        dt, dr = transform_delta(T, T2)
        assert st == st2 == 0 and info["iterations"] == info2["iterations"]
        assert dt < 1e-6 and dr < 1e-6               # tolerance 1e-4 m / 1e-4 rad (north star); achieved: ~0


def test_device_pipeline_gicp_within_north_star_tolerance(emu, oracle_mod, fixture_clouds):
    """quadratic-form GICP + float normals vs the oracle's per-evaluation loops + full covariances,
    both on the smooth objective (eval_precision 2).  Tolerance: 1e-4 m / 1e-4 rad (BASELINE.json)."""
    oracle_mod.set_eval_precision(2)
    try:
        for a, b in ((0, 1), (1, 2), (2, 3)):
            st, T, info = oracle_mod.align(fixture_clouds[a], fixture_clouds[b])
            st2, T2, info2 = emu_align(emu, oracle_mod, fixture_clouds[a], fixture_clouds[b])
            dt, dr = transform_delta(T, T2)
            assert st == st2 == 0
            assert dt < 1e-4 and dr < 1e-4, (a, b, dt, dr)
    finally:
        oracle_mod.set_eval_precision(0)


def test_device_pipeline_gicp_vs_pcl_literal_oracle_same_basin(emu, oracle_mod, fixture_clouds):
    """Against the PCL-literal functor the result is only defined to millimetres (test_conditioning.py);
    the device formulation must land in the same basin and be an equally good minimiser of the
    reference's own objective."""
    for a, b in ((0, 1), (1, 2)):
        st, T, _ = oracle_mod.align(fixture_clouds[a], fixture_clouds[b])
        st2, T2, _ = emu_align(emu, oracle_mod, fixture_clouds[a], fixture_clouds[b])
        dt, dr = transform_delta(T, T2)
        assert st == st2 == 0 and dt < 6e-3 and dr < 1e-3
        c_ref, _ = oracle_mod.gicp_cost(fixture_clouds[a], fixture_clouds[b], T)
        c_dev, _ = oracle_mod.gicp_cost(fixture_clouds[a], fixture_clouds[b], T2)
        assert c_dev < c_ref * 1.01


def test_correspondence_revalidation_is_exact(emu, oracle_mod, fixture_clouds):
    """The triangle-inequality shortcut of the NN kernel (nn_still_nearest) must return exactly what a
    full search returns; the emulation harness checks every shortcut hit against a full search."""
    p = oracle_mod.default_params(registration_algorithm=oracle_mod.ALG_ICP, maximum_iterations=30)
    before = (C.c_longlong * 4)()
    emu.emu_reval_stats(before)
    for a, b in ((0, 1), (2, 3)):
        emu_align(emu, oracle_mod, fixture_clouds[a], fixture_clouds[b], params=p, force=1)
    stats = (C.c_longlong * 4)()
    emu.emu_reval_stats(stats)
    hits, misses, mismatch, far_seeded = (stats[i] - before[i] for i in range(4))
    assert stats[2] == 0                # no shortcut hit / trusted far seed of the whole module run ever disagreed with a full search
    assert far_seeded > 100             # queries without a near neighbour (partial overlap) do take the trusted-seed path
    assert hits > 2 * misses, (hits, misses)   # once ICP has converged nearly every correspondence is re-validated


def test_record_level_revalidation_is_exact(emu, oracle_mod, fixture_clouds):
    """Round 4: the settled passes re-validate 64 queries at once (s3d_core.h nn_record_move_bound / nn_margin,
    s3d_nn_settled_kernel).  The emulation runs the same record logic from the third outer iteration on and checks
    EVERY query of every skipped record against a full search; the registration must come out bit-identical to the
    run without records, and most records must be skipped once the registration has settled."""
    emu.emu_set_records_from.argtypes = [C.c_int]
    before = (C.c_longlong * 4)()
    emu.emu_reval_stats(before)
    for alg, pair in ((oracle_mod.ALG_ICP, (0, 1)), (oracle_mod.ALG_GICP, (2, 3))):
        p = oracle_mod.default_params(registration_algorithm=alg, maximum_iterations=30)
        emu.emu_set_records_from(-1)
        st0, T0, i0 = emu_align(emu, oracle_mod, fixture_clouds[pair[0]], fixture_clouds[pair[1]], params=p, force=1)
        emu.emu_set_records_from(2)
        st1, T1, i1 = emu_align(emu, oracle_mod, fixture_clouds[pair[0]], fixture_clouds[pair[1]], params=p, force=1)
        rs = (C.c_longlong * 3)()
        emu.emu_record_stats(rs)
        emu.emu_set_records_from(-1)
        assert st0 == st1 and np.array_equal(T0, T1) and i0 == i1
        tested, skipped, queries = rs[0], rs[1], rs[2]
        assert skipped > 0.5 * tested, (tested, skipped)     # 28 passes x ~500 records
        assert queries > 100000
    after = (C.c_longlong * 4)()
    emu.emu_reval_stats(after)
    assert after[2] == before[2] == 0       # no skipped query ever disagreed with a full search


def test_record_move_bound_covers_the_float_transforms(emu):
    """nn_record_move_bound must bound |fl(T p) - fl(Tt p)| for every float point of the box, including the rounding
    of the two float transforms: random boxes up to 150 m from the origin, transforms a few micrometres to centimetres
    apart, 4096 points per box (corners included)."""
    emu.emu_record_move_bound.restype = C.c_double
    rng = np.random.default_rng(5)

    def rot(r):
        cx, cy, cz = np.cos(r)
        sx, sy, sz = np.sin(r)
        return (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @
                np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))

    worst_all, tight = 0.0, []
    for trial in range(300):
        scale = 10.0 ** rng.uniform(-7, -2)
        r0, t0 = rng.uniform(-0.05, 0.05, 3), rng.uniform(-1, 1, 3)
        T = np.eye(4); T[:3, :3] = rot(r0); T[:3, 3] = t0
        Tt = np.eye(4); Tt[:3, :3] = rot(r0 + rng.normal(0, scale, 3)); Tt[:3, 3] = t0 + rng.normal(0, 10 * scale, 3)
        Tf = np.ascontiguousarray(T.T.reshape(-1), np.float32)       # column-major floats (Mat4f)
        Ttf = np.ascontiguousarray(Tt.T.reshape(-1), np.float32)
        c = rng.uniform(-150, 150, 3).astype(np.float32)
        c[2] = np.float32(rng.uniform(-5, 5))
        e = (10.0 ** rng.uniform(-2, 1, 3)).astype(np.float32)
        u = rng.uniform(-1, 1, (4096, 3))
        u[:8] = np.array([[i, j, k] for i in (-1, 1) for j in (-1, 1) for k in (-1, 1)], np.float64)
        pts = np.clip((c + u * e).astype(np.float32), c - e, c + e)
        ratio = C.c_double()
        b = emu.emu_record_move_bound(Tf.ctypes.data_as(fp), Ttf.ctypes.data_as(fp), c.ctypes.data_as(fp),
                                      e.ctypes.data_as(fp), np.ascontiguousarray(pts).ctypes.data_as(fp), len(pts), C.byref(ratio))
        assert ratio.value <= 1.0, (trial, b, ratio.value)
        worst_all = max(worst_all, ratio.value)
        tight.append(ratio.value)
    assert worst_all > 0.5     # the bound is not vacuous: some box comes within a factor two of it
    assert np.median(tight) > 0.2


def test_normal_record_round_trip(emu):
    """The 16-byte stored form of a unit normal (s3d_core.h NormalRec: float xyz + three 10-bit remainders in units
    of 2^-34): restored to 2^-34 = 6e-11 per component (half of that except at the +2^-25 tie), the float part is the plain float rounding."""
    rng = np.random.default_rng(0)
    v = rng.normal(size=(20000, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v[:6] = np.array([[1, 0, 0], [0, -1, 0], [0, 0, 1], [-1, 0, 0], [1e-12, 1, 0], [0.6, -0.8, 1e-30]], np.float64)
    worst = 0.0
    for n in v:
        n = np.ascontiguousarray(n, np.float64)
        out = np.zeros(3)
        fpart = np.zeros(3, np.float32)
        emu.emu_normal_roundtrip(n.ctypes.data_as(dp), out.ctypes.data_as(dp), fpart.ctypes.data_as(fp))
        assert np.array_equal(fpart, n.astype(np.float32))
        worst = max(worst, float(np.abs(out - n).max()))
    assert worst <= 2.0 ** -34 + 1e-18, worst


# ------------------------------------------------------------------ round 5: the fused pre-pass (one sort for K2 + K3)

up = C.POINTER(C.c_uint)


def _fused_voxel(emu, cloud, leaf, cpp=2):
    c = np.ascontiguousarray(cloud[:, :3], np.float32)
    xyz = np.empty((len(c) + 1, 3), np.float32)
    vox = np.empty(len(c) + 1, np.uint32)
    cell = np.empty(len(c) + 1, np.int32)
    info = np.zeros(8, np.int64)
    grid = np.zeros(4, np.float32)
    n = emu.emu_fused_voxel(c.ctypes.data_as(fp), len(c), 3, C.c_double(leaf), cpp, xyz.ctypes.data_as(fp),
                            vox.ctypes.data_as(up), cell.ctypes.data_as(ip), info.ctypes.data_as(C.POINTER(C.c_longlong)),
                            grid.ctypes.data_as(fp))
    return xyz[:n], vox[:n], cell[:n], info, grid


@pytest.mark.parametrize("leaf,cpp", [(0.1, 2), (0.2, 2), (0.2, 16), (0.5, 1), (1.0, 2), (0.07, 2)])
def test_fused_prepass_centroids_are_pcl_voxelgrid(emu, oracle_mod, fixture_clouds, leaf, cpp):
    """ONE sort on (cell, voxel) keys yields exactly pcl::VoxelGrid's centroids: sorted by the voxel key they travel with
    they ARE the oracle's output, bit for bit and in its order; in cell order they fill the cell table consistently and
    every one lies in the box of its cell (the searches' assumption)."""
    for cloud in fixture_clouds[:2]:
        ref, _ = oracle_mod.voxel_downsample(cloud, leaf)
        xyz, vox, cell, info, grid = _fused_voxel(emu, cloud, leaf, cpp)
        assert info[0] == 1 and info[7] == 0, info
        assert info[5] == info[2] * info[3] * info[4] and info[5] <= max(cpp * len(cloud), 64) and info[1] >= 2
        order = np.argsort(vox, kind="stable")
        assert len(np.unique(vox)) == len(vox) == len(ref)
        assert np.array_equal(xyz[order], ref)
        # geometry: the centroid lies in its cell (+- 1e-3 cell)
        cz, r = np.divmod(cell, info[2] * info[3]); cy, cx = np.divmod(r, info[2])
        f = (xyz.astype(np.float64) - grid[:3].astype(np.float64)) / float(grid[3]) - np.stack([cx, cy, cz], 1)
        assert f.min() >= -1e-3 and f.max() <= 1 + 1e-3, (f.min(), f.max())


def test_fused_prepass_edge_cases(emu, oracle_mod, fixture_clouds):
    """empty / all-non-finite / one point / a flat cloud / non-finite points mixed in / PCL's index overflow"""
    c = fixture_clouds[0][:, :3]
    xyz, vox, cell, info, _ = _fused_voxel(emu, np.zeros((0, 3), np.float32), 0.2)
    assert len(xyz) == 0 and info[0] == 1 and info[7] == 0
    xyz, vox, cell, info, _ = _fused_voxel(emu, np.full((5, 3), np.nan, np.float32), 0.2)
    assert len(xyz) == 0 and info[0] == 1 and info[7] == 0
    xyz, vox, cell, info, _ = _fused_voxel(emu, c[:1], 0.2)
    assert len(xyz) == 1 and info[0] == 1 and info[7] == 0 and np.array_equal(xyz, c[:1])
    flat = c[:5000].copy(); flat[:, 2] = 1.25
    ref, _ = oracle_mod.voxel_downsample(flat, 0.1)
    xyz, vox, cell, info, _ = _fused_voxel(emu, flat, 0.1)
    assert info[0] == 1 and info[7] == 0 and info[4] == 1 and np.array_equal(xyz[np.argsort(vox)], ref)
    dirty = c[:4000].copy(); dirty[::7, 1] = np.inf; dirty[3::11, 0] = np.nan
    ref, _ = oracle_mod.voxel_downsample(dirty, 0.3)
    xyz, vox, cell, info, _ = _fused_voxel(emu, dirty, 0.3)
    assert info[0] == 1 and info[7] == 0 and np.array_equal(xyz[np.argsort(vox)], ref)
    # leaf far too small for the extent: PCL refuses (index overflow, output = input) - the fused path must decline
    xyz, vox, cell, info, _ = _fused_voxel(emu, c[:2000], 1e-5)
    assert info[0] < 0 and len(xyz) == 0 and info[7] == 0


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("leaf,cpp", [(0.2, 2), (0.1, 2), (0.3, 16)])
def test_fused_prepass_searches_are_exact(emu, oracle_mod, fixture_clouds, leaf, cpp, mode):
    """Every 1-NN search of the registration on the fused grid returns the oracle's neighbour (named by its voxel key =
    its rank in pcl::VoxelGrid's output) and float d2: the box search from any hint, its first-pass form, the flat
    27-cell scan and its pre-scan form (which may decline, never answer wrongly)."""
    src = np.ascontiguousarray(fixture_clouds[0][:, :3])
    v1, _ = oracle_mod.voxel_downsample(src, leaf)
    v2, _ = oracle_mod.voxel_downsample(fixture_clouds[1], leaf)
    _, vox1, _, info, _ = _fused_voxel(emu, src, leaf, cpp)
    keys_sorted = np.sort(vox1)                       # rank in this array = index in the oracle's filtered cloud
    rng = np.random.default_rng(7)
    far = rng.uniform(-150, 150, (300, 3)).astype(np.float32)
    q = np.ascontiguousarray(np.concatenate([v2[::3], far, v1[:300]]))
    oi, od = oracle_mod.nn_search(v1, q)
    hint = rng.uniform(0.0, 3.0, len(q)).astype(np.float32)
    nnv = np.empty(len(q), np.uint32); d2 = np.empty(len(q), np.float32); ans = np.empty(len(q), np.int32)
    emu.emu_fused_nn(src.ctypes.data_as(fp), len(src), 3, C.c_double(leaf), cpp, q.ctypes.data_as(fp), len(q),
                     C.c_float(2.5), hint.ctypes.data_as(fp), mode, nnv.ctypes.data_as(up), d2.ctypes.data_as(fp),
                     ans.ctypes.data_as(ip))
    assert (ans >= 0).all()
    a = ans == 1
    have = a & (nnv != 0xFFFFFFFF)
    idx = np.full(len(q), -1, np.int64)
    idx[have] = np.searchsorted(keys_sorted, nnv[have])
    m = od < 2.5 ** 2
    if mode < 2:
        assert a.all()
        assert np.array_equal(idx[m], oi[m]) and np.array_equal(d2[m], od[m])
        assert np.all((idx[~m] == -1) | (d2[~m] >= 2.5 ** 2))
    else:
        assert a.mean() > 0.3
        assert np.array_equal(idx[a], oi[a]) and np.array_equal(d2[a], od[a])


@pytest.mark.parametrize("leaf,cpp", [(0.3, 2), (0.2, 2), (0.5, 16)])
def test_fused_prepass_knn_is_exact(emu, fixture_clouds, leaf, cpp):
    """k-NN on the fused grid: the exact search that names its neighbours by POSITION equals a brute-force scan, and the
    med3 pre-pass returns that set for every point it answers."""
    src = np.ascontiguousarray(fixture_clouds[0][:, :3])
    out = np.zeros(4, np.int64)
    emu.emu_fused_knn_check(src.ctypes.data_as(fp), len(src), 3, C.c_double(leaf), cpp, out.ctypes.data_as(C.POINTER(C.c_longlong)))
    assert out[0] > 1000 and out[2] == 0 and out[3] == 0, out
    assert out[1] < 0.7 * out[0], out


@pytest.mark.parametrize("leaf,cpp,rmax,only", [(0.2, 2, 6, 1), (0.2, 2, 6, 0), (0.3, 2, 4, 1), (0.5, 16, 6, 1), (0.2, 1, 3, 1),
                                                (0.2, 2, 606, 1)])     # (+ 600: a 64-entry table instead of the device's 32)
def test_ring_by_ring_knn_is_exact(emu, fixture_clouds, leaf, cpp, rmax, only):
    """Round 6: the sparse parts of a scan ring by ring (grid_knn_med3_rings: whole rings while the list is short, then
    one pruned box) - every point it answers has the exact 20-NN set of the search by position, on the points the fast
    path declines (what the device hands to it) and on every point; and it answers most of those declines (the
    reference's scans: 93 % of them have fewer than 20 points in their 27 cells, tools_dev/knn3_declines.cpp)."""
    for cloud in (fixture_clouds[0], fixture_clouds[3]):
        src = np.ascontiguousarray(cloud[:, :3])
        out = np.zeros(4, np.int64)
        emu.emu_fused_knn_rings_check(src.ctypes.data_as(fp), len(src), 3, C.c_double(leaf), cpp, rmax, only,
                                      out.ctypes.data_as(C.POINTER(C.c_longlong)))
        assert out[0] > 200 and out[2] == 0, out
        if only and rmax % 100 >= 6:
            assert out[1] < 0.25 * out[0], out          # most of the fast path's declines are answered here
        print("leaf %g cpp %d rmax %d: tried %d, not answered %d, fast-path declines %d" % (leaf, cpp, rmax, out[0], out[1], out[3]))
