"""CPU suite: the oracle (oracle/s3d_oracle.c) against (a) independent numpy/scipy restatements of
each stage and (b) the committed golden vectors.  The reference holds no expected values for this
path and cannot be built here — parity vs PCL itself is UNPINNED."""
import hashlib
import json
import os

import numpy as np
import pytest
from scipy.spatial import cKDTree

from conftest import GOLDEN, transform_delta


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLDEN, "oracle_golden.json")) as f:
        return json.load(f)


def numpy_voxel(xyzi, leaf):
    """Independent restatement of pcl::VoxelGrid for XYZ."""
    p = xyzi[:, :3].astype(np.float32)
    leaf = np.float32(leaf)
    inv = np.float32(1.0) / leaf
    mn, mx = p.min(0), p.max(0)
    min_b = np.floor(mn * inv).astype(np.int64)
    max_b = np.floor(mx * inv).astype(np.int64)
    div = max_b - min_b + 1
    ijk = (np.floor(p * inv) - min_b.astype(np.float32)).astype(np.int64)
    key = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    starts = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
    cnt = np.diff(np.r_[starts, len(ks)])
    sums = np.add.reduceat(p[order].astype(np.float64), starts, axis=0)
    return (sums / cnt[:, None]), div


def test_voxel_counts_match_survey(oracle_mod, fixture_clouds):
    # SURVEY.md §8d "fixture characterisation": 60152 / 31834 / 10970 / 4273 voxels
    for leaf, n in ((0.1, 60152), (0.2, 31834), (0.5, 10970), (1.0, 4273)):
        v, info = oracle_mod.voxel_downsample(fixture_clouds[0], leaf)
        assert len(v) == n
    v, info = oracle_mod.voxel_downsample(fixture_clouds[0], 0.2)
    assert list(info.div_b) == [781, 504, 73]
    assert len(oracle_mod.voxel_downsample(fixture_clouds[1], 0.2)[0]) == 31481


@pytest.mark.parametrize("leaf", [0.1, 0.2, 0.5])
def test_voxel_vs_numpy(oracle_mod, fixture_clouds, leaf):
    v, info = oracle_mod.voxel_downsample(fixture_clouds[0], leaf)
    ref, div = numpy_voxel(fixture_clouds[0], leaf)
    assert list(info.div_b) == list(div)
    assert v.shape == ref.shape
    # float32 running sums vs float64 sums: a few ulp of the coordinates
    assert np.abs(v - ref).max() < 2e-4


def test_voxel_golden(oracle_mod, fixture_clouds, golden):
    for leaf, g in golden["voxel"].items():
        v, info = oracle_mod.voxel_downsample(fixture_clouds[0], float(leaf))
        assert len(v) == g["n"] and list(info.div_b) == g["div_b"] and list(info.min_b) == g["min_b"]
        assert hashlib.sha256(v.tobytes()).hexdigest() == g["sha256"]
        assert np.array_equal(v[:8], np.array(g["first8"], np.float32))


def test_voxel_edge_cases(oracle_mod):
    v, _ = oracle_mod.voxel_downsample(np.zeros((0, 3), np.float32), 0.2)
    assert len(v) == 0                                   # PointCloudSensor.cpp:193
    one = np.array([[1.0, 2.0, 3.0]], np.float32)
    v, _ = oracle_mod.voxel_downsample(one, 0.2)
    assert np.array_equal(v, one)
    # leaf too small: dx*dy*dz > INT_MAX -> PCL returns the input cloud
    far = np.array([[0, 0, 0], [1000, 1000, 1000], [5, 5, 5]], np.float32)
    v, info = oracle_mod.voxel_downsample(far, 0.0005)
    assert info.passthrough == 1 and np.array_equal(v, far)
    # non-finite points are dropped
    bad = np.array([[0, 0, 0], [np.nan, 0, 0], [0.01, 0, 0]], np.float32)
    v, _ = oracle_mod.voxel_downsample(bad, 1.0)
    assert len(v) == 1 and np.allclose(v[0], [0.005, 0, 0])


def test_nn_vs_scipy_and_brute(oracle_mod, fixture_clouds, golden):
    v1, _ = oracle_mod.voxel_downsample(fixture_clouds[0], 0.2)
    v2, _ = oracle_mod.voxel_downsample(fixture_clouds[1], 0.2)
    idx, d2 = oracle_mod.nn_search(v1, v2)
    dd, ii = cKDTree(v1.astype(np.float64)).query(v2.astype(np.float64))
    assert (ii != idx).mean() < 1e-4            # only exact-tie / rounding cases may differ
    assert np.abs(dd ** 2 - d2).max() < 1e-4
    bi, bd = oracle_mod.nn_search(v1[:3000], v2[:500], brute=True)
    ki, kd = oracle_mod.nn_search(v1[:3000], v2[:500])
    assert np.array_equal(bi, ki) and np.array_equal(bd, kd)
    g = golden["nn"]
    assert idx[g["queries"]].tolist() == g["idx"]
    assert hashlib.sha256(idx.tobytes()).hexdigest() == g["sha256_idx"]


def test_covariances_vs_numpy(oracle_mod, fixture_clouds):
    v1, _ = oracle_mod.voxel_downsample(fixture_clouds[0], 0.5)
    cov, nrm = oracle_mod.gicp_covariances(v1, 20)
    tree = cKDTree(v1.astype(np.float64))
    _, nb = tree.query(v1.astype(np.float64), k=20)
    rng = np.random.default_rng(0)
    for i in rng.choice(len(v1), 200, replace=False):
        # PCL computeCovariances: float32 products, float64 sums, E[xx^T] - mean mean^T
        Pf = v1[nb[i]]
        prod = (Pf[:, :, None] * Pf[:, None, :]).astype(np.float64).sum(0) / 20.0
        mean = Pf.astype(np.float64).sum(0) / 20.0
        c = prod - np.outer(mean, mean)
        w, U = np.linalg.eigh(c)                       # ascending
        if w[1] - w[0] < 1e-6 * max(w[2], 1e-12):
            continue                                   # (near-)degenerate smallest direction
        ref = U @ np.diag([1e-3, 1.0, 1.0]) @ U.T      # PCL: (1, 1, gicp_epsilon) on descending order
        assert np.abs(cov[i] - ref).max() < 1e-5
        assert abs(abs(nrm[i] @ U[:, 0]) - 1) < 1e-6


def test_gicp_oracle_golden(oracle_mod, fixture_clouds, golden):
    """Replays every golden align() case bit-for-bit (same binary arithmetic on any x86-64 host)."""
    for case in golden["align"]:
        alg = oracle_mod.ALG_GICP if case["algorithm"] == "gicp" else oracle_mod.ALG_ICP
        oracle_mod.set_eval_precision(case["eval_precision"])
        try:
            st, T, info = oracle_mod.align(fixture_clouds[case["source"] - 1], fixture_clouds[case["target"] - 1],
                                           np.array(case["guess"]), oracle_mod.default_params(registration_algorithm=alg))
        finally:
            oracle_mod.set_eval_precision(0)
        assert st == case["status"], case
        assert info["iterations"] == case["info"]["iterations"]
        dt, dr = transform_delta(np.array(case["T"]), T)
        assert dt < 1e-9 and dr < 1e-7


def test_fixture_motion_matches_survey(golden):
    # consecutive sweeps move ~0.67 / 0.68 / 0.72 m; cloud4->cloud1 ~2.1 m and is
    # rejected by the default max_translation = 1.0 gate when the guess is identity
    by = {(c["algorithm"], c["eval_precision"], c["source"], c["target"], c["guess"][0][3]): c for c in golden["align"]}
    assert abs(by[("gicp", 0, 1, 2, 0.0)]["T"][0][3] - 0.68) < 0.02
    assert abs(by[("gicp", 0, 2, 3, 0.0)]["T"][0][3] - 0.69) < 0.02
    assert abs(by[("gicp", 0, 3, 4, 0.0)]["T"][0][3] - 0.72) < 0.02
    assert by[("gicp", 0, 1, 4, 0.0)]["status"] == oracle_status("TOO_FAR_FROM_GUESS")
    assert by[("gicp", 0, 1, 4, 2.0)]["status"] == 0 and abs(by[("gicp", 0, 1, 4, 2.0)]["T"][0][3] - 2.1) < 0.03


def oracle_status(name):
    import oracle
    return getattr(oracle, "STATUS_" + name)


def test_align_gates(oracle_mod, fixture_clouds):
    o = oracle_mod
    few = fixture_clouds[0][:20]                                    # test.ply has 20 vertices
    st, _, _ = o.align(few, fixture_clouds[1])
    assert st == o.STATUS_TOO_FEW_POINTS                            # PointCloudSensor.cpp:134-135
    st, _, _ = o.align(fixture_clouds[0], fixture_clouds[1], params=o.default_params(registration_algorithm=9))
    assert st == o.STATUS_UNKNOWN_ALGORITHM                         # :163-164
    st, _, info = o.align(fixture_clouds[0], fixture_clouds[1], params=o.default_params(max_fitness_score=0.01))
    assert st == o.STATUS_FITNESS_EXCEEDED and info["fitness"] > 0.01   # :74
    st, _, _ = o.align(fixture_clouds[0], fixture_clouds[1], params=o.default_params(max_translation=0.1))
    assert st == o.STATUS_TOO_FAR_FROM_GUESS                        # :169-172


def test_create_constraint_frame_algebra(oracle_mod, fixture_clouds):
    o = oracle_mod
    ident = np.eye(4)
    # point-to-plane mode: well-conditioned, so the 1e-16 rounding of the composed guess cannot
    # change the result visibly (GICP can: tests/test_conditioning.py)
    fine = o.default_params(registration_algorithm=o.ALG_ICP)
    st, icp, _ = o.align(fixture_clouds[0], fixture_clouds[1], params=fine)
    Ps = np.eye(4); Ps[:3, 3] = [0.5, 0.1, 1.2]
    c, s = np.cos(0.3), np.sin(0.3)
    Pt = np.eye(4); Pt[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]; Pt[:3, 3] = [-0.2, 0.3, 1.0]
    odo = Ps @ np.linalg.inv(Pt)          # makes guess = Ps^-1 * odo * Pt = I  (PointCloudSensor.cpp:274)
    st2, rel, inf, _ = o.create_constraint(fixture_clouds[0], Ps, fixture_clouds[1], Pt, odo, fine=fine,
                                           covariance_scale=4.0)
    assert st == 0 and st2 == 0
    dt, dr = transform_delta(Ps @ icp @ np.linalg.inv(Pt), rel)      # :295
    assert dt < 1e-5 and dr < 1e-5
    assert np.allclose(inf, np.eye(6) / 4.0)                         # :296-298
    # loop closure: coarse + fine (:286-292)
    coarse = o.default_params(point_cloud_density=0.5, max_correspondence_distance=5.0)
    st3, rel3, _, _ = o.create_constraint(fixture_clouds[0], ident, fixture_clouds[1], ident, ident, loop=True,
                                          coarse=coarse)
    assert st3 == 0 and abs(rel3[0, 3] - 0.68) < 0.02


def test_synthetic_known_answer(oracle_mod):
    import slam3d_amd.synthetic as syn
    src, tgt, T_true = syn.make_pair(20000, 5)
    for alg in (oracle_mod.ALG_GICP, oracle_mod.ALG_ICP):
        p = oracle_mod.default_params(registration_algorithm=alg, point_cloud_density=0.02)
        st, T, info = oracle_mod.align(src, tgt, np.eye(4), p)
        dt, dr = transform_delta(T_true, T)
        assert st == 0 and dt < 5e-3 and dr < 1e-3      # independent resample + 1 cm noise: mm-level truth


def test_ndt_oracle_converges_next_to_gicp(oracle_mod, fixture_clouds):
    """doNDT restatement (PointCloudSensor.cpp:84-117, pcl::NormalDistributionsTransform): on the reference's
    fixture scans it must land within millimetres of the (independent) GICP result, reduce the NDT objective, and
    stop by the translation-epsilon rule well before the iteration cap."""
    o = oracle_mod
    o.set_eval_precision(2)
    for a, b in ((0, 1), (2, 3)):
        st_g, T_g, _ = o.align(fixture_clouds[a], fixture_clouds[b], np.eye(4), o.default_params())
        st_n, T_n, info = o.align(fixture_clouds[a], fixture_clouds[b], np.eye(4),
                                  o.default_params(registration_algorithm=o.ALG_NDT))
        assert st_g == st_n == 0
        dt, dr = transform_delta(T_g, T_n)
        assert dt < 5e-3 and dr < 2e-3
        assert info["converged"] == 1 and 3 <= info["iterations"] < 35 and info["correspondences"] > 500
    o.set_eval_precision(0)
    # a coarser NDT grid still converges; a wrong guess far outside max_translation is rejected by the align() gate
    st, T, info = o.align(fixture_clouds[0], fixture_clouds[1], np.eye(4),
                          o.default_params(registration_algorithm=o.ALG_NDT, resolution=2.0))
    assert st == 0 and abs(T[0, 3] - 0.68) < 0.05


def test_ndt_golden_replay(oracle_mod, fixture_clouds):
    """tests/golden/ndt_golden.json (tests/golden/make_ndt_golden.py) replayed bit for bit."""
    for case in json.load(open(os.path.join(GOLDEN, "ndt_golden.json"))):
        g = np.eye(4)
        g[0, 3] = case["guess_x"]
        p = oracle_mod.default_params(registration_algorithm=getattr(oracle_mod, "ALG_" + case.get("algorithm", "NDT")), **case["params"])
        st, T, info = oracle_mod.align(fixture_clouds[case["source"] - 1], fixture_clouds[case["target"] - 1], g, p)
        assert st == case["status"] and np.array_equal(T, np.array(case["T"]))
        assert info["iterations"] == case["info"]["iterations"] and info["fitness"] == case["info"]["fitness"]


def test_gicp_objective_vs_independent_numpy_and_scipy(oracle_mod, fixture_clouds):
    """The oracle's GICP result against an independent statement of the same problem: the objective
    mean(d^T (R C_T R^T + C_S)^-1 d) over the correspondences within max_correspondence_distance, evaluated with
    numpy / cKDTree at the oracle's result, equals s3o_gicp_cost; and scipy's BFGS on that objective (correspondences
    and Mahalanobis matrices frozen, as in PCL's inner problem) finds nothing better than a 1e-3 relative
    improvement a few millimetres away - the oracle stops where PCL's stopping rule says, next to the minimiser."""
    from scipy.optimize import minimize
    from scipy.spatial import cKDTree
    c1, c2 = fixture_clouds[0], fixture_clouds[1]
    p = oracle_mod.default_params(point_cloud_density=0.5)
    st, T, info = oracle_mod.align(c1, c2, np.eye(4), p)
    assert st == 0
    cost, cnt = oracle_mod.gicp_cost(c1, c2, T, p)
    S = oracle_mod.voxel_downsample(c1, 0.5)[0]
    Q = oracle_mod.voxel_downsample(c2, 0.5)[0]
    CS, _ = oracle_mod.gicp_covariances(S, 20, 1e-3)
    CQ, _ = oracle_mod.gicp_covariances(Q, 20, 1e-3)
    F = T.astype(np.float32)                                   # getFinalTransformation() is a Matrix4f
    q = (Q.astype(np.float32) @ F[:3, :3].T + F[:3, 3]).astype(np.float32)
    d, j = cKDTree(S.astype(np.float64)).query(q.astype(np.float64))
    m = d ** 2 < p.max_correspondence_distance ** 2
    assert int(m.sum()) == cnt
    R = F[:3, :3].astype(np.float64)
    M = np.linalg.inv(np.einsum("ab,nbc,dc->nad", R, CQ[m], R) + CS[j[m]])
    r = q[m].astype(np.float64) - S[j[m]].astype(np.float64)
    assert abs(np.einsum("na,nab,nb->n", r, M, r).mean() - cost) < 1e-6 * cost
    P, Sj = Q[m].astype(np.float64), S[j[m]].astype(np.float64)

    def rot(rx, ry, rz):
        cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
        return np.array([[cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx],
                         [sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx], [-sy, cy * sx, cy * cx]])

    def f(x):
        D = np.eye(4)
        D[:3, :3], D[:3, 3] = rot(*x[3:]), x[:3]
        A = D @ T
        rr = P @ A[:3, :3].T + A[:3, 3] - Sj
        return np.einsum("na,nab,nb->n", rr, M, rr).mean()

    res = minimize(f, np.zeros(6), method="BFGS", options={"gtol": 1e-10})
    assert (f(np.zeros(6)) - res.fun) < 1e-3 * res.fun
    assert np.abs(res.x[:3]).max() < 5e-3 and np.abs(res.x[3:]).max() < 1e-4


def test_point_to_plane_result_is_stationary_for_independent_normal_equations(oracle_mod, fixture_clouds):
    """The point-to-plane oracle's result against an independent statement: with correspondences from cKDTree and the
    target normals, the numpy least-squares step of the linearised point-to-plane residuals at the returned
    transform is (numerically) zero - the iteration has converged to the minimiser of its own objective."""
    from scipy.spatial import cKDTree
    c1, c2 = fixture_clouds[0], fixture_clouds[1]
    p = oracle_mod.default_params(point_cloud_density=0.5, registration_algorithm=oracle_mod.ALG_ICP)
    st, T, info = oracle_mod.align(c1, c2, np.eye(4), p)
    assert st == 0
    S = oracle_mod.voxel_downsample(c1, 0.5)[0]
    Q = oracle_mod.voxel_downsample(c2, 0.5)[0]
    _, NS = oracle_mod.gicp_covariances(S, 20, 1e-3)
    F = T.astype(np.float32)
    q = (Q.astype(np.float32) @ F[:3, :3].T + F[:3, 3]).astype(np.float32).astype(np.float64)
    d, j = cKDTree(S.astype(np.float64)).query(q)
    m = d ** 2 < p.max_correspondence_distance ** 2
    assert int(m.sum()) == info["correspondences"]
    n, s, qq = NS[j[m]], S[j[m]].astype(np.float64), q[m]
    r = np.einsum("na,na->n", qq - s, n)
    J = np.hstack([n, np.cross(qq, n)])          # d r / d (translation, small rotation)
    x = np.linalg.lstsq(J, -r, rcond=None)[0]
    assert np.abs(x[:3]).max() < 1e-5 and np.abs(x[3:]).max() < 1e-6


def test_ndt_result_maximises_independent_ndt_score(oracle_mod, fixture_clouds):
    """The NDT oracle against an independent numpy statement of pcl::NormalDistributionsTransform's score: voxel
    Gaussians of the target (>= 6 points per voxel of edge `resolution`, unbiased covariance, eigenvalues raised to
    1 % of the largest), score = sum over points and over the cells whose mean lies within `resolution` of
    -d1 exp(-d2/2 (x-mu)^T Sigma^-1 (x-mu)).  Same number of cells; the oracle's result scores far above the
    initial guess and every +-1 cm / +-5 mrad step away from it lowers the score."""
    from scipy.spatial import cKDTree
    c1, c2 = fixture_clouds[0], fixture_clouds[1]
    p = oracle_mod.default_params(registration_algorithm=oracle_mod.ALG_NDT)
    st, T, info = oracle_mod.align(c1, c2, np.eye(4), p)
    assert st == 0
    res, orat = p.resolution, p.outlier_ratio
    S = oracle_mod.voxel_downsample(c1, p.point_cloud_density)[0].astype(np.float64)
    Q = oracle_mod.voxel_downsample(c2, p.point_cloud_density)[0].astype(np.float64)
    _, inv, cnt = np.unique(np.floor(S / res).astype(np.int64), axis=0, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    mus, icovs = [], []
    for c in np.nonzero(cnt >= 6)[0]:
        P = S[inv == c]
        Cv = np.cov(P.T)
        w, V = np.linalg.eigh(Cv)
        if w[0] < 0.01 * w[2]:
            Cv = V @ np.diag(np.maximum(w, 0.01 * w[2])) @ V.T
        mus.append(P.mean(0))
        icovs.append(np.linalg.inv(Cv))
    mus, icovs = np.array(mus), np.array(icovs)
    assert len(mus) == info["correspondences"]            # the oracle reports its cell count there
    g1, g2 = 10 * (1 - orat), orat / res ** 3
    d3 = -np.log(g2)
    d1 = -np.log(g1 + g2) - d3
    d2 = -2 * np.log((-np.log(g1 * np.exp(-0.5) + g2) - d3) / d1)
    tree = cKDTree(mus)

    def score(A):
        X = Q @ A[:3, :3].T + A[:3, 3]
        nb = tree.query_ball_point(X, res)
        ii = np.repeat(np.arange(len(X)), [len(b) for b in nb])
        jj = np.concatenate([np.array(b, dtype=int) for b in nb])
        dd = X[ii] - mus[jj]
        return float((-d1 * np.exp(-d2 / 2 * np.einsum("na,nab,nb->n", dd, icovs[jj], dd))).sum())

    s0 = score(T)
    assert s0 > 1.4 * score(np.eye(4))
    for k in range(6):
        for sign in (1.0, -1.0):
            D = np.eye(4)
            if k < 3:
                D[k, 3] = sign * 1e-2
            else:
                a, b = [(1, 2), (0, 2), (0, 1)][k - 3]
                c, s = np.cos(5e-3), np.sin(sign * 5e-3)
                D[a, a], D[b, b], D[a, b], D[b, a] = c, c, -s, s
            assert score(D @ T) < s0, (k, sign)


def test_ndt_omp_result_maximises_independent_direct7_score(oracle_mod, fixture_clouds):
    """NDT_OMP = pclomp's NDT with its default DIRECT7 neighbour search (PointCloudSensor.cpp:155-157).  The oracle's
    restatement against an independent numpy statement of that objective: the same voxel Gaussians as PCL's NDT, but a
    point is scored against the voxel that holds it and its six face neighbours (those that are cells) and nothing
    else - no radius.  The oracle's NDT_OMP result scores far above the guess, every +-1 cm / +-5 mrad step away from
    it lowers the score, and it is not the plain NDT's result."""
    c1, c2 = fixture_clouds[0], fixture_clouds[1]
    p = oracle_mod.default_params(registration_algorithm=oracle_mod.ALG_NDT_OMP)
    st, T, info = oracle_mod.align(c1, c2, np.eye(4), p)
    st_p, T_p, info_p = oracle_mod.align(c1, c2, np.eye(4), oracle_mod.default_params(registration_algorithm=oracle_mod.ALG_NDT))
    assert st == 0 and st_p == 0 and not np.array_equal(T, T_p) and info["correspondences"] == info_p["correspondences"]
    assert np.linalg.norm(T[:3, 3] - T_p[:3, 3]) < 0.01          # two objectives, the same scene
    res, orat = p.resolution, p.outlier_ratio
    S = oracle_mod.voxel_downsample(c1, p.point_cloud_density)[0].astype(np.float64)
    Q = oracle_mod.voxel_downsample(c2, p.point_cloud_density)[0].astype(np.float64)
    vox = np.floor(S / res).astype(np.int64)
    keys, inv, cnt = np.unique(vox, axis=0, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    cell_of, mus, icovs = {}, [], []
    for c in np.nonzero(cnt >= 6)[0]:
        P = S[inv == c]
        Cv = np.cov(P.T)
        w, V = np.linalg.eigh(Cv)
        if w[0] < 0.01 * w[2]:
            Cv = V @ np.diag(np.maximum(w, 0.01 * w[2])) @ V.T
        cell_of[tuple(keys[c])] = len(mus)
        mus.append(P.mean(0))
        icovs.append(np.linalg.inv(Cv))
    mus, icovs = np.array(mus), np.array(icovs)
    assert len(mus) == info["correspondences"]
    g1, g2 = 10 * (1 - orat), orat / res ** 3
    d3 = -np.log(g2)
    d1 = -np.log(g1 + g2) - d3
    d2 = -2 * np.log((-np.log(g1 * np.exp(-0.5) + g2) - d3) / d1)
    rel = [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]

    def score(A):
        X = Q @ A[:3, :3].T + A[:3, 3]
        V = np.floor(X / res).astype(np.int64)
        ii, jj = [], []
        for i, v in enumerate(map(tuple, V)):
            for r in rel:
                c = cell_of.get((v[0] + r[0], v[1] + r[1], v[2] + r[2]))
                if c is not None:
                    ii.append(i); jj.append(c)
        dd = X[ii] - mus[jj]
        return float((-d1 * np.exp(-d2 / 2 * np.einsum("na,nab,nb->n", dd, icovs[jj], dd))).sum())

    s0 = score(T)
    assert s0 > 1.4 * score(np.eye(4))
    for k in range(6):
        for sign in (1.0, -1.0):
            D = np.eye(4)
            if k < 3:
                D[k, 3] = sign * 1e-2
            else:
                a, b = [(1, 2), (0, 2), (0, 1)][k - 3]
                c, s = np.cos(5e-3), np.sin(sign * 5e-3)
                D[a, a], D[b, b], D[a, b], D[b, a] = c, c, -s, s
            assert score(D @ T) < s0, (k, sign)
