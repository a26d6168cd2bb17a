"""dev tool (round 5): the k-NN pre-pass's fast paths on odd clouds - registrations with the default path, with every far
decline through the wave-cooperative kernel, through the per-lane exact search, and with the exact 64-bit search for every
point must agree bit for bit (status, transform, iteration counts), on both pre-pass layouts.  env: SEED0 (0), CASES (40)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
A = s3d.api
rng = np.random.default_rng(int(os.environ.get('SEED0', '0')))
ctx = s3d.Context(0)
def cloud(kind, n):
    if kind == 0: return rng.uniform(-15, 15, (n, 3)).astype(np.float32)
    if kind == 1:                                   # a plane with a sparse far field
        p = rng.uniform(-40, 40, (n, 3)).astype(np.float32); p[: 2 * n // 3, 2] = rng.normal(0, 0.03, 2 * n // 3); return p
    if kind == 2:                                   # lidar-like rings: dense near, sparse far
        r = np.abs(rng.normal(0, 12, n)) + 0.5; a = rng.uniform(0, 2 * np.pi, n)
        return np.stack([r * np.cos(a), r * np.sin(a), rng.normal(0, 0.05, n) + 0.02 * r], 1).astype(np.float32)
    if kind == 3:                                   # clusters far apart
        c = rng.uniform(-60, 60, (6, 3)); return (c[rng.integers(0, 6, n)] + rng.normal(0, 0.4, (n, 3))).astype(np.float32)
    return s3d.make_scene_cloud(n, int(rng.integers(1 << 30))).astype(np.float32)
bad = 0; ncase = int(os.environ.get('CASES', '40'))
for case in range(ncase):
    kind = int(rng.integers(5)); n = int(rng.integers(800, 60000))
    a = cloud(kind, n)
    T = np.eye(4); T[:3, 3] = rng.uniform(-0.2, 0.2, 3)
    b = (a + rng.normal(0, 0.004, a.shape).astype(np.float32) - T[:3, 3].astype(np.float32)).astype(np.float32)
    dens = float(rng.choice([0.05, 0.2, 0.5])); alg = s3d.ALG_GICP if rng.random() < 0.75 else s3d.ALG_ICP
    k = int(rng.choice([20, 20, 12, 30]))
    p = s3d.default_params(registration_algorithm=alg, point_cloud_density=dens, maximum_iterations=6, correspondence_randomness=k)
    da, db = ctx.upload(a), ctx.upload(b)
    res = []
    for fl in (0, A.DBG_KNN_FORCE_FAR_COOP, A.DBG_KNN_NO_FAR_COOP, A.DBG_KNN_NO_FAR_COOP | A.DBG_KNN_FORCE_RINGS, A.DBG_KNN_EXACT64, A.DBG_NO_FUSED_PREPASS,
               A.DBG_NO_FUSED_PREPASS | A.DBG_KNN_EXACT64):
        res.append((fl, ctx.align_batch([da], [db], None, p, s3d.ExecOptions(debug_flags=fl))))
    da.release(); db.release()
    same_fused = all(np.array_equal(res[0][1], r[1]) for r in res[1:5])
    same_two = np.array_equal(res[5][1], res[6][1])
    # the two layouts order points differently inside a cell: equal statuses, transforms to 2e-6
    close = res[0][1][0, 15] == res[5][1][0, 15] and np.abs(res[0][1][0, :12] - res[5][1][0, :12]).max() < 2e-6
    if not (same_fused and same_two and close):
        bad += 1; print('DIFF case %d kind %d n %d density %g k %d alg %d: fused paths equal %s, two-sort paths equal %s, layouts close %s' %
                        (case, kind, n, dens, k, alg, same_fused, same_two, close), flush=True)
print('%d cases, %d differ' % (ncase, bad))
