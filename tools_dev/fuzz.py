"""dev tool: randomized GPU-vs-oracle checks of the bit-exact stages (voxel grid, 1-NN incl. far / outside
queries, radius outlier removal, patch accumulation) and of point-to-plane align() on odd inputs."""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
import slam3d_amd as s3d, oracle
from tests.conftest import transform_delta
ctx = s3d.Context(0)
rng = np.random.default_rng(int(os.environ.get('SEED', '1')))
N = int(os.environ.get('CASES', '40'))
bad = 0
def cloud(kind, n):
    if kind == 0: return rng.uniform(-20, 20, (n, 3)).astype(np.float32)                       # volume
    if kind == 1:                                                                                 # planes + outliers
        p = rng.uniform(-30, 30, (n, 3)).astype(np.float32); p[: n // 2, 2] = rng.normal(0, 0.02, n // 2)
        p[n // 2: 3 * n // 4, 0] = 5 + rng.normal(0, 0.02, 3 * n // 4 - n // 2); p[-5:] *= 40; return p
    if kind == 2: return (rng.normal(0, 1, (n, 3)) * [30, 0.05, 0.05]).astype(np.float32)        # a line
    if kind == 3: return np.repeat(rng.uniform(-5, 5, (max(n // 8, 1), 3)).astype(np.float32), 8, 0)   # duplicates
    return (s3d.make_scene_cloud(n, int(rng.integers(1 << 30)))).astype(np.float32)               # the bench scene
t0 = time.time()
for case in range(N):
    kind = int(rng.integers(5)); n = int(rng.integers(200, 60000))
    c = cloud(kind, n)
    leaf = float(rng.choice([0.05, 0.2, 0.5, 1.0, 3.0]))
    v_o = oracle.voxel_downsample(c, leaf)[0]; v_g = ctx.voxel_downsample(c, leaf)
    if v_o.shape != v_g.shape or not np.array_equal(v_o, v_g): bad += 1; print('VOXEL mismatch', case, kind, n, leaf)
    q = (c[rng.integers(0, len(c), 3000)] + rng.normal(0, rng.choice([0.01, 0.3, 3.0]), (3000, 3))).astype(np.float32)
    q[:20] *= 50                                                                                   # far outside the grid
    md = float(rng.choice([0.5, 2.5, 10.0]))
    io, do = oracle.nn_search(v_o, q); ig, dg = ctx.nn_search(v_o, q, md)
    m = do < md * md
    if not (np.array_equal(ig[m], io[m]) and np.array_equal(dg[m], do[m])): bad += 1; print('NN mismatch', case, kind, n, leaf, md, int((ig[m] != io[m]).sum()))
    r = float(rng.choice([0.1, 0.3, 1.0])); k = int(rng.choice([1, 3, 10]))
    ro = oracle.remove_outliers(v_o, r, k); rg = ctx.remove_outliers(v_o, r, k)
    if ro.shape != rg.shape or not np.array_equal(ro, rg): bad += 1; print('OUTLIER mismatch', case, kind, n, r, k)
    if len(v_o) >= 200 and kind in (1, 4):
        T = np.eye(4); T[:3, 3] = rng.uniform(-0.3, 0.3, 3)
        b = (v_o @ np.eye(3) + T[:3, 3]).astype(np.float32)
        po = oracle.default_params(registration_algorithm=oracle.ALG_ICP, point_cloud_density=0.0, maximum_iterations=15)
        pg = s3d.default_params(registration_algorithm=s3d.ALG_ICP, point_cloud_density=0.0, maximum_iterations=15)
        so, To, _ = oracle.align(b, v_o, np.eye(4), po); sg, Tg, _ = ctx.align(b, v_o, np.eye(4), pg)
        dt, dr = transform_delta(To, Tg)
        if so != sg or (so == 0 and (dt > 1e-4 or dr > 1e-4)): bad += 1; print('ALIGN mismatch', case, kind, n, so, sg, dt, dr)
print('cases', N, 'mismatches', bad, '%.1f s' % (time.time() - t0))
