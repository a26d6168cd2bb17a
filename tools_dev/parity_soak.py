"""dev tool: GICP / point-to-plane registrations of random synthetic pairs, GPU against the oracle's smooth-objective
variant (eval_precision 2: what the tests compare with), tolerance of the tests."""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
import slam3d_amd as s3d, oracle
from tests.conftest import transform_delta
ctx = s3d.Context(0)
oracle.set_eval_precision(2)
rng = np.random.default_rng(int(os.environ.get('SEED', '5')))
N = int(os.environ.get('CASES', '24'))
worst = [0.0, 0.0]; bad = 0
for case in range(N):
    n = int(rng.choice([3000, 20000, 60000]))
    a = s3d.make_scene_cloud(n, int(rng.integers(1 << 30)))
    b = s3d.make_scene_cloud(n, int(rng.integers(1 << 30))) if rng.random() < 0.3 else a + rng.normal(0, 0.005, a.shape).astype(np.float32)
    T = np.eye(4); T[:3, 3] = rng.uniform(-0.4, 0.4, 3)
    c, s_ = np.cos(rng.uniform(-0.03, 0.03)), 0.0
    ang = rng.uniform(-0.03, 0.03); T[:2, :2] = [[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]]
    bl = ((b.astype(np.float64) - T[:3, 3]) @ T[:3, :3]).astype(np.float32)
    alg = oracle.ALG_GICP if rng.random() < 0.7 else oracle.ALG_ICP
    dens = float(rng.choice([0.02, 0.1, 0.3]))
    its = int(rng.choice([5, 12, 20]))
    po = oracle.default_params(registration_algorithm=alg, point_cloud_density=dens, maximum_iterations=its)
    pg = s3d.default_params(registration_algorithm=alg, point_cloud_density=dens, maximum_iterations=its)
    so, To, io = oracle.align(a, bl, np.eye(4), po)
    sg, Tg, ig = ctx.align(a, bl, np.eye(4), pg)
    if so != sg: bad += 1; print('STATUS', case, so, sg); continue
    if so != 0: continue
    dt, dr = transform_delta(To, Tg)
    worst = [max(worst[0], dt), max(worst[1], dr)]
    if dt > 1e-4 or dr > 1e-4 or io['iterations'] != ig['iterations']:
        bad += 1; print('MISMATCH', case, n, alg, dens, its, dt, dr, io['iterations'], ig['iterations'])
print('cases', N, 'bad', bad, 'worst dt %.2e m dr %.2e rad' % tuple(worst))
