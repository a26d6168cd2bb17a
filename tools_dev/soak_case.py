"""dev tool (round 5): one case of parity_soak_big.py (SEED, CASE) in detail: GPU and smooth oracle with the iteration cap
at 1 ... its, so that a difference in the early-exit decision can be told from a difference in the iterates."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import slam3d_amd as s3d, oracle
from conftest import transform_delta
seed = int(os.environ.get('SEED', '316')); want = int(os.environ.get('CASE', '23'))
rng = np.random.default_rng(seed)
for case in range(24):
    n = int(rng.choice([3000, 20000, 60000]))
    a = s3d.make_scene_cloud(n, int(rng.integers(1 << 30)))
    b = s3d.make_scene_cloud(n, int(rng.integers(1 << 30))) if rng.random() < 0.3 else \
        a + rng.normal(0, 0.005, a.shape).astype(np.float32)
    T = np.eye(4); T[:3, 3] = rng.uniform(-0.4, 0.4, 3)
    ang = rng.uniform(-0.03, 0.03)
    T[:2, :2] = [[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]]
    bl = ((b.astype(np.float64) - T[:3, 3]) @ T[:3, :3]).astype(np.float32)
    alg = oracle.ALG_GICP if rng.random() < 0.7 else oracle.ALG_ICP
    dens = float(rng.choice([0.02, 0.1, 0.3])); its = int(rng.choice([5, 12, 20]))
    if case == want: break
print('seed %d case %d: n %d alg %d density %g max iterations %d' % (seed, case, n, alg, dens, its))
ctx = s3d.Context(0)
oracle.set_eval_precision(2)
prev_o = prev_g = None
for cap in range(1, min(its, 8) + 1):
    po = oracle.default_params(registration_algorithm=alg, point_cloud_density=dens, maximum_iterations=cap)
    so, To, io = oracle.align(a, bl, np.eye(4), po)
    sg, Tg, ig = ctx.align(a, bl, np.eye(4), s3d.default_params(registration_algorithm=alg, point_cloud_density=dens, maximum_iterations=cap))
    dt, dr = transform_delta(To, Tg)
    step_o = transform_delta(prev_o, To)[0] if prev_o is not None else float('nan')
    step_g = transform_delta(prev_g, Tg)[0] if prev_g is not None else float('nan')
    print('cap %d: oracle status %d it %d conv %s | gpu status %d it %d conv %s | gpu - oracle %.3e m %.3e rad | last step oracle %.3e gpu %.3e m' %
          (cap, so, io['iterations'], io.get('converged'), sg, ig['iterations'], ig.get('converged'), dt, dr, step_o, step_g))
    prev_o, prev_g = To, Tg
# which of the two final transforms is the better minimiser of the oracle's own objective (correspondences re-established)?
po = oracle.default_params(registration_algorithm=alg, point_cloud_density=dens, maximum_iterations=its)
so, To, io = oracle.align(a, bl, np.eye(4), po)
sg, Tg, ig = ctx.align(a, bl, np.eye(4), s3d.default_params(registration_algorithm=alg, point_cloud_density=dens, maximum_iterations=its))
try:
    co, no = oracle.gicp_cost(a, bl, To, po) if oracle.gicp_cost.__code__.co_argcount > 3 else oracle.gicp_cost(a, bl, To)
    cg, ng = oracle.gicp_cost(a, bl, Tg, po) if oracle.gicp_cost.__code__.co_argcount > 3 else oracle.gicp_cost(a, bl, Tg)
    print('objective at the oracle result %.12g (%d correspondences), at the gpu result %.12g (%d): gpu / oracle %.9f' % (co, no, cg, ng, cg / co))
except Exception as e:
    print('gicp_cost:', e)
