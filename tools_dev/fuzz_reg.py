"""dev tool: GICP (vs the smooth-objective oracle) and NDT (vs its oracle) on random synthetic pairs."""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
import slam3d_amd as s3d, oracle
from tests.conftest import transform_delta
ctx = s3d.Context(0)
N = int(os.environ.get('CASES', '24'))
worst = {'gicp': (0, 0), 'ndt': (0, 0)}
bad = 0
for i in range(N):
    n = [8000, 20000, 40000][i % 3]
    a, b, T = s3d.make_pair(n, 500 + i)
    dens = [0.1, 0.2, 0.05][i % 3]
    oracle.set_eval_precision(2)
    po = oracle.default_params(point_cloud_density=dens, maximum_iterations=30)
    so, To, io = oracle.align(a, b, np.eye(4), po)
    oracle.set_eval_precision(0)
    pg = s3d.default_params(point_cloud_density=dens, maximum_iterations=30)
    sg, Tg, ig = ctx.align(a, b, np.eye(4), pg)
    dt, dr = transform_delta(To, Tg)
    if so != sg or (so == 0 and (dt > 1e-4 or dr > 1e-4)): bad += 1; print('GICP', i, n, dens, so, sg, dt, dr, io['iterations'], ig['iterations'])
    worst['gicp'] = (max(worst['gicp'][0], dt), max(worst['gicp'][1], dr))
    pn = oracle.default_params(registration_algorithm=oracle.ALG_NDT, point_cloud_density=dens)
    so, To, io = oracle.align(a, b, np.eye(4), pn)
    sg, Tg, ig = ctx.align(a, b, np.eye(4), s3d.default_params(registration_algorithm=s3d.ALG_NDT, point_cloud_density=dens))
    dt, dr = transform_delta(To, Tg)
    if so != sg or (so == 0 and (dt > 1e-4 or dr > 1e-4)): bad += 1; print('NDT', i, n, dens, so, sg, dt, dr, io['iterations'], ig['iterations'])
    worst['ndt'] = (max(worst['ndt'][0], dt), max(worst['ndt'][1], dr))
print('cases', N, 'mismatches', bad, 'worst deltas', worst)
