"""dev tool: tests/test_gpu_fuzz.py::test_registration_soak_240_random_pairs over many more seeds (SEEDS, default 100..119 = 480
registrations): GPU against the oracle's smooth-objective variant; prints every case that differs in status / iteration
count or by more than 1e-4 m / 1e-4 rad, and the worst deltas."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import slam3d_amd as s3d, oracle
from multiprocessing.pool import ThreadPool
from conftest import transform_delta
s0 = int(os.environ.get('SEED0', '100')); ns = int(os.environ.get('SEEDS', '20'))
cases = []
for seed in range(s0, s0 + ns):
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
        cases.append((seed, case, a, bl, alg, float(rng.choice([0.02, 0.1, 0.3])), int(rng.choice([5, 12, 20]))))


def ref(c):
    _, _, a, bl, alg, dens, its = c
    return oracle.align(a, bl, np.eye(4), oracle.default_params(registration_algorithm=alg, point_cloud_density=dens, maximum_iterations=its))


oracle.set_eval_precision(2)
with ThreadPool(int(os.environ.get('THREADS', '64'))) as pool:
    refs = pool.map(ref, cases)
oracle.set_eval_precision(0)
ctx = s3d.Context(0)
worst = [0.0, 0.0]; bad = 0
for (seed, case, a, bl, alg, dens, its), (so, To, io) in zip(cases, refs):
    sg, Tg, ig = ctx.align(a, bl, np.eye(4), s3d.default_params(registration_algorithm=alg, point_cloud_density=dens, maximum_iterations=its))
    if sg != so:
        bad += 1; print('STATUS', seed, case, alg, dens, its, len(a), sg, so); continue
    if so != 0:
        continue
    dt, dr = transform_delta(To, Tg)
    worst = [max(worst[0], dt), max(worst[1], dr)]
    if ig['iterations'] != io['iterations'] or dt > 1e-4 or dr > 1e-4:
        bad += 1; print('DIFF', seed, case, alg, dens, its, len(a), 'iterations', ig['iterations'], io['iterations'], 'dt %.3e dr %.3e' % (dt, dr))
print('%d registrations, %d differ; worst %.3e m %.3e rad' % (len(cases), bad, worst[0], worst[1]))
