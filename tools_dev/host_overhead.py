"""dev tool (round 5): wall time of one s3d_align_batch call against the span of its GPU work (HIP events around the stages)."""
import os, sys, time, numpy as np, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NP = int(os.environ.get('NPAIRS', '256')); PTS = int(os.environ.get('POINTS', '100000')); IT = int(os.environ.get('ITERS', '20'))
with ThreadPool(16) as pool: pairs = pool.map(lambda i: s3d.make_pair(PTS, i), range(NP))
ctx = s3d.Context(0)
a = [ctx.upload(p[0]) for p in pairs]; b = [ctx.upload(p[1]) for p in pairs]
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=IT)
for prof in (1, 0, 1, 0):
    o = s3d.ExecOptions(force_iterations=1, profile=prof)
    ws, gs = [], []
    for i in range(8):
        t = time.perf_counter(); ctx.align_batch(a, b, None, p, o); ws.append((time.perf_counter() - t) * 1e3)
        if prof: gs.append(ctx.last_profile()['total_ms'])
    print('profile %d: wall %.3f ms' % (prof, np.mean(ws[2:])), ('gpu span %.3f ms' % np.mean(gs[2:])) if prof else '', flush=True)
# back to back without python in between: a loop of calls, wall per call
o = s3d.ExecOptions(force_iterations=1)
t = time.perf_counter()
for i in range(10): ctx.align_batch(a, b, None, p, o)
print('10 calls: %.3f ms per call' % ((time.perf_counter() - t) * 100))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(5): ctx.align_batch(a, b, None, p, o)
pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(8)
