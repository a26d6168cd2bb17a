"""dev tool (round 5): two full batches in flight on ONE device (two contexts, two host threads, each looping over its own
batches) against the same batches one after the other - what pipelining consecutive batches of a sweep would buy.
env: NPAIRS (256), POINTS (100000), ITERS (20), STEPS (6)"""
import os, sys, time, threading, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NP = int(os.environ.get('NPAIRS', '256')); PTS = int(os.environ.get('POINTS', '100000')); IT = int(os.environ.get('ITERS', '20'))
STEPS = int(os.environ.get('STEPS', '6'))
with ThreadPool(16) as pool: pairs = pool.map(lambda i: s3d.make_pair(PTS, i), range(NP))
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=IT)
o = s3d.ExecOptions(force_iterations=1)
ctxs = [s3d.Context(0), s3d.Context(0)]
ab = [([c.upload(q[0]) for q in pairs], [c.upload(q[1]) for q in pairs]) for c in ctxs]
for c, (a, b) in zip(ctxs, ab):
    for i in range(2): ref = c.align_batch(a, b, None, p, o)
for rep in range(3):
    t = time.perf_counter()
    for i in range(STEPS):
        for c, (a, b) in zip(ctxs, ab): c.align_batch(a, b, None, p, o)
    seq = (time.perf_counter() - t) * 1e3 / (2 * STEPS)
    def loop(k):
        for i in range(STEPS): ctxs[k].align_batch(ab[k][0], ab[k][1], None, p, o)
    t = time.perf_counter()
    th = [threading.Thread(target=loop, args=(k,)) for k in range(2)]
    for x in th: x.start()
    for x in th: x.join()
    par = (time.perf_counter() - t) * 1e3 / (2 * STEPS)
    print('one after the other %.2f ms per batch; two in flight %.2f ms per batch (%.1f %%)' % (seq, par, 100 * (seq / par - 1)), flush=True)
