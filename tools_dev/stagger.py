"""dev tool (round 5): two half batches on two contexts (two streams) of ONE device, the second started `d` ms after
the first: what does overlapping one half's k-NN pre-pass (VALU-bound) with the other's ICP loop (memory-bound) buy?
env: NPAIRS (256), POINTS (100000), ITERS (20)"""
import os, sys, time, threading, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NP = int(os.environ.get('NPAIRS', '256')); PTS = int(os.environ.get('POINTS', '100000')); IT = int(os.environ.get('ITERS', '20'))
with ThreadPool(16) as pool: pairs = pool.map(lambda i: s3d.make_pair(PTS, i), range(NP))
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=IT)
o = s3d.ExecOptions(force_iterations=1)
c0 = s3d.Context(0); c1 = s3d.Context(0); c2 = s3d.Context(0)
H = NP // 2
a0 = [c0.upload(q[0]) for q in pairs]; b0 = [c0.upload(q[1]) for q in pairs]
a1 = [c1.upload(q[0]) for q in pairs[:H]]; b1 = [c1.upload(q[1]) for q in pairs[:H]]
a2 = [c2.upload(q[0]) for q in pairs[H:]]; b2 = [c2.upload(q[1]) for q in pairs[H:]]
for i in range(3): ref = c0.align_batch(a0, b0, None, p, o)
ts = []
for i in range(5):
    t = time.perf_counter(); c0.align_batch(a0, b0, None, p, o); ts.append((time.perf_counter() - t) * 1e3)
print('one batch of %d: %.2f ms' % (NP, np.mean(ts)), flush=True)
for i in range(2): c1.align_batch(a1, b1, None, p, o); c2.align_batch(a2, b2, None, p, o)
ts = []
for i in range(5):
    t = time.perf_counter(); c1.align_batch(a1, b1, None, p, o); ts.append((time.perf_counter() - t) * 1e3)
print('one half alone: %.2f ms' % np.mean(ts), flush=True)
out = [None, None]
def run(ctx, a, b, k, d, t0):
    while (time.perf_counter() - t0) * 1e3 < d: pass
    out[k] = ctx.align_batch(a, b, None, p, o)
for d in [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12]:
    ts = []
    for i in range(6):
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(c1, a1, b1, 0, 0.0, t0)), threading.Thread(target=run, args=(c2, a2, b2, 1, float(d), t0))]
        for x in th: x.start()
        for x in th: x.join()
        ts.append((time.perf_counter() - t0) * 1e3)
    same = np.array_equal(np.concatenate(out), ref)
    print('stagger %2d ms: %s  mean(last 4) %.2f ms  same %s' % (d, ' '.join('%.2f' % x for x in ts), np.mean(ts[2:]), same), flush=True)
