"""dev tool (round 5): the ICP loop's progress word (check_interval = 0) against the stream-wait poll (check_interval = 4):
one registration of two of the reference's scans, pairs 0-2, cache off / on; the synthetic pair with early exit; results equal?"""
import os, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')
fc = [np.load(os.path.join(G, 'cloud%d.npz' % i))['xyzi'].astype(np.float32) for i in range(1, 5)]
ctx = s3d.Context(0)
dev = [ctx.upload(c) for c in fc]
a, b, _ = s3d.make_pair(100000, 0)
dev += [ctx.upload(a), ctx.upload(b)]
p = s3d.default_params()
ps = s3d.default_params(point_cloud_density=0.02)
def run(i, j, par, ci, cache):
    o = s3d.ExecOptions(cache_prepass=cache, check_interval=ci)
    for _ in range(3): st = ctx.align_clouds(dev[i], dev[j], np.eye(4), par, o)
    ts = []
    for _ in range(30):
        t = time.perf_counter(); st = ctx.align_clouds(dev[i], dev[j], np.eye(4), par, o); ts.append((time.perf_counter() - t) * 1e3)
    return st, np.median(ts), np.min(ts)
for rep in range(2):
    for (i, j, par, name) in [(0, 1, p, 'real 1->2'), (1, 2, p, 'real 2->3'), (2, 3, p, 'real 3->4'), (4, 5, ps, 'synthetic 100k')]:
        for cache in (0, 1):
            r = {ci: run(i, j, par, ci, cache) for ci in (4, 0)}
            same = r[4][0][0] == r[0][0][0] and np.array_equal(r[4][0][1], r[0][0][1]) and r[4][0][2] == r[0][0][2]
            print('%-15s cache %d  it %2d  poll/4: %.3f (min %.3f)  progress word: %.3f (min %.3f) ms  same %s' %
                  (name, cache, r[0][0][2]['iterations'], r[4][1], r[4][2], r[0][1], r[0][2], same), flush=True)
