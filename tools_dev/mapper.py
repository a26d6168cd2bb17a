"""dev tool: the mapper pattern (each new scan against 8 earlier ones, one batch call per new scan), cache off / on."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NS, NB = 24, 8
with ThreadPool(16) as pool: scans = pool.map(lambda i: s3d.make_scene_cloud(100000, 3000 + i) if hasattr(s3d, 'make_scene_cloud') else s3d.make_pair(100000, i)[0], range(NS))
ctx = s3d.Context(0)
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=20)
for rep in range(3):
    for cache in (0, 1):
        ctx.cache_control(clear=True)
        cl = [ctx.upload(x) for x in scans]
        o = s3d.ExecOptions(force_iterations=1, cache_prepass=cache)
        ref = []
        t = []
        for i in range(NB, NS):
            t0 = time.perf_counter()
            r = ctx.align_batch([cl[i]] * NB, cl[i - NB:i], None, p, o)
            t.append((time.perf_counter() - t0) * 1e3)
        print('cache', cache, 'ms per new scan: median %.3f min %.3f' % (np.median(t[2:]), min(t[2:])), flush=True)
        for c in cl: c.release()
