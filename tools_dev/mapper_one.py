"""dev tool: the mapper pattern with the cache on, a few new scans (for a kernel trace: tools_dev/prof_mapper.sh)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NS, NB = 14, 8
with ThreadPool(16) as pool: scans = pool.map(lambda i: s3d.make_scene_cloud(100000, 3000 + i), range(NS))
ctx = s3d.Context(0)
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=20)
cl = [ctx.upload(x) for x in scans]
o = s3d.ExecOptions(force_iterations=1, cache_prepass=1)
for i in range(NB, NS):
    t0 = time.perf_counter()
    r = ctx.align_batch([cl[i]] * NB, cl[i - NB:i], None, p, o)
    print('%.3f ms' % ((time.perf_counter() - t0) * 1e3), flush=True)
