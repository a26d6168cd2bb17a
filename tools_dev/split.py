"""dev tool (round 5): does a large batch gain from running as several concurrent sub-batches on ONE device?
s3d_align_batch_multi with the device listed 1, 2, 3, 4 times (one context + host thread + stream per rank).
env: NPAIRS (256), POINTS (100000), ITERS (20)"""
import os, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NP = int(os.environ.get('NPAIRS', '256')); PTS = int(os.environ.get('POINTS', '100000')); IT = int(os.environ.get('ITERS', '20'))
with ThreadPool(16) as pool: pairs = pool.map(lambda i: s3d.make_pair(PTS, i), range(NP))
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=IT)
o = s3d.ExecOptions(force_iterations=1)
ref = None
for ranks in [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4, 1, 2]:
    sw = s3d.Sweep([0] * ranks)
    a = [sw.upload(q[0]) for q in pairs]; b = [sw.upload(q[1]) for q in pairs]
    ts = []
    for i in range(6):
        t = time.perf_counter(); rec = sw.align_batch(a, b, None, p, o); ts.append((time.perf_counter() - t) * 1e3)
    if ref is None: ref = rec.copy()
    print('ranks %d (%s): step %s  mean(last 4) %.2f ms  same records %s' %
          (ranks, sw.collective, ' '.join('%.2f' % x for x in ts), np.mean(ts[2:]), np.array_equal(ref, rec)), flush=True)
    sw.close()
