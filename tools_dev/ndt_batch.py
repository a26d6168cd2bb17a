"""dev tool: batch of synthetic pairs through NDT (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NP = int(os.environ.get('NPAIRS', '64'))
with ThreadPool(16) as pool: pairs = pool.map(lambda i: s3d.make_pair(100000, i), range(NP))
ctx = s3d.Context(0)
sa = [ctx.upload(x[0]) for x in pairs]; sb = [ctx.upload(x[1]) for x in pairs]
pg = s3d.default_params(registration_algorithm=s3d.ALG_NDT, point_cloud_density=0.02)
ctx.align_batch(sa, sb, None, pg)
t = time.time(); rec, infos = ctx.align_batch(sa, sb, None, pg, want_infos=True); dt = time.time() - t
print('batch of %d NDT: %.1f ms (%.2f ms per pair), ok %d, evals max %d mean %.1f' % (NP, dt * 1e3, dt * 1e3 / NP, int((rec[:, 15] == 0).sum()), max(i['evaluations'] for i in infos), np.mean([i['evaluations'] for i in infos])))
