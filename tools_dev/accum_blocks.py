"""dev tool: the ICP stage of the default batch for several REAL accumulate blocks per pair (s3d_exec_options.debug_accum_blocks;
the sums are defined over 32 virtual blocks whatever this is, so the records do not change).  env: NPAIRS (256)"""
import os, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NP = int(os.environ.get('NPAIRS', '256'))
with ThreadPool(16) as pool: pairs = pool.map(lambda i: s3d.make_pair(100000, i), range(NP))
ctx = s3d.Context(0)
both = ctx.upload_many([p[0] for p in pairs] + [p[1] for p in pairs]); a, b = both[:NP], both[NP:]
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=20)
for rep in range(2):
    for ab in (0, 1, 2, 4, 8, 16, 32):
        o = s3d.ExecOptions(force_iterations=1, profile=1, debug_accum_blocks=ab)
        r = []
        for i in range(3):
            out = ctx.align_batch(a, b, None, p, o); pr = ctx.last_profile(); r.append((pr['total_ms'], pr['icp_ms'], pr['nn_ms']))
        r = np.array(r)[1:].mean(0)
        print('accum blocks %2d: step %.2f icp %.2f (nn %.2f)  hash %.17g' % (ab, r[0], r[1], r[2], float(np.abs(out[:, :12]).sum())), flush=True)
