"""dev tool: normals stage (K4) time of the default batch for the library in S3D_LIB_PATH (A/B of builds), NPAIRS pairs."""
import os, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NP = int(os.environ.get('NPAIRS', '128'))
with ThreadPool(16) as pool: pairs = pool.map(lambda i: s3d.make_pair(100000, i), range(NP))
ctx = s3d.Context(0)
a = [ctx.upload(p[0]) for p in pairs]; b = [ctx.upload(p[1]) for p in pairs]
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=20)
o = s3d.ExecOptions(force_iterations=1, profile=1)
r = []
for i in range(4):
    t = time.perf_counter(); out = ctx.align_batch(a, b, None, p, o); dt = (time.perf_counter() - t) * 1e3
    pr = ctx.last_profile(); r.append((dt, pr['voxel_ms'], pr['grid_ms'], pr['normals_ms'], pr['nn_ms'], pr['icp_ms']))
r = np.array(r)[1:].mean(0)
print('nn launches ms:', ' '.join('%.3f' % x for x in pr['nn_launch_ms'][:20]))
print(os.path.basename(os.environ.get('S3D_LIB_PATH', 'default')), 'step %.2f voxel %.2f grid %.2f normals %.2f nn %.2f icp %.2f ms' % tuple(r), ' hash %.17g' % float(np.abs(out[:, :12]).sum()), flush=True)
