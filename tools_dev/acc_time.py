"""dev tool: time of the GICP accumulate + control part per step (icp - nn) for the loaded library."""
import os, sys, numpy as np
sys.path.insert(0, '.')
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
with ThreadPool(16) as pool: pairs = pool.map(lambda i: s3d.make_pair(100000, i), range(256))
ctx = s3d.Context(0)
a = [ctx.upload(x[0]) for x in pairs]; b = [ctx.upload(x[1]) for x in pairs]
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=20)
o = s3d.ExecOptions(force_iterations=1, profile=1)
for i in range(3): ctx.align_batch(a, b, None, p, o); pr = ctx.last_profile()
print(os.environ.get('S3D_LIB_PATH', 'default'), 'icp-nn %.2f ms per step (20 accumulate + 20 control launches)' % (pr['icp_ms'] - pr['nn_ms']))
