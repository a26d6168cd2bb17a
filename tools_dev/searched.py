"""dev tool: how many queries still search (and how many without a near seed) in every NN launch of the default batch."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NP = 64
with ThreadPool(16) as pool: pairs = pool.map(lambda i: s3d.make_pair(100000, i), range(NP))
ctx = s3d.Context(0)
a = [ctx.upload(x[0]) for x in pairs]; b = [ctx.upload(x[1]) for x in pairs]
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=20)
o = s3d.ExecOptions(force_iterations=1, profile=2)
for _ in range(2): ctx.align_batch(a, b, None, p, o)
pr = ctx.last_profile()
nq = pr['nn_queries'] / pr['nn_launches']
for i in range(pr['nn_launches']):
    print("pass %2d: %.3f ms  searched %7.4f %% (%d)  of which without a near seed %5.1f %%" % (i + 1, pr['nn_launch_ms'][i], 100.0 * pr["nn_searched"][i] / nq, pr["nn_searched"][i], 100.0 * pr['nn_unseeded'][i] / max(pr['nn_searched'][i], 1)))
