"""dev tool: run only the k-NN normals stage a few times (for PMC collection)."""
import os, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NP=64
with ThreadPool(16) as pool: pairs=pool.map(lambda i: s3d.make_pair(100000,i), range(NP))
ctx=s3d.Context(0)
a=[ctx.upload(x[0]) for x in pairs]; b=[ctx.upload(x[1]) for x in pairs]
p=s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=1)
o=s3d.ExecOptions(force_iterations=1)
for i in range(2): ctx.align_batch(a,b,None,p,o)
