import os, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
pairs=[s3d.make_pair(100000,0)]
ctx=s3d.Context(0)
a=[ctx.upload(p[0]) for p in pairs]; b=[ctx.upload(p[1]) for p in pairs]
p=s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=20)
o=s3d.ExecOptions(force_iterations=1, profile=0)
for i in range(20): rec=ctx.align_batch(a,b,None,p,o)
print(rec[0][12:])
