import os, sys, time, numpy as np
sys.path.insert(0, '.')
import slam3d_amd as s3d
pairs=[s3d.make_pair(100000,0)]
ctx=s3d.Context(0)
a=[ctx.upload(p[0]) for p in pairs]; b=[ctx.upload(p[1]) for p in pairs]
for alg in (s3d.ALG_GICP, s3d.ALG_ICP):
    p=s3d.default_params(registration_algorithm=alg, point_cloud_density=0.02, maximum_iterations=20)
    o=s3d.ExecOptions(force_iterations=1, profile=1)
    for i in range(5): rec=ctx.align_batch(a,b,None,p,o)
    pr = ctx.last_profile()
    print(alg, {k: round(v,3) if isinstance(v,float) else v for k,v in pr.items() if not isinstance(v,(list,tuple,np.ndarray))})
    o=s3d.ExecOptions(force_iterations=1, profile=0)
    t=time.perf_counter()
    for i in range(50): rec=ctx.align_batch(a,b,None,p,o)
    print("ms per registration", (time.perf_counter()-t)/50*1e3)
    st, T, info = ctx.align_clouds(a[0], b[0], np.eye(4), p, o)
    print(st, info)
