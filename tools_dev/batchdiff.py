import sys, numpy as np
sys.path.insert(0, ".")
import slam3d_amd as s3d
pairs=[s3d.make_pair(100000,i) for i in range(8)]
ctx=s3d.Context(0)
a=[ctx.upload(p[0]) for p in pairs]; b=[ctx.upload(p[1]) for p in pairs]
for alg in (s3d.ALG_GICP, s3d.ALG_ICP):
    p=s3d.default_params(registration_algorithm=alg, point_cloud_density=0.02, maximum_iterations=20)
    o=s3d.ExecOptions(force_iterations=1, profile=0)
    full=ctx.align_batch(a,b,None,p,o)
    full2=ctx.align_batch(a,b,None,p,o)
    h1=ctx.align_batch(a[:4],b[:4],None,p,o); h2=ctx.align_batch(a[4:],b[4:],None,p,o)
    one=np.vstack([ctx.align_batch(a[i:i+1],b[i:i+1],None,p,o) for i in range(8)])
    half=np.vstack([h1,h2])
    print(alg, "repeat identical", np.array_equal(full,full2), "halves identical", np.array_equal(full,half), "singles identical", np.array_equal(full, one))
    d=np.abs(full-half); print(" max diff per column", d.max(0))
    d=np.abs(full-one); print(" max diff per column (singles)", d.max(0))
