"""dev tool: two-library A/B (old vs new build) — run as two processes alternating, reports step/icp/single-pair."""
import os, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NP=int(os.environ.get('NPAIRS','256'))
with ThreadPool(16) as pool: pairs=pool.map(lambda i: s3d.make_pair(100000,i), range(NP))
ctx=s3d.Context(0)
a=[ctx.upload(p[0]) for p in pairs]; b=[ctx.upload(p[1]) for p in pairs]
for alg,name in ((s3d.ALG_GICP,'gicp'),(s3d.ALG_ICP,'p2p')):
    p=s3d.default_params(registration_algorithm=alg, point_cloud_density=0.02, maximum_iterations=20)
    o=s3d.ExecOptions(force_iterations=1, profile=1)
    r=[]
    for i in range(4):
        t=time.perf_counter(); ctx.align_batch(a,b,None,p,o); dt=(time.perf_counter()-t)*1e3
        pr=ctx.last_profile(); r.append((dt,pr['nn_ms'],pr['normals_ms'],pr['icp_ms']))
    r=np.array(r)[1:].mean(0)
    o=s3d.ExecOptions(force_iterations=1, profile=0)
    ctx.align_batch(a[:1],b[:1],None,p,o)
    t=time.perf_counter()
    for i in range(10): ctx.align_batch(a[:1],b[:1],None,p,o)
    one=(time.perf_counter()-t)*100
    print(os.environ.get('S3D_LIB_PATH','default'),name,'step %.2f nn %.2f normals %.2f icp %.2f single %.3f ms'%(*r,one))
