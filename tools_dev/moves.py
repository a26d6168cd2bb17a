"""dev tool: how much does each pair's transform still move per outer iteration at the end of the forced 20?"""
import os, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NP=int(os.environ.get('NPAIRS','256'))
with ThreadPool(16) as pool: pairs=pool.map(lambda i: s3d.make_pair(100000,i), range(NP))
ctx=s3d.Context(0)
a=[ctx.upload(x[0]) for x in pairs]; b=[ctx.upload(x[1]) for x in pairs]
for alg in (s3d.ALG_GICP, s3d.ALG_ICP):
    T={}
    for it in (10, 15, 18, 19, 20):
        p=s3d.default_params(registration_algorithm=alg, point_cloud_density=0.02, maximum_iterations=it)
        o=s3d.ExecOptions(force_iterations=1, profile=1)
        rec=ctx.align_batch(a,b,None,p,o); T[it]=rec[:,9:12].copy()
        pr=ctx.last_profile(); l=pr['nn_launch_ms']
    for x,y in ((10,15),(15,18),(18,19),(19,20)):
        mv=np.linalg.norm(T[y]-T[x],axis=1)/(y-x)
        print('alg',alg,'iters %d->%d: translation move/iter: median %.2e  p90 %.2e  max %.2e  #>1e-4: %d  #>1e-3: %d  #==0: %d'%(x,y,np.median(mv),np.percentile(mv,90),mv.max(),(mv>1e-4).sum(),(mv>1e-3).sum(),(mv==0).sum()))
    print('   steady nn', np.mean(l[8:]))
