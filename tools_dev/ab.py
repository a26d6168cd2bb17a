"""dev tool: in-process A/B of kernel variants selected by env vars (timing noise between processes is ~2x)."""
import os, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NP=int(os.environ.get('NPAIRS','64'))
with ThreadPool(16) as pool: pairs=pool.map(lambda i: s3d.make_pair(100000,i), range(NP))
ctx=s3d.Context(0)
a=[ctx.upload(p[0]) for p in pairs]; b=[ctx.upload(p[1]) for p in pairs]
alg=s3d.ALG_GICP
variants=[v for v in os.environ.get('VARIANTS','0').split(',')]
cpps=[int(c) for c in os.environ.get('CPPS','16').split(',')]
p=s3d.default_params(registration_algorithm=alg, point_cloud_density=0.02, maximum_iterations=20)
res={}
for rnd in range(3):
    for cpp in cpps:
        for v in variants:
            os.environ[os.environ.get('VARKEY','S3D_DBG_NN')]=v
            o=s3d.ExecOptions(force_iterations=1, profile=1, grid_cells_per_point=cpp)
            t=time.perf_counter(); rec=ctx.align_batch(a,b,None,p,o); dt=(time.perf_counter()-t)*1e3
            pr=ctx.last_profile()
            res.setdefault((cpp,v),[]).append((dt,pr['nn_ms'],pr['normals_ms'],pr['icp_ms']))
            if rnd==2: print(cpp,v,[round(x,2) for x in pr['nn_launch_ms']])
for k,v in res.items():
    v=np.array(v)[1:]
    print('cpp %3d variant %s: step %.2f nn %.2f normals %.2f icp %.2f'%(k[0],k[1],*v.mean(0)))
