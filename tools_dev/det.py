"""dev tool: is the batch result bitwise identical across processes, and does the steady NN time follow it?"""
import os, sys, hashlib, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NP=int(os.environ.get('NPAIRS','128'))
with ThreadPool(16) as pool: pairs=pool.map(lambda i: s3d.make_pair(100000,i), range(NP))
h=hashlib.sha256()
for x in pairs: h.update(x[0].tobytes()); h.update(x[1].tobytes())
print('inputs sha', h.hexdigest()[:12], flush=True)
ctx=s3d.Context(0)
a=[ctx.upload(x[0]) for x in pairs]; b=[ctx.upload(x[1]) for x in pairs]
for alg in (s3d.ALG_GICP, s3d.ALG_ICP):
    p=s3d.default_params(registration_algorithm=alg, point_cloud_density=0.02, maximum_iterations=20)
    o=s3d.ExecOptions(force_iterations=1, profile=int(os.environ.get('PROFILE','1')))
    for i in range(2):
        rec,infos=ctx.align_batch(a,b,None,p,o,want_infos=True); pr=ctx.last_profile()
        l=pr['nn_launch_ms']
        print('alg %d run %d: sha %s steady %.3f  inner %d evals %d'%(alg, i, hashlib.sha256(rec.tobytes()).hexdigest()[:12], np.mean(l[8:]), sum(x['inner_iterations'] for x in infos), sum(x['evaluations'] for x in infos)), flush=True)
        if i==1: print('   searched', pr['nn_searched'], '\n   unseeded', pr['nn_unseeded'], flush=True)

# stage hashes on pair 0
v=ctx.voxel_downsample(pairs[0][0], 0.02); print('voxel sha', hashlib.sha256(v.tobytes()).hexdigest()[:12], len(v))
nr=ctx.knn_normals(v, 20); print('normals sha', hashlib.sha256(nr.tobytes()).hexdigest()[:12])
idx,d2=ctx.nn_search(v, pairs[0][1], 2.5); print('nn sha', hashlib.sha256(idx.tobytes()+d2.tobytes()).hexdigest()[:12])
for alg in (s3d.ALG_GICP, s3d.ALG_ICP):
    p=s3d.default_params(registration_algorithm=alg, point_cloud_density=0.02, maximum_iterations=20)
    o=s3d.ExecOptions(force_iterations=1)
    rec=ctx.align_batch(a[:1],b[:1],None,p,o); print('single alg',alg,'sha', hashlib.sha256(rec.tobytes()).hexdigest()[:12], rec[0][9:12])
    rec=ctx.align_batch(a[:8],b[:8],None,p,o); print('eight alg',alg,'sha', hashlib.sha256(rec.tobytes()).hexdigest()[:12])
