"""dev tool: does the NN steady-state launch time depend on the arena placement? re-create context in one process."""
import os, sys, time, numpy as np
import torch; torch.cuda.init(); torch.zeros(1, device="cuda")
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NP=int(os.environ.get('NPAIRS','256'))
with ThreadPool(16) as pool: pairs=pool.map(lambda i: s3d.make_pair(100000,i), range(NP))
p=s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=20)
o=s3d.ExecOptions(force_iterations=1, profile=1)
keep=[]
import torch
def bw_probe():
    x=torch.empty(512*1024*1024//4, dtype=torch.float32, device='cuda'); y=torch.empty_like(x); x.fill_(1.0)
    res=[]
    for name,fn,nbytes in (('copy', lambda: y.copy_(x), 2*x.numel()*4), ('read', lambda: x.sum(), x.numel()*4), ('write', lambda: y.fill_(2.0), x.numel()*4)):
        fn(); torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        res.append('%s %.2f TB/s'%(name, nbytes*10/(e0.elapsed_time(e1)*1e-3)/1e12))
    print('   bw:', ', '.join(res), flush=True)
for rnd in range(int(os.environ.get("ROUNDS","1"))):
    ctx=s3d.Context(0)
    a=[ctx.upload(x[0]) for x in pairs]; b=[ctx.upload(x[1]) for x in pairs]
    for i in range(3):
        ctx.align_batch(a,b,None,p,o); pr=ctx.last_profile()
    l=pr['nn_launch_ms']
    bw_probe()
    print('ctx %d: nn %.2f first4 %s steady %.3f normals %.2f icp %.2f voxel %.2f grid %.2f'%(rnd, pr['nn_ms'], [round(x,2) for x in l[:4]], np.mean(l[8:]), pr['normals_ms'], pr['icp_ms'], pr['voxel_ms'], pr['grid_ms']), flush=True)
    if not os.environ.get('KEEP'):
        for c in a+b: c.release()
        ctx.close()
    else:
        keep.append((ctx,a,b))
