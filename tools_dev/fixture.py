"""dev tool: real-data batch (the reference's fixture scans, default parameters): per-launch NN time and search counts."""
import os, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
G='tests/golden'
clouds=[np.load(os.path.join(G,'cloud%d.npz'%i))['xyzi'].astype(np.float32) for i in range(1,5)]
ctx=s3d.Context(0)
dev=[ctx.upload(c) for c in clouds]
REP=int(os.environ.get('REP','32'))
src=[]; tgt=[]
for r in range(REP):
    for a,b in ((0,1),(1,2),(2,3)):
        src.append(dev[a]); tgt.append(dev[b])
for alg in (s3d.ALG_GICP, s3d.ALG_ICP):
    p=s3d.default_params(registration_algorithm=alg, maximum_iterations=20)
    o=s3d.ExecOptions(force_iterations=1, profile=int(os.environ.get('PROFILE','1')), grid_cells_per_point=int(os.environ.get('CPP','0')))
    for i in range(3): rec=ctx.align_batch(src,tgt,None,p,o); pr=ctx.last_profile()
    print('alg',alg,'pairs',len(src),'total %.2f voxel %.2f grid %.2f normals %.2f icp %.2f nn %.2f'%(pr['total_ms'],pr['voxel_ms'],pr['grid_ms'],pr['normals_ms'],pr['icp_ms'],pr['nn_ms']))
    print('  nn ms', [round(x,3) for x in pr['nn_launch_ms']])
    print('  searched', pr['nn_searched']); print('  unseeded', pr['nn_unseeded'])
    print('  queries/launch', pr['nn_queries']//max(pr['nn_launches'],1), 'status', sorted(set(rec[:,15].astype(int))))
