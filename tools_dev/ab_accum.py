"""dev tool: step time of the default batch for several numbers of REAL accumulate blocks per pair (S3D_ACCUM_BLOCKS;
the results are identical for every value - block_reduce_store_fixed)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NP = int(os.environ.get('PAIRS', '256'))
with ThreadPool(16) as pool: pairs = pool.map(lambda i: s3d.make_pair(100000, i), range(NP))
ctx = s3d.Context(0)
a = [ctx.upload(x[0]) for x in pairs]; b = [ctx.upload(x[1]) for x in pairs]
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=20)
o = s3d.ExecOptions(force_iterations=1, profile=1)
ref = None
for blocks in os.environ.get('BLOCKS', '0,4,8,16,32,64').split(','):
    if blocks == '0': os.environ.pop('S3D_ACCUM_BLOCKS', None)
    else: os.environ['S3D_ACCUM_BLOCKS'] = blocks
    for _ in range(2): ctx.align_batch(a, b, None, p, o)
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); rec = ctx.align_batch(a, b, None, p, o); t.append((time.perf_counter() - t0) * 1e3)
    pr = ctx.last_profile()
    if ref is None: ref = rec
    print('blocks', blocks, 'step %.2f ms' % np.median(t), 'voxel %.2f grid %.2f normals %.2f icp %.2f nn %.2f rest %.2f' % (pr['voxel_ms'], pr['grid_ms'], pr['normals_ms'], pr['icp_ms'], pr['nn_ms'], pr['icp_ms'] - pr['nn_ms']),
          'identical', bool(np.array_equal(rec, ref)), flush=True)
