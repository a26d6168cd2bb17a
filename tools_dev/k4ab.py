"""dev tool: the k-NN pre-pass (K4) with the exact 64-bit search (S3D_KNN_EXACT64=1) against the 32-bit med3 kernel:
normals stage time on NPAIRS pairs of the default batch, and the largest difference of the normals of one cloud."""
import os, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NP = int(os.environ.get('NPAIRS', '128'))
with ThreadPool(16) as pool: pairs = pool.map(lambda i: s3d.make_pair(100000, i), range(NP))
ctx = s3d.Context(0)
a = [ctx.upload(p[0]) for p in pairs]; b = [ctx.upload(p[1]) for p in pairs]
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=20)
o = s3d.ExecOptions(force_iterations=1, profile=1)
res = {}
for mode in ('1', '0', '1', '0'):
    os.environ['S3D_KNN_EXACT64'] = mode
    r = []
    for i in range(3):
        t = time.perf_counter(); out = ctx.align_batch(a, b, None, p, o); dt = (time.perf_counter() - t) * 1e3
        pr = ctx.last_profile(); r.append((dt, pr['nn_ms'], pr['normals_ms'], pr['icp_ms']))
    r = np.array(r)[1:].mean(0)
    res[mode] = out
    print('exact64=%s step %.2f nn %.2f normals %.2f icp %.2f ms' % (mode, *r), flush=True)
T0 = res['1'][:, :12]; T1 = res['0'][:, :12]
print('max |T(exact64) - T(med3)| over the batch: %.3e' % np.abs(T0 - T1).max())
import oracle
for name, cloud, leaf in (('synthetic', pairs[0][0], 0.02), ('fixture', np.load('tests/golden/cloud1.npz')['xyzi'][:, :3], 0.2)):
    v, _ = oracle.voxel_downsample(np.ascontiguousarray(cloud, dtype=np.float32), leaf)
    os.environ['S3D_KNN_EXACT64'] = '1'; n0 = ctx.knn_normals(v, 20).astype(np.float64)
    os.environ['S3D_KNN_EXACT64'] = '0'; os.environ['S3D_DBG_KNN'] = '1'; n1 = ctx.knn_normals(v, 20).astype(np.float64); del os.environ['S3D_DBG_KNN']
    d = np.abs(n0 - n1).max(1)
    print(name, 'normals: max diff %.3e, rows differing by more than 1e-6: %d of %d' % (d.max(), (d > 1e-6).sum(), len(d)))
