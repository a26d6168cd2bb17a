"""dev tool: create / use / destroy contexts, sweeps, clouds and cache blobs repeatedly; HBM in use must not grow."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch, slam3d_amd as s3d
a, b, _ = s3d.make_pair(20000, 0)
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.05, maximum_iterations=6)
used = []
for it in range(int(os.environ.get('ROUNDS', '30'))):
    ctx = s3d.Context(0)
    ca, cb = ctx.upload(a), ctx.upload(b)
    r0 = ctx.align_batch([ca, cb], [cb, ca], None, p, s3d.ExecOptions(cache_prepass=1))
    blob = ctx.cache_export(ca)
    ca.release(); ca = ctx.upload(a)
    assert ctx.cache_import(ca, blob) == 0
    r1 = ctx.align_batch([ca, cb], [cb, ca], None, p, s3d.ExecOptions(cache_prepass=1))
    assert np.array_equal(r0, r1)
    sw = s3d.Sweep([0, 0])
    sa, sb = sw.upload(a), sw.upload(b)
    r2 = sw.align_batch([sa, sb], [sb, sa], None, p)
    assert np.array_equal(np.asarray(r2)[:, :12], np.asarray(r0)[:, :12])
    sw.close(); ca.release(); cb.release(); del ctx
    free, total = torch.cuda.mem_get_info(0)
    used.append((total - free) >> 20)
print('HBM in use (MiB) per round:', used[:3], '...', used[-3:])
assert used[-1] <= used[2] + 64, used
print('soak OK')
