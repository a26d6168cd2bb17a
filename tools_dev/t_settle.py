"""dev tool (round 5): does transformation_ still change bit-wise in the forced late iterations?  One synthetic pair,
max_iterations = 4 ... 20 (each run is a prefix of the next), consecutive final transforms compared."""
import os, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
ctx = s3d.Context(0)
for seed in (0, 1, 2):
    a, b, _ = s3d.make_pair(100000, seed)
    da, db = ctx.upload(a), ctx.upload(b)
    prev = None; out = []
    for it in range(4, 21):
        p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=it)
        rec = ctx.align_batch([da], [db], None, p, s3d.ExecOptions(force_iterations=1))
        T = s3d.api.record_transform(rec[0]).astype(np.float32)
        if prev is not None: out.append('%d:%s' % (it, '=' if np.array_equal(T, prev) else '%.1e' % np.abs(T - prev).max()))
        prev = T
    print('seed', seed, ' '.join(out), flush=True)
