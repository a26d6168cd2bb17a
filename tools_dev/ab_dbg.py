"""dev tool: per-pass NN launch times of the default batch for several S3D_DBG_NN settings (A/B bits of the NN kernel)."""
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
for flags in os.environ.get('FLAGS', '0,256,512,1024').split(','):
    if flags == '0': os.environ.pop('S3D_DBG_NN', None)
    else: os.environ['S3D_DBG_NN'] = flags
    for _ in range(2): ctx.align_batch(a, b, None, p, o)
    rec = ctx.align_batch(a, b, None, p, o)
    pr = ctx.last_profile()
    if ref is None: ref = rec
    print('flags', flags, 'nn %.2f ms' % pr['nn_ms'], 'passes', ' '.join('%.2f' % x for x in pr['nn_launch_ms'][:6]), 'identical', bool(np.array_equal(rec, ref)), flush=True)
