import os, sys, numpy as np
sys.path.insert(0, '.')
import slam3d_amd as s3d
fc = [np.load('tests/golden/cloud%d.npz' % i)['xyzi'].astype(np.float32) for i in range(1, 5)]
ctx = s3d.Context(0)
dev = [ctx.upload(c) for c in fc]
p = s3d.default_params()
for fl in (0,):
    o = s3d.ExecOptions(profile=2, debug_flags=fl)
    ctx.align_batch([dev[0]], [dev[1]], None, p, o); pr = ctx.last_profile()
    print('searched', pr['nn_searched'][:8]); print('unseeded', pr['nn_unseeded'][:8]); print('ms', [round(x,3) for x in pr['nn_launch_ms'][:8]])
