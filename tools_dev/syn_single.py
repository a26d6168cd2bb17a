"""dev tool (round 5): ONE synthetic 100 k-point pair, 20 forced GICP iterations (BASELINE configs[1]), REPS times - for a kernel trace"""
import os, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
a, b, _ = s3d.make_pair(100000, 0)
ctx = s3d.Context(0)
da, db = ctx.upload(a), ctx.upload(b)
p = s3d.default_params(point_cloud_density=0.02, maximum_iterations=20)
o = s3d.ExecOptions(force_iterations=1, check_interval=0, debug_flags=int(os.environ.get('FLAGS', '0'), 0))
for _ in range(3): ctx.align_batch([da], [db], None, p, o)
ts = []
for _ in range(int(os.environ.get('REPS', '20'))):
    t = time.perf_counter(); ctx.align_batch([da], [db], None, p, o); ts.append((time.perf_counter() - t) * 1e3)
print('median %.3f ms  min %.3f ms' % (np.median(ts), np.min(ts)))
