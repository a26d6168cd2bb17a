"""dev tool (round 5): the reference's actual call - ONE registration of two real scans with default parameters
(s3d_align_clouds(cloud1, cloud2), early exit) - REPS times, for a kernel trace.  env: REPS (20), CACHE (0), PAIR (0)"""
import os, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')
fc = [np.load(os.path.join(G, 'cloud%d.npz' % i))['xyzi'].astype(np.float32) for i in range(1, 5)]
ctx = s3d.Context(0)
dev = [ctx.upload(c) for c in fc]
k = int(os.environ.get('PAIR', '0'))
o = s3d.ExecOptions(cache_prepass=int(os.environ.get('CACHE', '0')), debug_flags=int(os.environ.get('FLAGS', '0'), 0))
p = s3d.default_params()
for _ in range(3): st = ctx.align_clouds(dev[k], dev[k + 1], np.eye(4), p, o)
ts = []
for _ in range(int(os.environ.get('REPS', '20'))):
    t = time.perf_counter(); st = ctx.align_clouds(dev[k], dev[k + 1], np.eye(4), p, o); ts.append((time.perf_counter() - t) * 1e3)
print('status %d iterations %d  median %.3f ms  min %.3f ms' % (st[0], st[2]['iterations'], np.median(ts), np.min(ts)))
o1 = s3d.ExecOptions(cache_prepass=0, profile=1)
ctx.align_batch([dev[k]], [dev[k + 1]], None, p, o1)
pr = ctx.last_profile()
print({q: round(v, 3) for q, v in pr.items() if q.endswith('_ms') and q != 'nn_launch_ms'}, [round(x, 3) for x in pr['nn_launch_ms'][:8]])
