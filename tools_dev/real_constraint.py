"""dev tool (round 6): s3d_create_constraint_clouds(cloud1, cloud4, odometry 2 m, loop = true: coarse 0.5 m + fine) on the
reference's scans, median of 30 calls, for a list of debug_flags (0x400 = everything on one stream / context)."""
import os, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')
fc = [np.ascontiguousarray(np.load(os.path.join(G, 'cloud%d.npz' % i))['xyzi'].astype(np.float32)[:, :3]) for i in (1, 4)]
ctx = s3d.Context(0)
a, b = ctx.upload(fc[0]), ctx.upload(fc[1])
odo = np.eye(4); odo[0, 3] = 2.0
fine = s3d.default_params(); coarse = s3d.default_params(point_cloud_density=0.5)
for rep in range(2):
    for fl in [int(x, 0) for x in sys.argv[1:]] or [0, 0x400]:
        o = s3d.ExecOptions(debug_flags=fl)
        ts = []
        for _ in range(34):
            t = time.perf_counter(); r = ctx.create_constraint_clouds(a, np.eye(4), b, np.eye(4), odo, True, fine, coarse, 1.0, o); ts.append((time.perf_counter() - t) * 1e3)
        print('flags %#x: status %d iterations %d  median %.3f ms  min %.3f ms  t = %s' % (fl, r[0], r[3]['iterations'], np.median(ts[4:]), np.min(ts[4:]), np.round(r[1][:3, 3], 5)), flush=True)
