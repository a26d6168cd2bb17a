"""dev tool (round 5): 96 registrations of the reference's scans on 192 DISTINCT clouds (every copy moved by its own small
rigid motion), default parameters, early exit: stage split for a list of debug_flags.  usage: real_batch.py [flags ...]"""
import os, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')
fc = [np.load(os.path.join(G, 'cloud%d.npz' % i))['xyzi'].astype(np.float32) for i in range(1, 5)]
rng = np.random.default_rng(5)
ds, dt_ = [], []
for k in range(32):
    for a, b in ((0, 1), (1, 2), (2, 3)):
        for which, lst in ((a, ds), (b, dt_)):
            ang = rng.normal(0, 0.01, 3); tr = rng.normal(0, 0.05, 3)
            cz, sz = np.cos(ang[2]), np.sin(ang[2])
            R = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[1, 0, ang[1]], [0, 1, -ang[0]], [-ang[1], ang[0], 1]])
            lst.append(np.ascontiguousarray((fc[which][:, :3].astype(np.float64) @ R.T + tr).astype(np.float32)))
ctx = s3d.Context(0)
dev = ctx.upload_many(ds + dt_)
src, tgt = dev[:96], dev[96:]
p = s3d.default_params()
for fl in [int(x, 0) for x in sys.argv[1:]] or [0]:
    o = s3d.ExecOptions(profile=1, debug_flags=fl)
    for _ in range(3):
        t = time.perf_counter(); rec = ctx.align_batch(src, tgt, None, p, o); ms = (time.perf_counter() - t) * 1e3
    pr = ctx.last_profile()
    print('flags %#x: call %.2f ms  voxel %.2f grid %.2f normals %.2f icp %.2f (nn %.2f) fit %.2f  ok %d  hash %.15g' % (
        fl, ms, pr['voxel_ms'], pr['grid_ms'], pr['normals_ms'], pr['icp_ms'], pr['nn_ms'], pr['fitness_ms'],
        int((rec[:, 15] == 0).sum()), float(np.abs(rec[:, :12]).sum())), flush=True)
    print('   nn ms:', ' '.join('%.3f' % x for x in pr['nn_launch_ms'][:12]))
