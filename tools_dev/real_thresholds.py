"""dev tool (round 5): are the hard-coded choices (grid budget 2 cells per point, record-wise passes from 65 536 records, the
sort form, the cooperative far k-NN kernel for small batches only) right on the REFERENCE's scans too?  The 96
registrations on 192 distinct real clouds (tools_dev/real_batch.py) for each alternative, in one process."""
import os, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
A = s3d.api
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
variants = [('default', dict()), ('grid budget 1', dict(grid_cells_per_point=1)), ('grid budget 3', dict(grid_cells_per_point=3)),
            ('grid budget 4', dict(grid_cells_per_point=4)), ('record-wise passes forced', dict(debug_flags=A.DBG_NN_FORCE_SETTLED)),
            ('no record-wise passes', dict(debug_flags=A.DBG_NN_NO_SETTLED)), ('sort: classic', dict(debug_flags=A.DBG_SORT_CLASSIC)),
            ('sort: one sweep', dict(debug_flags=A.DBG_SORT_ONESWEEP)), ('far k-NN: cooperative', dict(debug_flags=A.DBG_KNN_FORCE_FAR_COOP)),
            ('far k-NN: per lane', dict(debug_flags=A.DBG_KNN_NO_FAR_COOP)), ('two-sort pre-pass', dict(debug_flags=A.DBG_NO_FUSED_PREPASS))]
for rep in range(2):
    for name, kw in variants:
        o = s3d.ExecOptions(profile=1, **kw)
        ts = []
        for _ in range(4):
            t = time.perf_counter(); rec = ctx.align_batch(src, tgt, None, p, o); ts.append((time.perf_counter() - t) * 1e3)
        pr = ctx.last_profile()
        if rep == 1:
            print('%-28s call %.2f ms  pre-pass %.2f normals %.2f icp %.2f (nn %.2f) fit %.2f  ok %d' % (
                name, np.mean(ts[1:]), pr['voxel_ms'] + pr['grid_ms'], pr['normals_ms'], pr['icp_ms'], pr['nn_ms'], pr['fitness_ms'],
                int((rec[:, 15] == 0).sum())), flush=True)
