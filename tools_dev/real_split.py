"""dev tool (round 6): the 96 default registrations on 192 distinct real clouds (tools_dev/real_batch.py) as 1 ... 4 concurrent
sub-batches on ONE device (s3d_align_batch_multi with the device listed several times: a context + host thread + stream
per rank) - how much of the controller's one-wave-per-pair launches and the launch gaps of a short real batch overlap."""
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
p = s3d.default_params()
ref = None
for ranks in [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4, 1, 2]:
    sw = s3d.Sweep([0] * ranks)
    a = [sw.upload(q) for q in ds]; b = [sw.upload(q) for q in dt_]
    ts = []
    for i in range(8):
        t = time.perf_counter(); rec = sw.align_batch(a, b, None, p, None); ts.append((time.perf_counter() - t) * 1e3)
    if ref is None: ref = rec.copy()
    print('ranks %d (%s): mean(last 5) %.2f ms  min %.2f  same records %s' %
          (ranks, sw.collective, np.mean(ts[3:]), np.min(ts[3:]), np.array_equal(ref, rec)), flush=True)
    sw.close()
