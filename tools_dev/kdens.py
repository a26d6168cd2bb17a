"""dev tool: GICP vs the smooth-objective oracle on fixture pair 1->2 for several (k, density)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import slam3d_amd as s3d, oracle
from conftest import transform_delta, GOLDEN
c = [np.load(os.path.join(GOLDEN, "cloud%d.npz" % i))["xyzi"].astype(np.float32) for i in (1, 2, 3)]
ctx = s3d.Context(0)
for a, b in ((0, 1), (1, 2)):
    for k in (20, 30, 40):
        for dens in (0.2, 0.3):
            oracle.set_eval_precision(2)
            so, To, io = oracle.align(c[a], c[b], np.eye(4), oracle.default_params(correspondence_randomness=k, point_cloud_density=dens))
            oracle.set_eval_precision(0)
            sg, Tg, ig = ctx.align(c[a], c[b], np.eye(4), s3d.default_params(correspondence_randomness=k, point_cloud_density=dens))
            v, _ = oracle.voxel_downsample(c[a], dens)
            ng = ctx.knn_normals(v, k).astype(np.float64); _, nr = oracle.gicp_covariances(v, k)
            dots = np.abs((ng * nr).sum(1))
            print("pair", a, b, "k", k, "dens", dens, "it", ig["iterations"], io["iterations"], "dt %.2e dr %.2e" % transform_delta(To, Tg),
                  "normals off: %d of %d (worst dot %.6f)" % ((dots < 1 - 1e-6).sum(), len(v), dots.min()), flush=True)
