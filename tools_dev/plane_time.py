"""fillGroundPlane timing: RANSAC plane fit (s3d_fit_plane) on a synthetic multi-scan map, GPU vs the CPU oracle."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import slam3d_amd  # noqa: E402
import oracle  # noqa: E402

ctx = slam3d_amd.Context(0)
for n_scans in (1, 4, 16):
    c = np.vstack([slam3d_amd.make_scene_cloud(1000000, s) + np.float32([40 * s, 0, 0]) for s in range(n_scans)])
    ctx.fit_plane(c[:1000])
    t = time.perf_counter()
    fit = ctx.fit_plane(c)
    dt = time.perf_counter() - t
    line = "%2d M points: %7.1f ms incl. upload (%d iterations, %d hypotheses scored, %d inliers)" % (
        n_scans, dt * 1e3, fit["iterations"], fit["hypotheses_scored"], fit["n_inliers"])
    if n_scans <= 4:
        t = time.perf_counter()
        ok, co, ninl, it = oracle.fit_plane_ransac(c)
        line += "; oracle %.0f ms, same result: %s" % ((time.perf_counter() - t) * 1e3,
                                                        np.array_equal(co, fit["coefficients"]) and ninl == fit["n_inliers"])
    print(line)
