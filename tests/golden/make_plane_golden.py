#!/usr/bin/env python3
"""Generates tests/golden/plane_golden.json: the CPU oracle's fillGroundPlane results (RANSAC plane + ring points)
on the reference's fixture clouds (tests/golden/cloud{1..4}.npz = /root/reference/test/cloud{1..4}.bin), raw and
after the 0.2 m voxel filter.

The reference has no test of fillGroundPlane (PointCloudSensor.cpp:362-388), so these vectors pin the oracle,
not PCL.  Run from the repo root:  python tests/golden/make_plane_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
out = {"cases": []}
for i in range(1, 5):
    raw = np.load(os.path.join(G, "cloud%d.npz" % i))["xyzi"]
    for name, cloud in (("raw", raw), ("voxel0.2", oracle.voxel_downsample(raw, 0.2)[0])):
        ok, co, ninl, it = oracle.fit_plane_ransac(cloud)
        filled = oracle.fill_ground_plane(cloud, 5.0, 0.1)
        ring = filled[len(cloud):]
        out["cases"].append({"cloud": i, "input": name, "n": int(len(cloud)), "found": ok,
                             "coefficients_hex": [float(c).hex() for c in co], "n_inliers": ninl, "iterations": it,
                             "radius": 5.0, "map_resolution": 0.1, "n_ring": int(len(ring)),
                             "ring_sha256": hashlib.sha256(np.ascontiguousarray(ring, np.float32).tobytes()).hexdigest(),
                             "ring_first2": ring[:2].astype(float).tolist()})
with open(os.path.join(G, "plane_golden.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote", len(out["cases"]), "cases")
