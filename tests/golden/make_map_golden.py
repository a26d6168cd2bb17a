#!/usr/bin/env python3
"""Generates tests/golden/map_golden.json: the CPU oracle's patch-accumulation / outlier-removal / map results
on the reference's fixture clouds (tests/golden/cloud{1..4}.npz = /root/reference/test/cloud{1..4}.bin).

The reference's own test of this path (slam3d/sensor/pcl/PointCloudSensorTest.cpp:73-96, map_building) only
checks that buildMap() of one EMPTY cloud does not throw, so — as for the registration path — these vectors pin
the oracle, not PCL.  Run from the repo root:  python tests/golden/make_map_golden.py
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
clouds = [np.load(os.path.join(G, "cloud%d.npz" % i))["xyzi"] for i in range(1, 5)]


def pose(i):
    """deterministic vertex poses: a slow left turn, 1.5 m per scan"""
    th = 0.05 * i
    T = np.eye(4)
    T[:3, :3] = [[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]]
    T[:3, 3] = [1.5 * i, 0.1 * i * i, 0.02 * i]
    return T


def digest(a):
    return {"n": int(len(a)), "sha256": hashlib.sha256(np.ascontiguousarray(a, np.float32).tobytes()).hexdigest(),
            "first4": a[:4].astype(float).tolist(), "last4": a[-4:].astype(float).tolist()}


poses = [pose(i) for i in range(4)]
out = {"poses": [p.tolist() for p in poses]}
acc = oracle.accumulate_clouds(clouds, poses)
out["accumulate"] = digest(acc)
out["combined_frame1"] = digest(oracle.accumulate_clouds(clouds, poses, poses[1]))
out["remove_outliers"] = {}
for r, k in ((0.2, 3), (0.1, 2), (0.5, 20)):
    out["remove_outliers"]["%.1f/%d" % (r, k)] = digest(oracle.remove_outliers(acc, r, k))
out["build_map"] = {}
for r, k, res in ((0.2, 3, 0.1), (0.3, 5, 0.25)):
    out["build_map"]["%.1f/%d/%.2f" % (r, k, res)] = digest(oracle.build_map(clouds, poses, r, k, res))
with open(os.path.join(G, "map_golden.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote map_golden.json", {k: (v["n"] if "n" in v else {a: b["n"] for a, b in v.items()}) for k, v in out.items() if k != "poses"})
