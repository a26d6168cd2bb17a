#!/usr/bin/env python3
"""Generates tests/golden/ndt_golden.json: the CPU oracle's NDT results (doNDT, PointCloudSensor.cpp:84-117) on the
reference's fixture scans.  As for the other paths the reference holds no expected value, so these vectors pin the
oracle.  Run from the repo root:  python tests/golden/make_ndt_golden.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
clouds = [np.load(os.path.join(G, "cloud%d.npz" % i))["xyzi"] for i in range(1, 5)]
out = []
for a, b, gx, kw, alg in ((0, 1, 0.0, {}, "NDT"), (1, 2, 0.0, {}, "NDT"), (2, 3, 0.0, {}, "NDT"), (0, 3, 2.0, {}, "NDT"),
                          (0, 1, 0.0, {"resolution": 2.0, "step_size": 0.1, "outlier_ratio": 0.55}, "NDT"),
                          # NDT_OMP: pclomp's DIRECT7 neighbourhood
                          (0, 1, 0.0, {}, "NDT_OMP"), (1, 2, 0.0, {}, "NDT_OMP"), (2, 3, 0.0, {"resolution": 2.0}, "NDT_OMP")):
    g = np.eye(4)
    g[0, 3] = gx
    p = oracle.default_params(registration_algorithm=getattr(oracle, "ALG_" + alg), **kw)
    st, T, info = oracle.align(clouds[a], clouds[b], g, p)
    out.append({"source": a + 1, "target": b + 1, "guess_x": gx, "params": kw, "algorithm": alg, "status": int(st),
                "T": T.tolist(), "info": info})
with open(os.path.join(G, "ndt_golden.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote ndt_golden.json", [(c["source"], c["target"], c["status"], c["info"]["iterations"]) for c in out])
