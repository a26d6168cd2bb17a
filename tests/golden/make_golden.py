#!/usr/bin/env python3
"""Generates tests/golden/oracle_golden.json from the CPU oracle on the reference's fixture clouds.

The reference holds NO expected values for this path (parity unpinned) and cannot be
built here (PCL/Eigen/Boost absent), so these vectors pin the ORACLE (oracle/s3d_oracle.c) — they
are what `-m "not gpu"` tests replay on any host, and what the GPU parity tests compare with.
Inputs: tests/golden/cloud{1..4}.npz = /root/reference/test/cloud{1..4}.bin (float32 x,y,z,intensity).

Run from the repo root:  python tests/golden/make_golden.py
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
out = {"voxel": {}, "nn": {}, "align": []}

for leaf in (0.1, 0.2, 0.5, 1.0):
    v, info = oracle.voxel_downsample(clouds[0], leaf)
    out["voxel"]["%.1f" % leaf] = {
        "n": int(len(v)), "div_b": list(info.div_b), "min_b": list(info.min_b),
        "first8": v[:8].astype(float).tolist(), "last8": v[-8:].astype(float).tolist(),
        "sha256": hashlib.sha256(v.tobytes()).hexdigest()}

v1, _ = oracle.voxel_downsample(clouds[0], 0.2)
v2, _ = oracle.voxel_downsample(clouds[1], 0.2)
idx, d2 = oracle.nn_search(v1, v2)
sel = np.linspace(0, len(v2) - 1, 256).astype(int)
out["nn"] = {"queries": sel.tolist(), "idx": idx[sel].tolist(), "d2": d2[sel].astype(float).tolist(),
             "sha256_idx": hashlib.sha256(idx.tobytes()).hexdigest()}

guess14 = np.eye(4)
guess14[0, 3] = 2.0
cases = [(0, 1, None), (1, 2, None), (2, 3, None), (0, 3, guess14), (0, 3, None)]
for alg, name in ((oracle.ALG_GICP, "gicp"), (oracle.ALG_ICP, "icp")):
    for mode in ((0, 2) if alg == oracle.ALG_GICP else (0,)):
        oracle.set_eval_precision(mode)
        for a, b, g in cases:
            p = oracle.default_params(registration_algorithm=alg)
            st, T, info = oracle.align(clouds[a], clouds[b], np.eye(4) if g is None else g, p)
            out["align"].append({"algorithm": name, "eval_precision": mode, "source": a + 1, "target": b + 1,
                                 "guess": (np.eye(4) if g is None else g).tolist(), "status": int(st),
                                 "T": T.tolist(), "info": info})
oracle.set_eval_precision(0)
with open(os.path.join(G, "oracle_golden.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote", os.path.join(G, "oracle_golden.json"))
