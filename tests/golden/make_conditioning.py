"""Generates tests/golden/conditioning_golden.json: how far the oracle's GICP result moves when PCL's Mahalanobis
matrices are perturbed by a relative 1e-15 / 1e-12 / 1e-9 (deterministic noise, three seeds), on the three
consecutive pairs of the reference's fixture scans (test/cloud1..4.bin), in the PCL-literal mode (eval_precision 0)
and in the smooth-objective mode (2).  The spread is the reproducibility band of the reference algorithm itself:
any re-implementation (another compiler, SIMD width, summation order) perturbs at the 1e-16 level.

Run from the repository root:  python tests/golden/make_conditioning.py   (CPU only, ~2 minutes on 8 cores)"""
import json
import os
import sys
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PAIRS = [(1, 2), (2, 3), (3, 4)]
EPS = [1e-15, 1e-12, 1e-9]
SEEDS = [1, 2, 3]


def _run(job):
    import oracle
    from conftest import GOLDEN
    a, b, mode, eps, seed = job
    ca = np.load(os.path.join(GOLDEN, "cloud%d.npz" % a))["xyzi"].astype(np.float32)
    cb = np.load(os.path.join(GOLDEN, "cloud%d.npz" % b))["xyzi"].astype(np.float32)
    oracle.set_eval_precision(mode)
    oracle.set_debug_perturbation(eps, seed)
    st, T, info = oracle.align(ca, cb)
    return job, st, T.tolist(), info["iterations"]


def main():
    from conftest import transform_delta
    jobs = [(a, b, mode, 0.0, 0) for a, b in PAIRS for mode in (0, 2)]
    jobs += [(a, b, mode, eps, seed) for a, b in PAIRS for mode in (0, 2) for eps in EPS for seed in SEEDS]
    with Pool(min(8, os.cpu_count() or 1)) as pool:      # processes: the perturbation setting is a global of the oracle
        res = pool.map(_run, jobs, chunksize=1)
    base = {(j[0], j[1], j[2]): np.array(T) for j, st, T, it in res if j[3] == 0.0}
    out = {"eps": EPS, "seeds": SEEDS, "pairs": []}
    for a, b in PAIRS:
        rec = {"source": a, "target": b}
        for mode, name in ((0, "pcl_literal"), (2, "smooth_objective")):
            rows = []
            for j, st, T, it in res:
                if (j[0], j[1], j[2]) == (a, b, mode) and j[3] != 0.0:
                    dt, dr = transform_delta(base[(a, b, mode)], np.array(T))
                    rows.append({"eps": j[3], "seed": j[4], "status": st, "iterations": it, "dt_m": dt, "dr_rad": dr})
            rec[name] = {"baseline_T": base[(a, b, mode)].tolist(), "runs": rows,
                         "max_dt_m": max(r["dt_m"] for r in rows), "max_dr_rad": max(r["dr_rad"] for r in rows)}
        out["pairs"].append(rec)
        print(a, b, {k: (rec[k]["max_dt_m"], rec[k]["max_dr_rad"]) for k in ("pcl_literal", "smooth_objective")})
    with open(os.path.join(ROOT, "tests", "golden", "conditioning_golden.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
