#!/usr/bin/env python3
"""bench_map.py — secondary benchmark: PointCloudSensor::buildMap on one MI355X.

Not the contract bench (that is bench.py / registrations per second).  One step = buildMap over `--scans`
device-resident synthetic scans of `--points` points: accumulate -> radius outlier removal -> voxel filter at
map resolution, everything in HBM.  Prints one JSON line: input points/s, per-stage HIP-event times and the
CPU oracle on a bounded sample of the same scans.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=96)
    ap.add_argument("--points", type=int, default=100000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--radius", type=float, default=0.2)       # PointCloudSensor.cpp:181
    ap.add_argument("--neighbors", type=int, default=3)        # :182
    ap.add_argument("--resolution", type=float, default=0.1)   # :180
    ap.add_argument("--cpu-scans", type=int, default=8)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    import slam3d_amd as s3d
    from multiprocessing.pool import ThreadPool

    def scan(i):
        rng = np.random.default_rng(7000 + i)
        n_air = args.points // 50
        air = rng.uniform([-40, -8, 5], [40, 8, 30], size=(n_air, 3)).astype(np.float32)
        return np.concatenate([s3d.make_scene_cloud(args.points - n_air, 5000 + i), air])

    with ThreadPool(min(16, os.cpu_count() or 1)) as pool:
        clouds = pool.map(scan, range(args.scans))
    poses = []
    for i in range(args.scans):
        T = np.eye(4)
        T[:3, 3] = [0.8 * i, 0.3 * (i % 7), 0.0]
        poses.append(T)
    ctx = s3d.Context(0)
    dev = [ctx.upload(c) for c in clouds]
    for _ in range(args.warmup):
        ctx.build_map(dev, poses, args.radius, args.neighbors, args.resolution).release()
    t0 = time.perf_counter()
    profs = []
    for _ in range(args.steps):
        m = ctx.build_map(dev, poses, args.radius, args.neighbors, args.resolution)
        profs.append(ctx.last_map_profile())
        m.release()
    dt = (time.perf_counter() - t0) / args.steps
    n_in = args.scans * args.points
    prof = {k: round(float(np.median([p[k] for p in profs])), 3) for k in profs[0]}
    cpu = None
    if not args.no_cpu:
        import oracle
        k = min(args.cpu_scans, args.scans)
        tc = time.perf_counter()
        mo = oracle.build_map(clouds[:k], poses[:k], args.radius, args.neighbors, args.resolution)
        tcpu = time.perf_counter() - tc
        cpu = {"value": round(k * args.points / tcpu, 1), "unit": "input points/s", "cores": 1, "kind": "port",
               "sample": "first %d scans of the same set (oracle/s3d_oracle.c s3o_build_map), %.1f s" % (k, tcpu),
               "n_map": int(len(mo))}
    print(json.dumps({"metric": "buildMap input points/sec", "value": round(n_in / dt, 1), "unit": "points/s",
                      "ms_per_step": round(dt * 1e3, 3), "steps": args.steps, "warmup": args.warmup,
                      "config": {"workload": "%d scans x %dk points, outlier radius %.2f m / %d neighbours, map "
                                             "resolution %.2f m" % (args.scans, args.points // 1000, args.radius,
                                                                    args.neighbors, args.resolution)},
                      "stage_ms": prof, "cpu_baseline": cpu}))


if __name__ == "__main__":
    main()
