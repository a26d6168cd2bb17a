"""Deterministic synthetic scan pairs (SURVEY.md §8d "Synthetic input generator").

Scene: axis-aligned street canyon — ground plane z = -1.7, two facing walls y = +-8 (6 m high),
end walls x = +-40, plus 32 seeded boxes — sampled uniformly by area, sigma = 0.01 m noise.
Cloud B is an independent resample of the same scene moved by a ground-truth SE(3).
Input data only; no registration arithmetic lives here.
"""
import numpy as np


def _scene_faces(rng_boxes):
    faces = []  # (origin, u, v) rectangles
    faces.append((np.array([-40.0, -8.0, -1.7]), np.array([80.0, 0, 0]), np.array([0, 16.0, 0])))      # ground
    for y in (-8.0, 8.0):
        faces.append((np.array([-40.0, y, -1.7]), np.array([80.0, 0, 0]), np.array([0, 0, 6.0])))       # side walls
    for x in (-40.0, 40.0):
        faces.append((np.array([x, -8.0, -1.7]), np.array([0, 16.0, 0]), np.array([0, 0, 6.0])))        # end walls
    for _ in range(32):
        c = np.array([rng_boxes.uniform(-36, 36), rng_boxes.uniform(-6.5, 6.5), -1.7])
        sx, sy, sz = rng_boxes.uniform(0.5, 3.0), rng_boxes.uniform(0.5, 2.0), rng_boxes.uniform(0.5, 2.5)
        o = c - np.array([sx / 2, sy / 2, 0])
        ex, ey, ez = np.array([sx, 0, 0]), np.array([0, sy, 0]), np.array([0, 0, sz])
        faces += [(o, ex, ez), (o + ey, ex, ez), (o, ey, ez), (o + ex, ey, ez), (o + ez, ex, ey)]
    return faces


_SCENE_SEED = 20240501


def make_scene_cloud(n, seed):
    """n points (float32, (n,3)) sampled on the canyon scene with per-cloud seed."""
    faces = _scene_faces(np.random.default_rng(_SCENE_SEED))
    areas = np.array([np.linalg.norm(np.cross(u, v)) for _, u, v in faces])
    prob = areas / areas.sum()
    rng = np.random.default_rng(seed)
    out = np.empty((0, 3))
    while len(out) < n:
        need = n - len(out)
        m = int(need * 1.05) + 16
        f = rng.choice(len(faces), size=m, p=prob)
        a, b = rng.random(m), rng.random(m)
        O = np.stack([faces[i][0] for i in range(len(faces))])
        U = np.stack([faces[i][1] for i in range(len(faces))])
        V = np.stack([faces[i][2] for i in range(len(faces))])
        pts = O[f] + a[:, None] * U[f] + b[:, None] * V[f] + rng.normal(0.0, 0.01, (m, 3))
        pts = pts[np.linalg.norm(pts, axis=1) <= 80.0]
        out = np.concatenate([out, pts[:need]])
    return out.astype(np.float32)


def _rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def make_pair(n, pair_index=0):
    """Pair p: cloud A (seed 1000+2p), cloud B (seed 1001+2p) moved by the pair's ground truth.

    Returns (source, target, T_true): `source` is the slam3d source scan (A), `target` the slam3d
    target scan, expressed in its own frame, so that the registration result (pose of target in
    source frame, i.e. T with source ~= T * target) should equal T_true.
    """
    a = make_scene_cloud(n, 1000 + 2 * pair_index)
    b = make_scene_cloud(n, 1001 + 2 * pair_index)
    rng = np.random.default_rng(77000 + pair_index)
    t = rng.uniform(-0.5, 0.5, 3)
    rpy = rng.uniform(-0.03, 0.03, 3)
    T = np.eye(4)
    T[:3, :3] = _rpy(*rpy)
    T[:3, 3] = t
    # rigid inverse and the per-point products are written out element-wise: a BLAS matmul here is neither
    # bit-reproducible across thread counts nor — observed with 16 generator threads on a 256-core host and
    # OpenBLAS 0.3.29 (MAX_THREADS=64) — always correct
    R, t3 = T[:3, :3], T[:3, 3]
    Ri = R.T.copy()
    ti = -(Ri[:, 0] * t3[0] + Ri[:, 1] * t3[1] + Ri[:, 2] * t3[2])
    x, y, z = (b[:, k].astype(np.float64) for k in range(3))
    b_local = np.stack([(x * Ri[r, 0] + y * Ri[r, 1]) + (z * Ri[r, 2] + ti[r]) for r in range(3)], axis=1)
    return a, b_local.astype(np.float32), T
