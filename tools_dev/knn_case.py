"""dev tool (round 6): ONE case of tools_dev/knn_neutral_fuzz.py (CASE, SEED0) in detail: the records of the default (fused) and the
two-sort layout, their difference, and both against the oracle's smooth-objective mode."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
import oracle
A = s3d.api
rng = np.random.default_rng(int(os.environ.get('SEED0', '0')))
ctx = s3d.Context(0)
def cloud(kind, n):
    if kind == 0: return rng.uniform(-15, 15, (n, 3)).astype(np.float32)
    if kind == 1:
        p = rng.uniform(-40, 40, (n, 3)).astype(np.float32); p[: 2 * n // 3, 2] = rng.normal(0, 0.03, 2 * n // 3); return p
    if kind == 2:
        r = np.abs(rng.normal(0, 12, n)) + 0.5; a = rng.uniform(0, 2 * np.pi, n)
        return np.stack([r * np.cos(a), r * np.sin(a), rng.normal(0, 0.05, n) + 0.02 * r], 1).astype(np.float32)
    if kind == 3:
        c = rng.uniform(-60, 60, (6, 3)); return (c[rng.integers(0, 6, n)] + rng.normal(0, 0.4, (n, 3))).astype(np.float32)
    return s3d.make_scene_cloud(n, int(rng.integers(1 << 30))).astype(np.float32)
want = int(os.environ.get('CASE', '52'))
for case in range(want + 1):
    kind = int(rng.integers(5)); n = int(rng.integers(800, 60000))
    a = cloud(kind, n)
    T = np.eye(4); T[:3, 3] = rng.uniform(-0.2, 0.2, 3)
    b = (a + rng.normal(0, 0.004, a.shape).astype(np.float32) - T[:3, 3].astype(np.float32)).astype(np.float32)
    dens = float(rng.choice([0.05, 0.2, 0.5])); alg = s3d.ALG_GICP if rng.random() < 0.75 else s3d.ALG_ICP
    k = int(rng.choice([20, 20, 12, 30]))
print('case', want, 'kind', kind, 'n', n, 'density', dens, 'k', k, 'alg', alg)
p = s3d.default_params(registration_algorithm=alg, point_cloud_density=dens, maximum_iterations=6, correspondence_randomness=k)
da, db = ctx.upload(a), ctx.upload(b)
recs = {}
for name, fl in (('fused', 0), ('two-sort', A.DBG_NO_FUSED_PREPASS)):
    r = ctx.align_batch([da], [db], None, p, s3d.ExecOptions(debug_flags=fl))[0]
    recs[name] = r
    print(name, 'status', r[15], 'iterations', r[13], 'corr', r[14], 'fitness %.9g' % r[12], 't', r[9:12])
d = np.abs(recs['fused'][:12] - recs['two-sort'][:12])
print('max |diff| rotation part %.3e translation part %.3e' % (d[:9].max(), d[9:].max()))
po = oracle.default_params(registration_algorithm=alg, point_cloud_density=dens, maximum_iterations=6, correspondence_randomness=k)
oracle.set_eval_precision(2 if alg == s3d.ALG_GICP else 0)
st, To, info = oracle.align(a, b, np.eye(4), po)
oracle.set_eval_precision(0)
print('oracle status', st, 'iterations', info['iterations'], 't', To[:3, 3])
for name in recs:
    Tg = s3d.api.record_transform(recs[name][None, :][0])
    print(name, 'vs oracle |dt| %.3e' % np.linalg.norm(Tg[:3, 3] - To[:3, 3]))
