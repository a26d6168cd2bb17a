"""dev tool: where does a single-pair registration spend its wall time?  raw C call vs Python wrapper vs GPU events."""
import ctypes as C, os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from slam3d_amd.api import EdgeRecord, _dp
a, b, _ = s3d.make_pair(100000, 0)
ctx = s3d.Context(0)
ca, cb = ctx.upload(a), ctx.upload(b)
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=20)
for prof in (0, 1):
    o = s3d.ExecOptions(force_iterations=1, profile=prof)
    g = np.ascontiguousarray(np.eye(4).reshape(1, 16))
    rec = np.zeros((1, 16))
    sa = (C.c_void_p * 1)(ca.handle); ta = (C.c_void_p * 1)(cb.handle)
    call = lambda: ctx._L.s3d_align_batch(ctx._h, 1, sa, ta, _dp(g), C.byref(p), C.byref(o), rec.ctypes.data_as(C.POINTER(EdgeRecord)), None)
    for _ in range(20): call()
    t = time.perf_counter()
    for _ in range(200): call()
    raw = (time.perf_counter() - t) / 200 * 1e3
    t = time.perf_counter()
    for _ in range(200): ctx.align_batch([ca], [cb], None, p, o)
    wrapped = (time.perf_counter() - t) / 200 * 1e3
    pr = ctx.last_profile()
    print("profile=%d: raw C call %.3f ms, python wrapper %.3f ms, GPU events total %.3f ms (voxel %.3f grid %.3f normals %.3f icp %.3f fitness %.3f)" %
          (prof, raw, wrapped, pr['total_ms'], pr['voxel_ms'], pr['grid_ms'], pr['normals_ms'], pr['icp_ms'], pr['fitness_ms']))
