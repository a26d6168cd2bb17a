"""PCIe-inclusive rates: the same work with the clouds handed over as HOST buffers.
  single pair : s3d_align (host pointers, upload inside the call) vs s3d_align_clouds (device handles)
  batch       : upload of all 512 clouds + s3d_align_batch + free, per step, vs s3d_align_batch alone"""
import sys
import time
from multiprocessing.pool import ThreadPool

import numpy as np

sys.path.insert(0, ".")
import slam3d_amd as s3d  # noqa: E402

NP = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pairs = ThreadPool(32).map(lambda i: s3d.make_pair(100000, i), range(NP))
ctx = s3d.Context(0)
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=20)
o = s3d.ExecOptions(force_iterations=1, profile=0)
src, tgt = pairs[0][0], pairs[0][1]
a, b = ctx.upload(src), ctx.upload(tgt)
for f, name in ((lambda: ctx.align(src, tgt, np.eye(4), p, o), "host buffers (s3d_align)"),
                (lambda: ctx.align_clouds(a, b, np.eye(4), p, o), "device handles (s3d_align_clouds)")):
    f(); f()
    t = time.perf_counter()
    for _ in range(20):
        f()
    print("single pair, %-34s %.3f ms" % (name + ":", (time.perf_counter() - t) / 20 * 1e3))
A = [ctx.upload(q[0]) for q in pairs]; B = [ctx.upload(q[1]) for q in pairs]
ctx.align_batch(A, B, None, p, o)
t = time.perf_counter()
for _ in range(3):
    ctx.align_batch(A, B, None, p, o)
dev = (time.perf_counter() - t) / 3
def step():
    A2 = [ctx.upload(q[0]) for q in pairs]; B2 = [ctx.upload(q[1]) for q in pairs]
    r = ctx.align_batch(A2, B2, None, p, o)
    for c in A2 + B2:
        c.release()
    return r
step()
t = time.perf_counter()
for _ in range(3):
    step()
host = (time.perf_counter() - t) / 3
mb = sum(q[0].nbytes + q[1].nbytes for q in pairs) / 1e6
print("batch of %d: device-resident %.1f ms (%.0f reg/s); upload of %.0f MB + batch + free %.1f ms (%.0f reg/s)"
      % (NP, dev * 1e3, NP / dev, mb, host * 1e3, NP / host))
