"""Which of batch size / second context / concurrency changes results? (follow-up of two_ctx.py)"""
import sys
import threading
from multiprocessing.pool import ThreadPool

import numpy as np

sys.path.insert(0, ".")
import slam3d_amd as s3d  # noqa: E402

NP = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pairs = ThreadPool(32).map(lambda i: s3d.make_pair(100000, i), range(NP))
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=20)
o = s3d.ExecOptions(force_iterations=1, profile=0)
c0, c1 = s3d.Context(0), s3d.Context(0)
H = NP // 2
a0 = [c0.upload(q[0]) for q in pairs]; b0 = [c0.upload(q[1]) for q in pairs]
a1 = [c1.upload(q[0]) for q in pairs[H:]]; b1 = [c1.upload(q[1]) for q in pairs[H:]]
full = c0.align_batch(a0, b0, None, p, o)
print("repeat          :", np.array_equal(full, c0.align_batch(a0, b0, None, p, o)))
seq = np.vstack([c0.align_batch(a0[:H], b0[:H], None, p, o), c0.align_batch(a0[H:], b0[H:], None, p, o)])
print("halves, 1 ctx   :", np.array_equal(full, seq), np.abs(full - seq).max(0)[[0, 9, 12, 13, 14, 15]])
seq2 = np.vstack([c0.align_batch(a0[:H], b0[:H], None, p, o), c1.align_batch(a1, b1, None, p, o)])
print("halves, 2 ctx   :", np.array_equal(full, seq2))
out = [None, None]
def w0(): out[0] = c0.align_batch(a0[:H], b0[:H], None, p, o)
def w1(): out[1] = c1.align_batch(a1, b1, None, p, o)
for rep in range(3):
    t0, t1 = threading.Thread(target=w0), threading.Thread(target=w1)
    t0.start(); t1.start(); t0.join(); t1.join()
    con = np.vstack(out)
    bad = np.nonzero(np.any(full != con, axis=1))[0]
    print("concurrent      :", np.array_equal(full, con), "differing pairs", bad[:10], "max diff", np.abs(full - con).max())
