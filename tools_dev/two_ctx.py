"""Does splitting the default batch over two contexts (two streams, two host threads) overlap anything?
256 pairs on one context vs 2 x 128 pairs on two contexts driven concurrently."""
import sys
import threading
import time
from multiprocessing.pool import ThreadPool

import numpy as np

sys.path.insert(0, ".")
import slam3d_amd as s3d  # noqa: E402

NP = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pairs = ThreadPool(32).map(lambda i: s3d.make_pair(100000, i), range(NP))
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=20)
o = s3d.ExecOptions(force_iterations=1, profile=0)


def run(nctx, reps=6):
    ctxs = [s3d.Context(0) for _ in range(nctx)]
    share = NP // nctx
    a = [[c.upload(pairs[k * share + i][0]) for i in range(share)] for k, c in enumerate(ctxs)]
    b = [[c.upload(pairs[k * share + i][1]) for i in range(share)] for k, c in enumerate(ctxs)]
    out = [None] * nctx

    def work(k):
        out[k] = ctxs[k].align_batch(a[k], b[k], None, p, o)

    def step():
        th = [threading.Thread(target=work, args=(k,)) for k in range(nctx)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    step(); step()
    t = time.perf_counter()
    for _ in range(reps):
        step()
    dt = (time.perf_counter() - t) / reps
    rec = np.concatenate([np.asarray(r) for r in out])
    for c in ctxs:
        c.close()
    return dt, rec


base, r1 = run(1)
print("1 context : %.2f ms per step (%.0f reg/s)" % (base * 1e3, NP / base))
for n in (2, 4):
    dt, rn = run(n)
    print("%d contexts: %.2f ms per step (%.0f reg/s), results identical: %s" % (n, dt * 1e3, NP / dt, np.array_equal(r1, rn)))
