"""dev tool (round 4): the 256-pair step as ONE call against two / four calls of half / quarter batches issued from host
threads on contexts (streams) of their own - does the latency-bound part of the settled iterations (record test / touch /
search / controller: ~0.1 of 0.39 ms per iteration, chip nearly idle) overlap another sub-batch's accumulate kernel?
usage: python tools_dev/two_ctx.py     env: NPAIRS (256), POINTS (100000), ITERS (20)"""
import os, sys, time, threading, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NP = int(os.environ.get('NPAIRS', '256')); PTS = int(os.environ.get('POINTS', '100000')); IT = int(os.environ.get('ITERS', '20'))
with ThreadPool(16) as pool: pairs = pool.map(lambda i: s3d.make_pair(PTS, i), range(NP))
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=IT)
o = s3d.ExecOptions(force_iterations=1)


def run(nctx, reps=6):
    ctxs = [s3d.Context(0) for _ in range(nctx)]
    per = NP // nctx
    a = [[c.upload(q[0]) for q in pairs[k * per:(k + 1) * per]] for k, c in enumerate(ctxs)]
    b = [[c.upload(q[1]) for q in pairs[k * per:(k + 1) * per]] for k, c in enumerate(ctxs)]
    outs = [None] * nctx

    def work(k):
        outs[k] = ctxs[k].align_batch(a[k], b[k], None, p, o)
    ts = []
    for r in range(reps):
        th = [threading.Thread(target=work, args=(k,)) for k in range(nctx)]
        t = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        ts.append((time.perf_counter() - t) * 1e3)
    h = float(sum(np.abs(x[:, :12]).sum() for x in outs))
    for c in ctxs: c.close()
    return min(ts[1:]), float(np.mean(ts[1:])), h


for rep in range(2):
    for n in (1, 2, 4, 1, 2):
        mn, mean, h = run(n)
        print('%d context(s): min %.2f mean %.2f ms   hash %.17g' % (n, mn, mean, h), flush=True)
