"""dev tool (round 5): are the k-NN pre-pass (VALU-bound) and the ICP loop (memory / latency-bound) complementary?
Context 1 loops a batch whose pre-pass comes from the cache (ICP loop + fitness only), context 2 loops the same batch with ONE
outer iteration and no cache (pre-pass + k-NN + one iteration): each alone, then side by side from two host threads."""
import os, sys, time, threading, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NP = int(os.environ.get('NPAIRS', '128')); PTS = 100000; N = 8
with ThreadPool(16) as pool: pairs = pool.map(lambda i: s3d.make_pair(PTS, i), range(NP))
c = [s3d.Context(0), s3d.Context(0)]
ab = [([x.upload(q[0]) for q in pairs], [x.upload(q[1]) for q in pairs]) for x in c]
p_icp = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=20)
p_pre = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=1)
o_icp = s3d.ExecOptions(force_iterations=1, cache_prepass=1)
o_pre = s3d.ExecOptions(force_iterations=1, cache_prepass=0)
def icp(n):
    for i in range(n): c[0].align_batch(ab[0][0], ab[0][1], None, p_icp, o_icp)
def pre(n):
    for i in range(n): c[1].align_batch(ab[1][0], ab[1][1], None, p_pre, o_pre)
icp(3); pre(3)
for rep in range(3):
    t = time.perf_counter(); icp(N); t_icp = (time.perf_counter() - t) * 1e3 / N
    t = time.perf_counter(); pre(N); t_pre = (time.perf_counter() - t) * 1e3 / N
    t = time.perf_counter()
    th = [threading.Thread(target=icp, args=(N,)), threading.Thread(target=pre, args=(N,))]
    for x in th: x.start()
    for x in th: x.join()
    both = (time.perf_counter() - t) * 1e3 / N
    print('%d pairs: ICP-only step %.2f ms, pre-pass + k-NN step %.2f ms, one after the other %.2f, side by side %.2f ms (%.0f %%)' %
          (NP, t_icp, t_pre, t_icp + t_pre, both, 100 * (1 - both / (t_icp + t_pre))), flush=True)
