"""dev tool: one pair next to a 128-pair batch, with and without a CU reservation (s3d_context_create_cu_mask)."""
import sys, threading, time, numpy as np
sys.path.insert(0, '.')
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
with ThreadPool(16) as pool: pairs = pool.map(lambda i: s3d.make_pair(100000, i), range(128))
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=20)
opts = s3d.ExecOptions(force_iterations=1)
ALL = [0xFFFFFFFF] * 8
def run(label, batch_mask, fast_mask):
    big = s3d.Context(0, cu_mask=batch_mask) if batch_mask else s3d.Context(0)
    fast = s3d.Context(0, cu_mask=fast_mask) if fast_mask else s3d.Context(0)
    src = [big.upload(q[0]) for q in pairs]; tgt = [big.upload(q[1]) for q in pairs]
    for _ in range(2): big.align_batch(src, tgt, None, p, opts)
    t = time.perf_counter()
    for _ in range(4): big.align_batch(src, tgt, None, p, opts)
    batch_alone = (time.perf_counter() - t) / 4 * 1e3
    a, b = fast.upload(pairs[5][0]), fast.upload(pairs[5][1])
    for _ in range(3): alone = fast.align_batch([a], [b], None, p, opts)
    t = time.perf_counter()
    for _ in range(10): fast.align_batch([a], [b], None, p, opts)
    idle = (time.perf_counter() - t) * 100
    stop = threading.Event(); done = []
    def sweep():
        while not stop.is_set(): done.append(time.perf_counter()); big.align_batch(src, tgt, None, p, opts)
    th = threading.Thread(target=sweep); th.start()
    time.sleep(0.05); lat = []
    for _ in range(20):
        t = time.perf_counter(); rec = fast.align_batch([a], [b], None, p, opts); lat.append((time.perf_counter() - t) * 1e3)
        assert np.array_equal(rec, alone); time.sleep(0.003)
    stop.set(); th.join()
    d = np.diff(done)
    print("%-28s batch alone %.2f ms, under load %.2f ms; one pair idle %.2f ms, next to the batch %.2f ms (max %.2f)" % (
        label, batch_alone, np.median(d) * 1e3 if len(d) else -1, idle, np.median(lat), max(lat)), flush=True)
    big.close(); fast.close()
run("no masks", None, None)
run("batch 224 CUs, pair 32", [0] + [0xFFFFFFFF] * 7, [0xFFFFFFFF] + [0] * 7)
run("batch 192 CUs, pair 64", [0, 0] + [0xFFFFFFFF] * 6, [0xFFFFFFFF] * 2 + [0] * 6)
run("batch all, pair 32", None, [0xFFFFFFFF] + [0] * 7)
run("striped: 4 CUs per word", [0xFFFFFFF0] * 8, [0xF] * 8)
