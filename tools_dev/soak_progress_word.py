import os, sys, time, numpy as np
sys.path.insert(0, '.')
import slam3d_amd as s3d
fc = [np.load('tests/golden/cloud%d.npz' % i)['xyzi'].astype(np.float32) for i in range(1, 5)]
ctx = s3d.Context(0)
dev = [ctx.upload(c) for c in fc]
p = s3d.default_params()
ref = {}
for k in range(3):
    ref[k] = ctx.align_clouds(dev[k], dev[k + 1], np.eye(4), p, s3d.ExecOptions(check_interval=4))
t = time.perf_counter(); bad = 0
for i in range(3000):
    k = i % 3
    st = ctx.align_clouds(dev[k], dev[k + 1], np.eye(4), p, s3d.ExecOptions(cache_prepass=i & 1))
    if not (st[0] == ref[k][0] and np.array_equal(st[1], ref[k][1]) and st[2] == ref[k][2]): bad += 1
print('3000 registrations through the progress word: %d differ from the polled result, %.3f ms per call' % (bad, (time.perf_counter() - t) / 3))
# a batch with early exit, pairs converging at different iterations
import itertools
S = [dev[i] for i, j in itertools.permutations(range(4), 2)] * 8; T = [dev[j] for i, j in itertools.permutations(range(4), 2)] * 8
r4 = ctx.align_batch(S, T, None, p, s3d.ExecOptions(check_interval=4))
bad = 0
for i in range(50):
    r0 = ctx.align_batch(S, T, None, p, s3d.ExecOptions())
    bad += not np.array_equal(r0, r4)
print('96-pair early-exit batch x 50: %d differ; iterations %s' % (bad, sorted(set(int(x) for x in r4[:, 13]))))
