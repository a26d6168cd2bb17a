import os, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0,'.')
exec(open('tools_dev/real_batch.py').read().split("for fl in")[0])
o = s3d.ExecOptions(profile=2)
for _ in range(2):
    rec = ctx.align_batch(src, tgt, None, p, o)
pr = ctx.last_profile()
n = pr['nn_launches']
print('launches', n, 'queries per launch', pr['nn_queries']/max(n,1))
print('searched', pr['nn_searched'][:n]); print('unseeded', pr['nn_unseeded'][:n]); print('records', pr['nn_records'][:n]); print('rec searched', pr['nn_records_searched'][:n])
print('ms', [round(x,3) for x in pr['nn_launch_ms'][:n]])
print('iterations', sorted(rec[:,13].astype(int).tolist()))
