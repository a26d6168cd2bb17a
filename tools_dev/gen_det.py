import sys, hashlib, numpy as np
sys.path.insert(0,'.')
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
with ThreadPool(16) as pool: pairs=pool.map(lambda i: s3d.make_pair(100000,i), range(32))
ser=[s3d.make_pair(100000,i) for i in range(32)]
nd=0
for i,(p,q) in enumerate(zip(pairs,ser)):
    for k in (0,1):
        if not np.array_equal(p[k],q[k]):
            d=np.abs(p[k]-q[k]); nd+=1
            print('pair',i,'cloud',k,'differs: n', (d>0).sum(), 'max', d.max())
print('differing clouds', nd, 'sha', hashlib.sha256(b''.join(x[0].tobytes()+x[1].tobytes() for x in pairs)).hexdigest()[:12])
