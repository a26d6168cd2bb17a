import sys, time, numpy as np
sys.path.insert(0,'.')
import slam3d_amd as s3d, oracle
from tests.conftest import transform_delta
G='tests/golden'
clouds=[np.load(f'{G}/cloud{i}.npz')['xyzi'].astype(np.float32) for i in range(1,5)]
ctx=s3d.Context(0)
for a,b,g in ((0,1,None),(1,2,None),(2,3,None),(0,3,2.0)):
    guess=np.eye(4)
    if g: guess[0,3]=g
    po=oracle.default_params(registration_algorithm=oracle.ALG_NDT)
    so,To,io=oracle.align(clouds[a],clouds[b],guess,po)
    pg=s3d.default_params(registration_algorithm=s3d.ALG_NDT)
    t=time.time(); sg,Tg,ig=ctx.align(clouds[a],clouds[b],guess,pg); dt=time.time()-t
    print('pair',a,b,'status',so,sg,'iters',io['iterations'],ig['iterations'],'cells',io['correspondences'],ig['correspondences'],'fit %.6f %.6f'%(io['fitness'],ig['fitness']),'delta',transform_delta(To,Tg),'gpu %.1f ms evals %d'%(dt*1e3, ig['evaluations']))
# synthetic 100k-point pair (bench generator), NDT defaults
a, b, Ttrue = s3d.make_pair(100000, 0)
da, db = ctx.upload(a), ctx.upload(b)
pg = s3d.default_params(registration_algorithm=s3d.ALG_NDT, point_cloud_density=0.02)
ctx.align_clouds(da, db, np.eye(4), pg)
t = time.time()
for _ in range(5): sg, Tg, ig = ctx.align_clouds(da, db, np.eye(4), pg)
dt = (time.time() - t) / 5
print('synthetic 100k NDT: status', sg, 'iters', ig['iterations'], 'evals', ig['evaluations'], 'cells', ig['correspondences'], '%.2f ms per registration' % (dt * 1e3), 'err vs truth', transform_delta(Ttrue, Tg))
# batch of 64 synthetic pairs
from multiprocessing.pool import ThreadPool
with ThreadPool(16) as pool: pairs = pool.map(lambda i: s3d.make_pair(100000, i), range(64))
sa = [ctx.upload(x[0]) for x in pairs]; sb = [ctx.upload(x[1]) for x in pairs]
ctx.align_batch(sa, sb, None, pg)
t = time.time(); rec = ctx.align_batch(sa, sb, None, pg); dt = time.time() - t
errs = [transform_delta(pairs[i][2], s3d.api.record_transform(rec[i]))[0] for i in range(64)]
print('batch of 64 NDT: %.1f ms (%.2f ms per pair), ok %d, median err %.2e m' % (dt * 1e3, dt * 1e3 / 64, int((rec[:, 15] == 0).sum()), np.median(errs)))
