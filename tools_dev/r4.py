"""dev tool (round 4): stage split, per-pass NN launch times, record-level skip rates and a hash of the records of
NPAIRS pairs of the default batch, for a list of s3d_exec_options.debug_flags settings in ONE process (timings of
variants are only comparable inside one process).  usage: python tools_dev/r4.py [flags_hex ...]   (default: 0 0x100000)
env: NPAIRS (128), POINTS (100000), ITERS (20), DENSITY (0.02), SINGLE=1 adds the one-pair latency per variant."""
import os, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slam3d_amd as s3d
from multiprocessing.pool import ThreadPool
NP = int(os.environ.get('NPAIRS', '128')); PTS = int(os.environ.get('POINTS', '100000'))
IT = int(os.environ.get('ITERS', '20')); DENS = float(os.environ.get('DENSITY', '0.02'))
variants = [int(x, 0) for x in sys.argv[1:]] or [0, 0x100000]
with ThreadPool(16) as pool: pairs = pool.map(lambda i: s3d.make_pair(PTS, i), range(NP))
ctx = s3d.Context(0)
a = [ctx.upload(p[0]) for p in pairs]; b = [ctx.upload(p[1]) for p in pairs]
p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=DENS, maximum_iterations=IT)
for rep in range(2):
    for fl in variants:
        o = s3d.ExecOptions(force_iterations=1, profile=1, debug_flags=fl)
        r = []
        for i in range(4):
            t = time.perf_counter(); out = ctx.align_batch(a, b, None, p, o); dt = (time.perf_counter() - t) * 1e3
            pr = ctx.last_profile(); r.append((dt, pr['voxel_ms'], pr['grid_ms'], pr['normals_ms'], pr['nn_ms'], pr['icp_ms'], pr['fitness_ms']))
        r = np.array(r)[1:].mean(0)
        print('flags %#x: step %.2f voxel %.2f grid %.2f normals %.2f nn %.2f icp %.2f fit %.2f ms  hash %.17g' %
              ((fl,) + tuple(r) + (float(np.abs(out[:, :12]).sum()),)), flush=True)
        print('   nn ms:', ' '.join('%.3f' % x for x in pr['nn_launch_ms'][:IT]), flush=True)
        if rep == 1:
            o2 = s3d.ExecOptions(force_iterations=1, profile=2, debug_flags=fl)
            ctx.align_batch(a, b, None, p, o2); pr = ctx.last_profile()
            print('   searched:', pr['nn_searched'][:IT])
            print('   unseeded:', pr['nn_unseeded'][:IT])
            print('   records tested:', pr['nn_records'][:IT])
            print('   records failed:', pr['nn_records_searched'][:IT], flush=True)
            if os.environ.get('SINGLE') == '1':
                o3 = s3d.ExecOptions(force_iterations=1, profile=0, debug_flags=fl)
                ctx.align_batch(a[:1], b[:1], None, p, o3)
                t = time.perf_counter()
                for i in range(20): ctx.align_batch(a[:1], b[:1], None, p, o3)
                print('   single pair %.3f ms' % ((time.perf_counter() - t) * 50), flush=True)
