"""dev tool (round 6): the 96 registrations on 192 distinct real clouds for a list of grid budgets (cells per point), with the
stage split and the k-NN decline counts.  usage: real_grid_budget.py [budget ...]"""
import os, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'real_batch.py')).read().split("for fl in")[0])
budgets = [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4, 6, 8]
for rep in range(2):
    for b in budgets:
        o = s3d.ExecOptions(profile=1, grid_cells_per_point=b)
        ts = []
        for _ in range(4):
            t = time.perf_counter(); rec = ctx.align_batch(src, tgt, None, p, o); ts.append((time.perf_counter() - t) * 1e3)
        pr = ctx.last_profile()
        if rep == 1:
            print('budget %d: call %.2f ms  pre-pass %.2f normals %.2f icp %.2f (nn %.2f) fit %.2f  ok %d' % (
                b, np.mean(ts[1:]), pr['voxel_ms'] + pr['grid_ms'], pr['normals_ms'], pr['icp_ms'], pr['nn_ms'], pr['fitness_ms'],
                int((rec[:, 15] == 0).sum())), flush=True)
            print('   nn ms:', ' '.join('%.3f' % x for x in pr['nn_launch_ms'][:10]), flush=True)
            ctx.align_batch(src[:1], tgt[:1], None, p, s3d.ExecOptions(profile=1, grid_cells_per_point=b, debug_flags=s3d.api.DBG_PRINT_KNN))
