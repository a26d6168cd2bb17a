"""First correspondence pass of the default batch (256 x 100k pairs) through s3d_profile_nn_kernel:
S3D_DBG_FIRSTPASS unset -> the product kernel; 1 -> lean probe kernel, lock-step row loops; 2 -> lean probe kernel,
every lane iterating over its own rows.  The probes are measurement aids (s3d_kernels.h), not product paths."""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ".")
    from multiprocessing.pool import ThreadPool
    import numpy as np
    import slam3d_amd as s3d
    NP = int(os.environ.get("NPAIRS", "256"))
    pairs = ThreadPool(32).map(lambda i: s3d.make_pair(100000, i), range(NP))
    ctx = s3d.Context(0)
    a = [ctx.upload(p[0]) for p in pairs]; b = [ctx.upload(p[1]) for p in pairs]
    p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, point_cloud_density=0.02, maximum_iterations=20)
    r = ctx.profile_nn_kernel(a, b, None, p, reps=10)
    print("%-8s first pass %.3f ms" % (os.environ.get("S3D_DBG_FIRSTPASS", "product"), r["avg_ms"]))
else:
    for v in (None, "1", "2"):
        env = dict(os.environ)
        env.pop("S3D_DBG_FIRSTPASS", None)
        if v:
            env["S3D_DBG_FIRSTPASS"] = v
        sys.stdout.write(subprocess.check_output([sys.executable, __file__, "child"], env=env, stderr=subprocess.DEVNULL).decode())
