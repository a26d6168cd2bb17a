"""The C ABI cannot be killed from outside.

The reference's contract is that a failed registration is an exception the caller logs and moves on from
(ScanSensor.cpp:74-77, :124-127, :159-166 catch std::exception around every createConstraint).  The drop-in boundary is
extern "C": an exception that left it would be std::terminate in the host application.  Every entry point therefore runs
under a function-level guard (s3d_api.hip fail_current) that turns ANY exception - a HIP error, std::bad_alloc /
std::length_error from a host container sized by the caller's counts, a std::system_error from a mutex or a thread - into
a status with s3d_last_error set.  Tested here: hostile arguments come back as statuses (in a child process, so that an
abort would be seen as an exit code, not as a dead test run), the context keeps working afterwards, and thousands of
mixed calls from two threads leave the device memory where it was."""
import ctypes as C
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_HOSTILE = r"""
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, %(root)r)
import slam3d_amd as s3d
A = s3d.api
L = s3d.load_library()
out = {}
ctx = s3d.Context(0)
other = s3d.Context(0)
h = ctx._h
fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
a, b, _ = s3d.make_pair(6000, 3)
a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
p = s3d.default_params(point_cloud_density=0.1, maximum_iterations=10)
ident = A._colmajor(np.eye(4))
res = np.empty(16); info = A.AlignInfo()
def align(src, ns, ss, tgt, nt, st, guess=ident, params=p, opts=None):
    return L.s3d_align(h, A._fp(src), ns, ss, A._fp(tgt), nt, st, A._dp(guess), C.byref(params),
                       C.byref(opts) if opts else None, A._dp(res), C.byref(info))
good = align(a, len(a), 3, b, len(b), 3); T_good = res.copy()
out["good"] = good
# ---- sizes and pointers
NULLV = C.POINTER(C.c_void_p)()
out["batch_2p30_null"] = L.s3d_align_batch(h, 2 ** 30, NULLV, NULLV, None, C.byref(p), None, None, None)
out["batch_negative"] = L.s3d_align_batch(h, -1, NULLV, NULLV, None, C.byref(p), None, None, None)
out["batch_null_params"] = L.s3d_align_batch(h, 0, NULLV, NULLV, None, None, None, None, None)
out["upload_negative_n"] = L.s3d_cloud_upload(h, A._fp(a), -5, 3, C.byref(C.c_void_p()))
out["upload_stride_2"] = L.s3d_cloud_upload(h, A._fp(a), 100, 2, C.byref(C.c_void_p()))
out["upload_null_xyz"] = L.s3d_cloud_upload(h, None, 2 ** 31 - 1, 3, C.byref(C.c_void_p()))
out["upload_null_out"] = L.s3d_cloud_upload(h, A._fp(a), 100, 3, None)
out["upload_many_2p30_null"] = L.s3d_cloud_upload_many(h, 2 ** 30, None, None, 3, None)
out["align_stride_2"] = align(a, len(a), 2, b, len(b), 3)
out["align_negative_n"] = align(a, -1, 3, b, len(b), 3)
out["align_null_ctx"] = L.s3d_align(None, A._fp(a), 10, 3, A._fp(b), 10, 3, A._dp(ident), C.byref(p), None, A._dp(res), None)
out["knn_k0"] = L.s3d_knn_normals(h, A._fp(a), 1000, 3, 0, A._fp(np.empty((1000, 3), np.float32)))
out["knn_k65"] = L.s3d_knn_normals(h, A._fp(a), 1000, 3, 65, A._fp(np.empty((1000, 3), np.float32)))
out["knn_k_gt_n"] = L.s3d_knn_normals(h, A._fp(a), 10, 3, 20, A._fp(np.empty((10, 3), np.float32)))
out["voxel_negative"] = L.s3d_voxel_downsample(h, A._fp(a), -3, 3, 0.1, A._fp(np.empty((4, 3), np.float32)), C.byref(C.c_int()))
out["nn_null_out"] = L.s3d_nn_search(h, A._fp(a), 100, 3, A._fp(b), 100, 3, 1.0, None, None)
out["export_null_cloud"] = int(L.s3d_cloud_cache_export(h, None, None, 0))
out["import_garbage"] = None
cl = ctx.upload(a)
junk = C.create_string_buffer(b"\x53\x33\x44\x42" + bytes(range(200)), 204)
out["import_garbage"] = L.s3d_cloud_cache_import(h, cl.handle, junk, 204)
out["import_tiny"] = L.s3d_cloud_cache_import(h, cl.handle, junk, 3)
out["export_negative_capacity"] = int(L.s3d_cloud_cache_export(h, cl.handle, junk, -5))
out["link_bad_vertex"] = L.s3d_link_candidates(3, A._dp(np.zeros(9)), None, 0, None, 7, C.byref(A.LinkPolicyC(1.0, 1, 10, 1, 0)),
                                               None, 0, C.byref(C.c_int()))
out["sweep_shard_null"] = (L.s3d_sweep_shard_range(10, 0, 0, None, None), 0)[1]
out["sweep_bad_device"] = L.s3d_sweep_create(1, (C.c_int * 1)(99), C.byref(C.c_void_p()))
out["context_bad_device"] = L.s3d_context_create(99, None, C.byref(C.c_void_p()))
out["context_null_out"] = L.s3d_context_create(0, None, None)
# ---- values: a NaN / Inf guess, NaN and huge parameters, hostile options
def with_guess(v):
    g = np.eye(4); g[0, 3] = v
    return align(a, len(a), 3, b, len(b), 3, guess=A._colmajor(g))
out["guess_nan"] = with_guess(np.nan)
out["guess_inf"] = with_guess(np.inf)
out["guess_1e30"] = with_guess(1e30)
g = np.full((4, 4), np.nan)
out["guess_all_nan"] = align(a, len(a), 3, b, len(b), 3, guess=A._colmajor(g))
for name, kw in (("density_nan", dict(point_cloud_density=float("nan"))), ("density_negative", dict(point_cloud_density=-1.0)),
                 ("density_1e-30", dict(point_cloud_density=1e-30)), ("density_1e30", dict(point_cloud_density=1e30)),
                 ("maxcorr_nan", dict(max_correspondence_distance=float("nan"))), ("maxcorr_negative", dict(max_correspondence_distance=-2.0)),
                 ("maxcorr_zero", dict(max_correspondence_distance=0.0)), ("maxcorr_1e30", dict(max_correspondence_distance=1e30)),
                 ("k_zero", dict(correspondence_randomness=0)), ("k_negative", dict(correspondence_randomness=-7)),
                 ("k_huge", dict(correspondence_randomness=2 ** 30)), ("k_65", dict(correspondence_randomness=65)),
                 ("iters_negative", dict(maximum_iterations=-3)), ("inner_negative", dict(maximum_optimizer_iterations=-3)),
                 ("inner_zero", dict(maximum_optimizer_iterations=0)), ("eps_nan", dict(transformation_epsilon=float("nan"))),
                 ("rot_eps_nan", dict(rotation_epsilon=float("nan"))), ("alg_99", dict(registration_algorithm=99)),
                 ("alg_negative", dict(registration_algorithm=-1)),
                 ("ndt_resolution_zero", dict(registration_algorithm=s3d.ALG_NDT, resolution=0.0)),
                 ("ndt_resolution_nan", dict(registration_algorithm=s3d.ALG_NDT, resolution=float("nan"))),
                 ("ndt_step_nan", dict(registration_algorithm=s3d.ALG_NDT, step_size=float("nan"))),
                 ("ndt_outlier_2", dict(registration_algorithm=s3d.ALG_NDT, outlier_ratio=2.0)),
                 ("icp_k_zero", dict(registration_algorithm=s3d.ALG_ICP, correspondence_randomness=0))):
    q = s3d.default_params(point_cloud_density=0.1, maximum_iterations=10)
    for k, v in kw.items():
        setattr(q, k, v)
    out["param_" + name] = align(a, len(a), 3, b, len(b), 3, params=q)
for name, kw in (("accum_blocks_5", dict(debug_accum_blocks=5)), ("accum_blocks_negative", dict(debug_accum_blocks=-4)),
                 ("cells_negative", dict(grid_cells_per_point=-9)), ("cells_huge", dict(grid_cells_per_point=2 ** 30)),
                 ("force_negative", dict(force_iterations=-1)), ("interval_negative", dict(check_interval=-5)),
                 ("interval_huge", dict(check_interval=2 ** 30)), ("profile_9", dict(profile=9)),
                 ("all_debug_bits", dict(debug_flags=0xFFFFFFFF))):
    out["opts_" + name] = align(a, len(a), 3, b, len(b), 3, opts=s3d.ExecOptions(**kw))
# ---- data: non-finite, degenerate and empty clouds
bad = a.copy(); bad[::7] = np.nan; bad[3::11] = np.inf
out["cloud_nonfinite"] = align(bad, len(bad), 3, b, len(b), 3)
allnan = np.full_like(a, np.nan)
out["cloud_all_nan"] = align(allnan, len(allnan), 3, b, len(b), 3)
same = np.zeros_like(a)
out["cloud_one_point_repeated"] = align(same, len(same), 3, same, len(same), 3)
huge = (a * np.float32(1e30)).astype(np.float32)
out["cloud_1e30"] = align(huge, len(huge), 3, b, len(b), 3)
line = np.zeros_like(a); line[:, 0] = np.linspace(0, 50, len(a))
out["cloud_collinear"] = align(line, len(line), 3, line, len(line), 3)
out["cloud_empty"] = align(a, 0, 3, b, len(b), 3)
# ---- the guard itself: exceptions of every kind raised inside an entry point
for kind in range(6):
    st = L.s3d_debug_raise(h, kind)
    out["raise_%%d" %% kind] = [st, ctx.last_error()]
# ---- a context keeps working after ANOTHER context is destroyed, and after everything above
other.close()
st = align(a, len(a), 3, b, len(b), 3)
out["after"] = [st, bool(np.array_equal(res, T_good))]
cl.release()
ctx.close()
print("RESULT " + json.dumps(out))
"""


def _run_child(code, timeout=600):
    """Runs `code` in a child python; every `out[...] = ` statement first announces itself on stdout (flushed), so that
    the case that killed the child can be named."""
    import re
    # (S3D_HOSTILE_SKIP="name,name": a dev run that steps over known crashers to find the next one - tools_dev/hostile_all.sh)
    code = "import os\n_SKIP = set(filter(None, os.environ.get('S3D_HOSTILE_SKIP', '').split(',')))\n" + re.sub(
        r"(?m)^(\s*)out\[(.+?)\] = ", r'\1print("CASE", \2, flush=True)\n\1if (\2) in _SKIP: pass\n\1else: out[\2] = ', code)
    env = dict(os.environ)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    return r


def test_hostile_arguments_return_a_status_and_never_abort():
    r = _run_child(_HOSTILE % {"root": ROOT})
    cases = [x for x in r.stdout.splitlines() if x.startswith("CASE ")]
    assert r.returncode == 0, "the child died (exit code %d) in %s: an entry point aborted\n%s" % (
        r.returncode, cases[-1] if cases else "its set-up", r.stderr[-3000:])
    line = [x for x in r.stdout.splitlines() if x.startswith("RESULT ")]
    assert line, r.stdout[-2000:]
    out = json.loads(line[-1][7:])
    INVALID, BACKEND = 7, 8
    assert out["good"] == 0
    for k in ("batch_2p30_null", "batch_negative", "batch_null_params", "upload_negative_n", "upload_stride_2",
              "upload_null_xyz", "upload_null_out", "upload_many_2p30_null", "align_stride_2", "align_negative_n",
              "align_null_ctx", "knn_k0", "knn_k65", "knn_k_gt_n", "voxel_negative", "nn_null_out", "import_garbage",
              "import_tiny", "link_bad_vertex", "context_null_out", "sweep_bad_device", "opts_accum_blocks_5",
              "opts_accum_blocks_negative"):
        assert out[k] == INVALID, (k, out[k])
    assert out["export_null_cloud"] == -INVALID and out["export_negative_capacity"] == -INVALID
    assert out["context_bad_device"] == BACKEND
    # values: whatever the registration makes of them, it is a status of the enumeration (a NaN guess / parameter makes
    # the reference throw or return garbage: "not OK or OK" is all that is promised - but never an abort or a hang)
    for k, v in out.items():
        if k.startswith(("guess_", "param_", "opts_", "cloud_")):
            assert isinstance(v, int) and 0 <= v <= 9, (k, v)
    assert out["param_alg_99"] == 5 and out["param_alg_negative"] == 5          # UNKNOWN_ALGORITHM (PointCloudSensor.cpp:163)
    assert out["cloud_empty"] == 1 and out["cloud_all_nan"] == 1                # TOO_FEW_POINTS (:132-135)
    for k in ("guess_nan", "guess_all_nan", "guess_inf"):
        assert out[k] != 0, (k, out[k])                                         # nothing can be "aligned" to a NaN pose
    # the guard: every kind of exception comes back as a status with a message
    assert out["raise_0"][0] == BACKEND and "HIP error" in out["raise_0"][1]
    assert out["raise_1"][0] == BACKEND and "bad_alloc" in out["raise_1"][1]
    assert out["raise_2"][0] == INVALID and "length_error" in out["raise_2"][1]
    assert out["raise_3"][0] == BACKEND and "raised on request" in out["raise_3"][1]
    assert out["raise_4"][0] == BACKEND and "unknown exception" in out["raise_4"][1]
    assert out["raise_5"][0] == BACKEND and "bad_alloc" in out["raise_5"][1]
    # and the context still gives the bit-identical registration afterwards
    assert out["after"] == [0, True]
    print("hostile statuses:", {k: v for k, v in out.items() if k.startswith(("guess_", "param_", "opts_", "cloud_"))})


def _mem_free():
    hip = C.CDLL("libamdhip64.so")
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value


def test_resource_soak_leaves_device_memory_where_it_was(fixture_clouds):
    """5 000 mixed calls (s3d_align, s3d_align_batch, s3d_cloud_upload_many, cache export / import, map building, k-NN,
    createConstraint with and without the coarse stage, failing calls in between) on two contexts from two threads: hipMemGetInfo ends within 64 MiB of where it started
    (the workspaces are grown on demand and kept: the first 50 calls of each kind are the warm-up), and the contexts'
    caches end empty."""
    import slam3d_amd as s3d
    clouds = [np.ascontiguousarray(c[::3, :3]) for c in fixture_clouds]            # ~10 k points: ~0.5 ms per call
    p = s3d.default_params(maximum_iterations=6)
    pn = s3d.default_params(registration_algorithm=s3d.ALG_NDT, maximum_iterations=3)
    pc = s3d.default_params(point_cloud_density=0.5, maximum_iterations=6)
    errors = []

    def worker(ctx, seed, rounds):
        rng = np.random.default_rng(seed)
        try:
            for it in range(rounds):
                kind = it % 10
                i, j = rng.integers(0, 4, 2)
                if kind in (0, 1, 2):
                    st, T, info = ctx.align(clouds[i], clouds[j], np.eye(4), p)
                    assert 0 <= st <= 4
                elif kind in (3, 4):
                    hs = ctx.upload_many([clouds[i], clouds[j], clouds[(i + 1) % 4]])
                    on = s3d.ExecOptions(cache_prepass=1 if kind == 4 else 0)
                    rec = ctx.align_batch(hs[:2], hs[1:], None, p, on)
                    assert rec.shape == (2, 16)
                    if kind == 4:
                        blob = ctx.cache_export(hs[0])
                        assert len(blob) > 0
                        ctx.cache_control(clear=True)
                        assert ctx.cache_import(hs[0], blob) == 0
                        assert ctx.cache_import(hs[1], blob[: len(blob) // 2]) == 7        # refused: nothing installed
                    for h in hs:
                        h.release()
                elif kind == 5:
                    assert ctx._L.s3d_debug_raise(ctx._h, it % 6) in (7, 8)             # a failing call in between
                    v = ctx.voxel_downsample(clouds[i], 0.3)
                    assert len(v) > 100
                elif kind == 6:
                    hs = [ctx.upload(clouds[i]), ctx.upload(clouds[j])]
                    m = ctx.build_map(hs, [np.eye(4), np.eye(4)], 0.5, 2, 0.25)
                    m.release()
                    for h in hs:
                        h.release()
                elif kind == 7:
                    n = ctx.knn_normals(clouds[i][:3000], 12)
                    assert n.shape == (3000, 3)
                elif kind == 8:
                    st, T, info = ctx.align(clouds[i], clouds[j], np.eye(4), pn)
                    assert 0 <= st <= 4
                else:
                    h = ctx.upload(clouds[i])
                    ident = np.eye(4)
                    if (it // 10) % 2 == 0:
                        r = ctx.create_constraint_clouds(h, ident, h, ident, ident, False, p, None, 1.0,
                                                         s3d.ExecOptions(cache_prepass=1))
                    else:     # (round 6) coarse + fine: the fine pre-pass on the context's private second context
                        g = ctx.upload(clouds[j])
                        r = ctx.create_constraint_clouds(h, ident, g, ident, ident, True, p, pc, 1.0, s3d.ExecOptions())
                        g.release()
                    assert r[0] in (0, 1, 2, 3, 4)
                    h.release()
        except Exception as e:       # noqa: BLE001 - reported by the main thread
            errors.append(repr(e))

    ctxs = [s3d.Context(0), s3d.Context(0)]
    try:
        for c in ctxs:
            worker(c, 1, 50)                   # warm-up: workspaces, staging, upload lanes grown to their size
        assert not errors, errors
        for c in ctxs:
            c.cache_control(clear=True)
        before = _mem_free()
        th = [threading.Thread(target=worker, args=(c, 10 + k, 2500)) for k, c in enumerate(ctxs)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors[:3]
        for c in ctxs:
            assert c.cache_control()["entries"] == 0
        after = _mem_free()
        print("device memory free before / after the soak: %.1f / %.1f MiB" % (before / 2 ** 20, after / 2 ** 20))
        assert abs(before - after) <= 64 << 20, (before, after)
    finally:
        for c in ctxs:
            c.close()
