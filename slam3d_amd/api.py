"""ctypes binding of include/slam3d_hip.h (libslam3d_hip.so).  No compute happens in Python."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("S3D_LIB_PATH") or os.path.join(_HERE, "lib", "libslam3d_hip.so")   # override: A/B of builds
_CSRC = os.path.join(_HERE, "csrc")

ALG_ICP, ALG_GICP, ALG_GICP_OMP, ALG_NDT, ALG_NDT_OMP = range(5)
STATUS_NAMES = ["OK", "TOO_FEW_POINTS", "NOT_CONVERGED", "FITNESS_EXCEEDED", "TOO_FAR_FROM_GUESS",
                "UNKNOWN_ALGORITHM", "UNSUPPORTED_ALGORITHM", "INVALID_ARGUMENT", "BACKEND_ERROR", "OMP_UNAVAILABLE"]
ABI_VERSION = 4          # include/slam3d_hip.h S3D_ABI_VERSION (struct layouts of this binding)
EDGE_RECORD_DOUBLES = 16


class BackendError(RuntimeError):
    """The HIP back-end is missing or failed.  There is deliberately no CPU fallback."""


class RegParams(C.Structure):
    """s3d_reg_params == slam3d::RegistrationParameters (RegistrationParameters.hpp:36-97)."""
    _fields_ = [
        ("registration_algorithm", C.c_int), ("point_cloud_density", C.c_double), ("max_fitness_score", C.c_double),
        ("max_translation", C.c_double), ("max_rotation", C.c_double), ("euclidean_fitness_epsilon", C.c_double),
        ("transformation_epsilon", C.c_double), ("max_correspondence_distance", C.c_double),
        ("maximum_iterations", C.c_int), ("rotation_epsilon", C.c_double), ("correspondence_randomness", C.c_int),
        ("maximum_optimizer_iterations", C.c_int), ("resolution", C.c_float), ("step_size", C.c_double),
        ("outlier_ratio", C.c_double)]


class ExecOptions(C.Structure):
    _fields_ = [("force_iterations", C.c_int), ("check_interval", C.c_int), ("grid_cells_per_point", C.c_int),
                ("profile", C.c_int), ("cache_prepass", C.c_int), ("omp_unavailable", C.c_int),
                ("debug_flags", C.c_uint), ("debug_accum_blocks", C.c_int)]


# s3d_exec_options.debug_flags (include/slam3d_hip_debug.h S3D_DBG_*): each switches one fast path off, none may change a bit
DBG_NN_NO_REVALIDATE, DBG_NN_NO_FAR_SEED, DBG_NN_NO_COOP = 0x40, 0x80, 0x800
DBG_NN_NO_COMPACT, DBG_NN_NO_FIRST_KERNEL, DBG_NN_NO_SCAN27, DBG_NN_NO_SETTLED = 0x10000, 0x40000, 0x80000, 0x100000
DBG_KNN_EXACT64, DBG_SORT_CLASSIC, DBG_SORT_ONESWEEP, DBG_SCAN27_NO_COMPACT = 0x200000, 0x400000, 0x800000, 0x1000000
DBG_PRINT_KNN, DBG_SORT_FULL_KEYS, DBG_NN_FORCE_SETTLED = 0x2000000, 0x4000000, 0x8000000
DBG_NO_FUSED_PREPASS, DBG_KNN_NO_FAR_COOP, DBG_KNN_FORCE_FAR_COOP = 0x10000000, 0x20000000, 0x40000000
DBG_KNN_NO_RINGS = 0x80000000
DBG_NO_K4_OVERLAP = 0x400       # a small batch's k-NN pre-pass on the context's own stream
DBG_KNN_FORCE_RINGS = 0x200     # the ring search whatever the length of the far list (no device-side hand-over)


class CacheStats(C.Structure):
    _fields_ = [("entries", C.c_longlong), ("bytes", C.c_longlong), ("hits", C.c_longlong), ("misses", C.c_longlong)]


class AlignInfo(C.Structure):
    _fields_ = [("n_source_filtered", C.c_int), ("n_target_filtered", C.c_int), ("iterations", C.c_int),
                ("converged", C.c_int), ("correspondences", C.c_int), ("fitness", C.c_double),
                ("inner_iterations", C.c_int), ("evaluations", C.c_int)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class EdgeRecord(C.Structure):
    _fields_ = [("transform", C.c_double * 12), ("fitness", C.c_double), ("iterations", C.c_double),
                ("correspondences", C.c_double), ("status", C.c_double)]


class Profile(C.Structure):
    _fields_ = [("voxel_ms", C.c_double), ("grid_ms", C.c_double), ("normals_ms", C.c_double), ("icp_ms", C.c_double),
                ("fitness_ms", C.c_double), ("total_ms", C.c_double), ("nn_ms", C.c_double), ("nn_launches", C.c_int),
                ("nn_queries", C.c_longlong), ("nn_targets", C.c_longlong), ("nn_launch_ms", C.c_float * 64),
                ("nn_searched", C.c_int * 64), ("nn_unseeded", C.c_int * 64), ("nn_records", C.c_int * 64),
                ("nn_records_searched", C.c_int * 64)]

    def asdict(self):
        arrays = ("nn_launch_ms", "nn_searched", "nn_unseeded", "nn_records", "nn_records_searched")
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in arrays}
        for k in arrays[1:]:
            d[k] = list(getattr(self, k)[:max(min(self.nn_launches, 64), 0)])
        d["nn_launch_ms"] = [round(float(x), 4) for x in self.nn_launch_ms[:max(self.nn_launch_ms and self.nn_launches, 0)]][:64]
        return d


class MapProfile(C.Structure):
    _fields_ = [("accumulate_ms", C.c_double), ("grid_ms", C.c_double), ("count_ms", C.c_double),
                ("compact_ms", C.c_double), ("voxel_ms", C.c_double), ("total_ms", C.c_double),
                ("n_accumulated", C.c_longlong), ("n_kept", C.c_longlong), ("n_map", C.c_longlong)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class PlaneFit(C.Structure):
    """s3d_plane_fit (include/slam3d_hip.h)."""
    _fields_ = [("coefficients", C.c_float * 4), ("found", C.c_int), ("n_inliers", C.c_int),
                ("iterations", C.c_int), ("hypotheses_scored", C.c_int)]

    def asdict(self):
        return {"coefficients": np.array(list(self.coefficients), np.float32), "found": bool(self.found),
                "n_inliers": self.n_inliers, "iterations": self.iterations,
                "hypotheses_scored": self.hypotheses_scored}


def lib_path():
    return _LIB


_SOURCES = [os.path.join(_CSRC, f) for f in ("s3d_api.hip", "s3d_kernels.h", "s3d_core.h", "s3d_ndt.h", "s3d_sweep.h",
                                              "s3d_candidates.h")] + \
           [os.path.join(_HERE, "..", "include", f) for f in ("slam3d_hip.h", "slam3d_hip_debug.h",
                                                              "slam3d_registration_types.h")]


def source_hash():
    """sha256 over the sources of libslam3d_hip.so, in the order of csrc/Makefile (SRCS then DEPS): what
    s3d_source_hash() of a binary built from this tree returns."""
    import hashlib
    h = hashlib.sha256()
    for f in _SOURCES:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def binary_source_hash(path=None):
    """s3d_source_hash() of a built library, read from the file's bytes (no dlopen: the file may be about to be
    rebuilt); '' if it carries none."""
    import re
    path = path or _LIB
    if not os.path.exists(path):
        return ""
    with open(path, "rb") as fh:
        m = re.search(rb"S3D_SOURCE_HASH=([0-9a-f]{64})", fh.read())
    return m.group(1).decode() if m else ""


def build(force=False, verbose=False):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU).  Stale = the hash compiled
    into the binary differs from the hash of the sources next to it (not file times: a pushed tree has fresh ones)."""
    stale = force or not os.path.exists(_LIB) or binary_source_hash() != source_hash()
    if stale:
        if _lib is not None:
            raise BackendError("libslam3d_hip.so is already loaded and stale: rebuild in a fresh process")
        cmd = ["make", "-C", _CSRC, "-B"]
        subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return _LIB


class GraphEdge(C.Structure):       # include/slam3d_hip.h s3d_graph_edge
    _fields_ = [("source", C.c_int), ("target", C.c_int), ("se3", C.c_int), ("own_sensor", C.c_int)]


class LinkPolicyC(C.Structure):     # include/slam3d_hip.h s3d_link_policy
    _fields_ = [("neighbor_radius", C.c_float), ("max_neighbor_links", C.c_int), ("min_loop_length", C.c_uint),
                ("patch_building_range", C.c_uint), ("static_graph", C.c_int)]


_lib = None


def load_library():
    """dlopen libslam3d_hip.so and declare every symbol of include/slam3d_hip.h.  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise BackendError("HIP extension missing: %s (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                           "slam3d_amd has no CPU fallback" % _LIB)
    L = C.CDLL(_LIB)
    vp, fp, dp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int)
    pp, op = C.POINTER(RegParams), C.POINTER(ExecOptions)
    sig = {
        "s3d_abi_version": (C.c_int, []),
        "s3d_source_hash": (C.c_char_p, []),
        "s3d_context_create": (C.c_int, [C.c_int, vp, C.POINTER(vp)]),
        "s3d_context_create_priority": (C.c_int, [C.c_int, C.c_int, C.POINTER(vp)]),
        "s3d_context_create_cu_mask": (C.c_int, [C.c_int, C.POINTER(C.c_uint32), C.c_int, C.POINTER(vp)]),
        "s3d_context_destroy": (None, [vp]),
        "s3d_last_error": (C.c_char_p, [vp]),
        "s3d_backend_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
        "s3d_last_profile": (C.c_int, [vp, C.POINTER(Profile)]),
        "s3d_context_cache_control": (C.c_int, [vp, C.c_longlong, C.c_int, C.POINTER(CacheStats)]),
        "s3d_cloud_cache_export": (C.c_longlong, [vp, vp, vp, C.c_longlong]),
        "s3d_cloud_cache_import": (C.c_int, [vp, vp, vp, C.c_longlong]),
        "s3d_default_params": (None, [pp]),
        "s3d_cloud_upload": (C.c_int, [vp, fp, C.c_int, C.c_int, C.POINTER(vp)]),
        "s3d_cloud_upload_many": (C.c_int, [vp, C.c_int, C.POINTER(fp), C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]),
        "s3d_context_set_upload_threads": (C.c_int, [vp, C.c_int]),
        "s3d_cloud_wrap_device": (C.c_int, [vp, vp, C.c_int, C.POINTER(vp)]),
        "s3d_cloud_size": (C.c_int, [vp]),
        "s3d_cloud_release": (None, [vp, vp]),
        "s3d_voxel_downsample": (C.c_int, [vp, fp, C.c_int, C.c_int, C.c_double, fp, ip]),
        "s3d_nn_search": (C.c_int, [vp, fp, C.c_int, C.c_int, fp, C.c_int, C.c_int, C.c_double, ip, fp]),
        "s3d_knn_normals": (C.c_int, [vp, fp, C.c_int, C.c_int, C.c_int, fp]),
        "s3d_align": (C.c_int, [vp, fp, C.c_int, C.c_int, fp, C.c_int, C.c_int, dp, pp, op, dp, C.POINTER(AlignInfo)]),
        "s3d_align_batch": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), dp, pp, op, C.POINTER(EdgeRecord),
                                      C.POINTER(AlignInfo)]),
        "s3d_create_constraint": (C.c_int, [vp, fp, C.c_int, C.c_int, dp, fp, C.c_int, C.c_int, dp, dp, C.c_int, pp, pp,
                                            C.c_double, op, dp, dp, C.POINTER(AlignInfo)]),
        "s3d_cloud_download": (C.c_int, [vp, vp, fp, C.c_int]),
        "s3d_cloud_accumulate": (C.c_int, [vp, C.c_int, C.POINTER(vp), dp, dp, C.POINTER(vp)]),
        "s3d_align_clouds": (C.c_int, [vp, vp, vp, dp, pp, op, dp, C.POINTER(AlignInfo)]),
        "s3d_create_constraint_clouds": (C.c_int, [vp, vp, dp, vp, dp, dp, C.c_int, pp, pp, C.c_double, op, dp, dp,
                                                   C.POINTER(AlignInfo)]),
        "s3d_remove_outliers": (C.c_int, [vp, fp, C.c_int, C.c_int, C.c_double, C.c_uint, fp, ip]),
        "s3d_remove_outliers_cloud": (C.c_int, [vp, vp, C.c_double, C.c_uint, C.POINTER(vp)]),
        "s3d_voxel_downsample_cloud": (C.c_int, [vp, vp, C.c_double, C.POINTER(vp)]),
        "s3d_build_map": (C.c_int, [vp, C.c_int, C.POINTER(vp), dp, C.c_double, C.c_uint, C.c_double, C.POINTER(vp)]),
        "s3d_last_map_profile": (C.c_int, [vp, C.POINTER(MapProfile)]),
        "s3d_fit_plane": (C.c_int, [vp, fp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.POINTER(PlaneFit)]),
        "s3d_fill_ground_plane": (C.c_int, [vp, fp, C.c_int, C.c_int, C.c_double, C.c_double, fp, C.c_int, ip,
                                            C.POINTER(PlaneFit)]),
        "s3d_sweep_create": (C.c_int, [C.c_int, ip, C.POINTER(vp)]),
        "s3d_sweep_create_cu_mask": (C.c_int, [C.c_int, ip, C.POINTER(C.c_uint32), C.c_int, C.POINTER(vp)]),
        "s3d_cu_masks": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_int]),
        "s3d_sweep_destroy": (None, [vp]),
        "s3d_sweep_ranks": (C.c_int, [vp]),
        "s3d_sweep_collective": (C.c_char_p, [vp]),
        "s3d_sweep_last_error": (C.c_char_p, [vp]),
        "s3d_sweep_context": (vp, [vp, C.c_int]),
        "s3d_sweep_cloud_create": (C.c_int, [vp, fp, C.c_int, C.c_int, C.POINTER(vp)]),
        "s3d_sweep_cloud_release": (None, [vp, vp]),
        "s3d_sweep_shard_range": (None, [C.c_int, C.c_int, C.c_int, ip, ip]),
        "s3d_align_batch_multi": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), dp, pp, op,
                                            C.POINTER(EdgeRecord)]),
        "s3d_sweep_gathered_records": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(EdgeRecord)]),
        "s3d_link_candidates": (C.c_int, [C.c_int, dp, C.POINTER(C.c_ubyte), C.c_int, C.POINTER(GraphEdge), C.c_int,
                                          C.POINTER(LinkPolicyC), ip, C.c_int, ip]),
    }
    debug_sig = {     # include/slam3d_hip_debug.h: test hooks, not part of the drop-in API
        "s3d_debug_fused_reruns": (C.c_longlong, [vp]),
        "s3d_debug_raise": (C.c_int, [vp, C.c_int]),
        "s3d_profile_nn_kernel": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), dp, pp, C.c_int, dp,
                                            C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
        "s3d_debug_filtered_nn": (C.c_int, [vp, vp, vp, C.c_double, C.c_int, C.c_double, C.c_int, fp, ip, fp, ip, ip, fp, ip]),
    }
    for name, (res, args) in list(sig.items()) + list(debug_sig.items()):
        f = getattr(L, name)  # AttributeError if the symbol is not exported
        f.restype = res
        f.argtypes = args
    if not os.environ.get("S3D_LIB_PATH") and all(os.path.exists(f) for f in _SOURCES):
        got, want = L.s3d_source_hash().decode(), source_hash()
        if got != want:     # (an A/B build named by S3D_LIB_PATH is what its user says it is)
            raise BackendError("libslam3d_hip.so was built from other sources (binary %s..., tree %s...): run "
                               "`python -c 'import __graft_entry__ as g; g.build()'`" % (got[:12], want[:12]))
    if L.s3d_abi_version() != ABI_VERSION:
        raise BackendError("libslam3d_hip.so has ABI version %d, this binding %d: rebuild the library" %
                           (L.s3d_abi_version(), ABI_VERSION))
    L._s3d_symbols = sorted(sig)
    _lib = L
    return L


def default_params(**overrides):
    p = RegParams()
    load_library().s3d_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def backend_info(device=0):
    buf = C.create_string_buffer(256)
    st = load_library().s3d_backend_info(device, buf, 256)
    if st != 0:
        raise BackendError("no usable HIP device (%s)" % buf.value.decode())
    name, arch, cus, mem = buf.value.decode().split("|")
    return dict(name=name, arch=arch, compute_units=int(cus), hbm_bytes=int(mem))


def _cloud(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] not in (3, 4):
        raise ValueError("cloud must be (n,3) or (n,4) float32, got %r" % (a.shape,))
    return a, a.shape[0], a.shape[1]


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _colmajor(T):
    return np.ascontiguousarray(np.asarray(T, np.float64).T.reshape(-1))


def _from_colmajor(v):
    return np.asarray(v, np.float64).reshape(4, 4).T.copy()


class Cloud:
    """Device-resident point cloud handle (s3d_cloud)."""

    def __init__(self, ctx, handle, n):
        self.ctx, self.handle, self.n = ctx, handle, n

    def download(self):
        """(n, 3) float32 copy of the device-resident points."""
        return self.ctx.download(self)

    def release(self):
        if self.handle:
            self.ctx._L.s3d_cloud_release(self.ctx._h, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def cu_masks(device=0, reserved_cus=32):
    """(reserved, rest): two complementary CU masks (lists of 32-bit words) for Context(cu_mask=) / Sweep(cu_mask=)."""
    L = load_library()
    a = (C.c_uint32 * 32)()
    b = (C.c_uint32 * 32)()
    n = L.s3d_cu_masks(int(device), int(reserved_cus), a, b, 32)
    if n <= 0:
        raise BackendError("s3d_cu_masks failed: device %d, %d reserved compute units" % (device, reserved_cus))
    return list(a[:n]), list(b[:n])


class Context:
    """One HIP device + stream + workspace (s3d_context)."""

    def __init__(self, device=0, stream=None, high_priority=False, cu_mask=None):
        """high_priority: a private stream of the device's highest priority (s3d_context_create_priority) - for the
        latency-critical sequential registration while a loop-closure batch runs on another context.
        cu_mask: a sequence of 32-bit words, the compute units this context's stream may use
        (s3d_context_create_cu_mask)."""
        self._L = load_library()
        h = C.c_void_p()
        if stream and (cu_mask is not None or high_priority):
            raise ValueError("cu_mask / high_priority create a private stream: they cannot be combined with stream=")
        if cu_mask is not None and len(cu_mask) == 0:
            raise ValueError("cu_mask must name at least one 32-bit word")
        if cu_mask is not None and not stream:
            words = (C.c_uint32 * len(cu_mask))(*[int(w) & 0xFFFFFFFF for w in cu_mask])
            st = self._L.s3d_context_create_cu_mask(int(device), words, len(cu_mask), C.byref(h))
        elif high_priority and not stream:
            st = self._L.s3d_context_create_priority(int(device), 1, C.byref(h))
        else:
            st = self._L.s3d_context_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if st != 0 or not h:
            raise BackendError("s3d_context_create failed with %s on HIP device %d (no CPU fallback)" %
                               (STATUS_NAMES[st] if 0 <= st < len(STATUS_NAMES) else st, device))
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.s3d_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st == 8:
            raise BackendError(self._L.s3d_last_error(self._h).decode())
        return st

    # ---- stage entry points -------------------------------------------------------
    def voxel_downsample(self, xyz, leaf):
        a, n, stride = _cloud(xyz)
        out = np.empty((max(n, 1), 3), np.float32)
        m = C.c_int(0)
        st = self._check(self._L.s3d_voxel_downsample(self._h, _fp(a), n, stride, float(leaf), _fp(out), C.byref(m)))
        if st:
            raise ValueError(STATUS_NAMES[st])
        return out[:m.value].copy()

    def nn_search(self, target, query, max_distance):
        t, n, st_ = _cloud(target)
        q, m, sq = _cloud(query)
        idx = np.empty(max(m, 1), np.int32)
        d2 = np.empty(max(m, 1), np.float32)
        st = self._check(self._L.s3d_nn_search(self._h, _fp(t), n, st_, _fp(q), m, sq, float(max_distance),
                                               idx.ctypes.data_as(C.POINTER(C.c_int)), _fp(d2)))
        if st:
            raise ValueError(STATUS_NAMES[st])
        return idx[:m], d2[:m]

    # ---- test hooks (include/slam3d_hip_debug.h) ------------------------------------
    def fused_reruns(self):
        """batches of this context that the fused pre-pass could not serve and that ran again on the two-sort path"""
        return int(self._L.s3d_debug_fused_reruns(self._h))

    def debug_filtered_nn(self, source, target, leaf, fused, max_distance):
        """The registration's pre-pass of two device clouds and one NN pass (s3d_debug_filtered_nn): cell-sorted points of
        both clouds with their tie-breaking ids, the neighbour positions / squared distances of the target's points."""
        cap = max(source.n, target.n, 1)
        s4 = np.empty((cap, 4), np.float32); t4 = np.empty((cap, 4), np.float32)
        pos = np.empty(cap, np.int32); d2 = np.empty(cap, np.float32)
        ns, nt, ok = C.c_int(), C.c_int(), C.c_int()
        st = self._check(self._L.s3d_debug_filtered_nn(self._h, source.handle, target.handle, float(leaf), int(fused),
                                                       float(max_distance), cap, _fp(s4), C.byref(ns), _fp(t4), C.byref(nt),
                                                       pos.ctypes.data_as(C.POINTER(C.c_int)), _fp(d2), C.byref(ok)))
        if st:
            raise ValueError(STATUS_NAMES[st])
        s4, t4 = s4[:ns.value], t4[:nt.value]
        return dict(source_xyz=s4[:, :3].copy(), source_id=s4[:, 3].copy().view(np.uint32), target_xyz=t4[:, :3].copy(),
                    target_id=t4[:, 3].copy().view(np.uint32), pos=pos[:nt.value].copy(), d2=d2[:nt.value].copy(),
                    fused_ok=ok.value)

    def knn_normals(self, xyz, k=20):
        a, n, stride = _cloud(xyz)
        out = np.empty((max(n, 1), 3), np.float32)
        st = self._check(self._L.s3d_knn_normals(self._h, _fp(a), n, stride, int(k), _fp(out)))
        if st:
            raise ValueError(STATUS_NAMES[st])
        return out[:n]

    # ---- align / createConstraint ---------------------------------------------------
    def align(self, source, target, guess=np.eye(4), params=None, opts=None):
        s, ns, ss = _cloud(source)
        t, nt, st_ = _cloud(target)
        params = params or default_params()
        g = _colmajor(guess)
        res = np.empty(16, np.float64)
        info = AlignInfo()
        st = self._check(self._L.s3d_align(self._h, _fp(s), ns, ss, _fp(t), nt, st_, _dp(g), C.byref(params),
                                           C.byref(opts) if opts else None, _dp(res), C.byref(info)))
        return st, _from_colmajor(res), info.asdict()

    def create_constraint(self, source, source_pose, target, target_pose, odometry, loop=False, fine=None,
                          coarse=None, covariance_scale=1.0, opts=None):
        s, ns, ss = _cloud(source)
        t, nt, st_ = _cloud(target)
        fine = fine or default_params()
        coarse = coarse or default_params()
        rel = np.empty(16, np.float64)
        inf = np.empty(36, np.float64)
        info = AlignInfo()
        sp, tp, od = _colmajor(source_pose), _colmajor(target_pose), _colmajor(odometry)
        st = self._check(self._L.s3d_create_constraint(
            self._h, _fp(s), ns, ss, _dp(sp), _fp(t), nt, st_, _dp(tp), _dp(od), int(loop), C.byref(fine),
            C.byref(coarse), float(covariance_scale), C.byref(opts) if opts else None, _dp(rel), _dp(inf),
            C.byref(info)))
        return st, _from_colmajor(rel), inf.reshape(6, 6), info.asdict()

    # ---- device-resident batch --------------------------------------------------------
    def upload(self, xyz):
        a, n, stride = _cloud(xyz)
        h = C.c_void_p()
        st = self._check(self._L.s3d_cloud_upload(self._h, _fp(a), n, stride, C.byref(h)))
        if st:
            raise ValueError(STATUS_NAMES[st])
        return Cloud(self, h, n)

    def upload_many(self, clouds):
        """Bulk hand-over (s3d_cloud_upload_many): a list of (n_i, stride) float32 arrays of one stride -> Cloud handles
        that share one device allocation."""
        arrs = [_cloud(x) for x in clouds]
        if not arrs:
            return []
        stride = arrs[0][2]
        if any(a[2] != stride for a in arrs):
            raise ValueError("upload_many: all clouds must have the same stride")
        m = len(arrs)
        ptrs = (C.POINTER(C.c_float) * m)(*[_fp(a[0]) for a in arrs])
        ns = (C.c_int * m)(*[a[1] for a in arrs])
        out = (C.c_void_p * m)()
        st = self._check(self._L.s3d_cloud_upload_many(self._h, m, ptrs, ns, stride, out))
        if st:
            raise ValueError(STATUS_NAMES[st])
        return [Cloud(self, C.c_void_p(out[i]), arrs[i][1]) for i in range(m)]

    def set_upload_threads(self, n):
        """host threads upload_many may use (0 = the default, up to 8): s3d_context_set_upload_threads"""
        st = self._check(self._L.s3d_context_set_upload_threads(self._h, int(n)))
        if st:
            raise ValueError(STATUS_NAMES[st])

    def wrap_device(self, device_ptr, n):
        h = C.c_void_p()
        st = self._check(self._L.s3d_cloud_wrap_device(self._h, C.c_void_p(device_ptr), int(n), C.byref(h)))
        if st:
            raise ValueError(STATUS_NAMES[st])
        return Cloud(self, h, n)

    def _handles(self, clouds):
        arr = (C.c_void_p * len(clouds))()
        for i, c in enumerate(clouds):
            arr[i] = c.handle
        return arr

    def align_batch(self, sources, targets, guesses=None, params=None, opts=None, want_infos=False):
        """Returns an (n_pairs, 16) float64 array of edge records (12 transform col-major 3x4, fitness,
        iterations, correspondences, status) and optionally the per-pair AlignInfo list."""
        n = len(sources)
        assert len(targets) == n
        params = params or default_params()
        if guesses is None:
            guesses = np.tile(np.eye(4), (n, 1, 1))
        g = np.ascontiguousarray(np.asarray(guesses, np.float64).transpose(0, 2, 1).reshape(n, 16))
        rec = np.zeros((max(n, 1), EDGE_RECORD_DOUBLES), np.float64)
        infos = (AlignInfo * max(n, 1))() if want_infos else None
        st = self._check(self._L.s3d_align_batch(self._h, n, self._handles(sources), self._handles(targets), _dp(g),
                                                 C.byref(params), C.byref(opts) if opts else None,
                                                 rec.ctypes.data_as(C.POINTER(EdgeRecord)), infos))
        if st not in (0, 5, 6, 9):
            raise ValueError(STATUS_NAMES[st])
        rec = rec[:n]
        if want_infos:
            return rec, [infos[i].asdict() for i in range(n)]
        return rec

    def profile_nn_kernel(self, sources, targets, guesses=None, params=None, reps=20):
        n = len(sources)
        params = params or default_params()
        if guesses is None:
            guesses = np.tile(np.eye(4), (n, 1, 1))
        g = np.ascontiguousarray(np.asarray(guesses, np.float64).transpose(0, 2, 1).reshape(n, 16))
        ms, nq, nt = C.c_double(), C.c_longlong(), C.c_longlong()
        st = self._check(self._L.s3d_profile_nn_kernel(self._h, n, self._handles(sources), self._handles(targets),
                                                       _dp(g), C.byref(params), int(reps), C.byref(ms), C.byref(nq),
                                                       C.byref(nt)))
        if st:
            raise ValueError(STATUS_NAMES[st])
        return dict(avg_ms=ms.value, n_queries=nq.value, n_targets=nt.value)

    # ---- patches and maps, device-resident -----------------------
    def _new_cloud(self, h):
        return Cloud(self, h, int(self._L.s3d_cloud_size(h)))

    def _poses(self, poses):
        P = np.asarray(poses, np.float64).reshape(-1, 4, 4)
        return np.ascontiguousarray(P.transpose(0, 2, 1).reshape(-1, 16))

    def download(self, cloud):
        out = np.empty((max(cloud.n, 1), 3), np.float32)
        st = self._check(self._L.s3d_cloud_download(self._h, cloud.handle, _fp(out), 3))
        if st:
            raise ValueError(STATUS_NAMES[st])
        return out[:cloud.n]

    def accumulate(self, clouds, poses, frame=None):
        """getAccumulatedCloud (PointCloudSensor.cpp:235-256); with `frame` createCombinedMeasurement (:258-266)."""
        P = self._poses(poses) if len(clouds) else np.zeros((1, 16))
        fr = _colmajor(frame) if frame is not None else None
        h = C.c_void_p()
        st = self._check(self._L.s3d_cloud_accumulate(self._h, len(clouds), self._handles(clouds), _dp(P),
                                                      _dp(fr) if fr is not None else None, C.byref(h)))
        if st:
            raise ValueError(STATUS_NAMES[st])
        return self._new_cloud(h)

    def align_clouds(self, source, target, guess=np.eye(4), params=None, opts=None):
        params = params or default_params()
        g = _colmajor(guess)
        res = np.empty(16, np.float64)
        info = AlignInfo()
        st = self._check(self._L.s3d_align_clouds(self._h, source.handle, target.handle, _dp(g), C.byref(params),
                                                  C.byref(opts) if opts else None, _dp(res), C.byref(info)))
        return st, _from_colmajor(res), info.asdict()

    def create_constraint_clouds(self, source, source_pose, target, target_pose, odometry, loop=False, fine=None,
                                 coarse=None, covariance_scale=1.0, opts=None):
        fine = fine or default_params()
        coarse = coarse or default_params()
        rel = np.empty(16, np.float64)
        inf = np.empty(36, np.float64)
        info = AlignInfo()
        sp, tp, od = _colmajor(source_pose), _colmajor(target_pose), _colmajor(odometry)
        st = self._check(self._L.s3d_create_constraint_clouds(
            self._h, source.handle, _dp(sp), target.handle, _dp(tp), _dp(od), int(loop), C.byref(fine),
            C.byref(coarse), float(covariance_scale), C.byref(opts) if opts else None, _dp(rel), _dp(inf),
            C.byref(info)))
        return st, _from_colmajor(rel), inf.reshape(6, 6), info.asdict()

    def remove_outliers(self, xyz, radius, min_neighbors):
        """removeOutliers (:211-226).  numpy in -> numpy out; Cloud in -> Cloud out."""
        if isinstance(xyz, Cloud):
            h = C.c_void_p()
            st = self._check(self._L.s3d_remove_outliers_cloud(self._h, xyz.handle, float(radius), int(min_neighbors),
                                                               C.byref(h)))
            if st:
                raise ValueError(STATUS_NAMES[st])
            return self._new_cloud(h)
        a, n, stride = _cloud(xyz)
        out = np.empty((max(n, 1), 3), np.float32)
        m = C.c_int(0)
        st = self._check(self._L.s3d_remove_outliers(self._h, _fp(a), n, stride, float(radius), int(min_neighbors),
                                                     _fp(out), C.byref(m)))
        if st:
            raise ValueError(STATUS_NAMES[st])
        return out[:m.value].copy()

    def fit_plane(self, xyz, threshold=0.01, max_iterations=1000, probability=0.99):
        """The RANSAC plane fit of fillGroundPlane (:364-368); returns PlaneFit.asdict()."""
        a, n, stride = _cloud(xyz)
        f = PlaneFit()
        st = self._check(self._L.s3d_fit_plane(self._h, _fp(a), n, stride, float(threshold), int(max_iterations),
                                               float(probability), C.byref(f)))
        if st:
            raise ValueError(STATUS_NAMES[st])
        return f.asdict()

    def fill_ground_plane(self, xyz, radius, map_resolution=0.1):
        """fillGroundPlane (:362-388): the cloud with the ring points on its RANSAC ground plane appended."""
        a, n, stride = _cloud(xyz)
        m = C.c_int(0)
        f = PlaneFit()
        # upper bound of the ring points: (radius / res + 1) rings of (2 pi radius / res + 2) points
        res = float(map_resolution)
        cap = int((radius / res + 2) * (2 * 3.141592654 * radius / res + 3)) if res > 0 else 0
        out = np.empty((max(cap, 1), 3), np.float32)
        st = self._check(self._L.s3d_fill_ground_plane(self._h, _fp(a), n, stride, float(radius), res, _fp(out), cap,
                                                       C.byref(m), C.byref(f)))
        if st:
            raise ValueError(STATUS_NAMES[st])
        assert m.value <= cap
        return np.vstack([np.ascontiguousarray(a[:, :3]), out[:m.value]])

    def voxel_downsample_cloud(self, cloud, leaf):
        h = C.c_void_p()
        st = self._check(self._L.s3d_voxel_downsample_cloud(self._h, cloud.handle, float(leaf), C.byref(h)))
        if st:
            raise ValueError(STATUS_NAMES[st])
        return self._new_cloud(h)

    def build_map(self, clouds, poses, outlier_radius=0.2, outlier_neighbors=3, map_resolution=0.1):
        """buildMap (:301-318); defaults = PointCloudSensor constructor (:176-183)."""
        P = self._poses(poses) if len(clouds) else np.zeros((1, 16))
        h = C.c_void_p()
        st = self._check(self._L.s3d_build_map(self._h, len(clouds), self._handles(clouds), _dp(P), float(outlier_radius),
                                               int(outlier_neighbors), float(map_resolution), C.byref(h)))
        if st:
            raise ValueError(STATUS_NAMES[st])
        return self._new_cloud(h)

    def last_map_profile(self):
        p = MapProfile()
        self._L.s3d_last_map_profile(self._h, C.byref(p))
        return p.asdict()

    def cache_control(self, limit_bytes=0, clear=False):
        """s3d_context_cache_control: set the budget / clear; returns the stats dict."""
        st = CacheStats()
        self._check(self._L.s3d_context_cache_control(self._h, int(limit_bytes), int(bool(clear)), C.byref(st)))
        return {k: getattr(st, k) for k, _ in CacheStats._fields_}

    def cache_export(self, cloud):
        """s3d_cloud_cache_export: the cached pre-pass products of `cloud` as bytes (b"" when nothing is cached) -
        what a checkpoint stores next to the measurement's .s3dm file."""
        need = self._L.s3d_cloud_cache_export(self._h, cloud.handle, None, 0)
        if need < 0:
            self._check(int(-need))
            raise ValueError("s3d_cloud_cache_export: status %d" % -need)
        if need == 0:
            return b""
        buf = C.create_string_buffer(int(need))
        got = self._L.s3d_cloud_cache_export(self._h, cloud.handle, buf, int(need))
        if got != need:
            self._check(int(-got) if got < 0 else 0)
            raise ValueError("s3d_cloud_cache_export: %d instead of %d bytes" % (got, need))
        return buf.raw

    def cache_import(self, cloud, blob):
        """s3d_cloud_cache_import: install a blob of cache_export() as the cache entries of `cloud` (same points:
        checked).  Returns the status (0 = installed, 1 = rejected: s3d_last_error says why)."""
        if not blob:
            return 0
        buf = C.create_string_buffer(bytes(blob), len(blob))
        return self._check(self._L.s3d_cloud_cache_import(self._h, cloud.handle, buf, len(blob)))

    def last_error(self):
        return self._L.s3d_last_error(self._h).decode()

    def last_profile(self):
        p = Profile()
        self._L.s3d_last_profile(self._h, C.byref(p))
        return p.asdict()


def link_candidates(positions, edges, vertex, neighbor_radius=1.0, max_neighbor_links=1, min_loop_length=10,
                    patch_building_range=0, linkable=None, static_graph=False):
    """s3d_link_candidates (host only): the sources of ScanSensor::linkToNeighbors' link(source, vertex) calls.
    positions: (n, 3) translations of the corrected poses in vertex insertion order; edges: (source, target, se3,
    own_sensor) out-edges as stored."""
    pos = np.ascontiguousarray(np.asarray(positions, np.float64).reshape(-1, 3))
    n = len(pos)
    E = (GraphEdge * max(len(edges), 1))()
    for k, e in enumerate(edges):
        E[k] = GraphEdge(int(e[0]), int(e[1]), int(e[2]), int(e[3]))
    pol = LinkPolicyC(float(neighbor_radius), int(max_neighbor_links), int(min_loop_length), int(patch_building_range),
                      1 if static_graph else 0)
    lk = None
    if linkable is not None:
        lk = np.ascontiguousarray(np.asarray(linkable, np.uint8)).ctypes.data_as(C.POINTER(C.c_ubyte))
    out = np.zeros(max(n, 1), np.int32)
    cnt = C.c_int(0)
    st = load_library().s3d_link_candidates(n, _dp(pos), lk, len(edges), E, int(vertex), C.byref(pol),
                                            out.ctypes.data_as(C.POINTER(C.c_int)), len(out), C.byref(cnt))
    if st:
        raise ValueError(STATUS_NAMES[st])
    return out[:cnt.value].tolist()


def record_transform(rec):
    """(16,) edge record -> 4x4 transform."""
    T = np.eye(4)
    T[:3, :4] = np.asarray(rec[:12]).reshape(4, 3).T
    return T


class Sweep:
    """s3d_sweep: one rank (context + host thread) per device, pair list sharded in contiguous blocks, one
    all-gather of the edge records (include/slam3d_hip.h, C1)."""

    def __init__(self, devices=None, cu_mask=None):
        """cu_mask: 32-bit words, the compute units every rank's stream may use (s3d_sweep_create_cu_mask)."""
        self._L = load_library()
        h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices) if devices is not None else None
        nd = len(devices) if devices is not None else 0
        if cu_mask is not None:
            words = (C.c_uint32 * len(cu_mask))(*[int(w) & 0xFFFFFFFF for w in cu_mask])
            st = self._L.s3d_sweep_create_cu_mask(nd, arr, words, len(cu_mask), C.byref(h))
        elif devices is None:
            st = self._L.s3d_sweep_create(0, None, C.byref(h))
        else:
            st = self._L.s3d_sweep_create(len(devices), arr, C.byref(h))
        if st != 0 or not h:
            raise BackendError("s3d_sweep_create failed (status %d): no usable HIP devices / RCCL (no CPU fallback)" % st)
        self._h = h
        self.ranks = int(self._L.s3d_sweep_ranks(h))
        self.collective = self._L.s3d_sweep_collective(h).decode()
        self._clouds = []

    def upload(self, xyz):
        a, n, stride = _cloud(xyz)
        h = C.c_void_p()
        st = self._L.s3d_sweep_cloud_create(self._h, _fp(a), n, stride, C.byref(h))
        if st:
            raise ValueError(STATUS_NAMES[st])
        self._clouds.append(h)
        return h

    def align_batch(self, sources, targets, guesses=None, params=None, opts=None):
        n = len(sources)
        assert len(targets) == n
        params = params or default_params()
        if guesses is None:
            guesses = np.tile(np.eye(4), (n, 1, 1))
        g = np.ascontiguousarray(np.asarray(guesses, np.float64).transpose(0, 2, 1).reshape(max(n, 0), 16))
        rec = np.zeros((max(n, 1), EDGE_RECORD_DOUBLES), np.float64)
        sa, ta = (C.c_void_p * max(n, 1))(), (C.c_void_p * max(n, 1))()
        for i in range(n):
            sa[i], ta[i] = sources[i], targets[i]
        st = self._L.s3d_align_batch_multi(self._h, n, sa, ta, _dp(g), C.byref(params), C.byref(opts) if opts else None,
                                           rec.ctypes.data_as(C.POINTER(EdgeRecord)))
        if st == 8:
            raise BackendError(self._L.s3d_sweep_last_error(self._h).decode())
        if st not in (0, 5, 6, 9):
            raise ValueError(STATUS_NAMES[st])
        return rec[:n]

    def gathered(self, rank, n_pairs):
        rec = np.zeros((max(n_pairs, 1), EDGE_RECORD_DOUBLES), np.float64)
        st = self._L.s3d_sweep_gathered_records(self._h, rank, n_pairs, rec.ctypes.data_as(C.POINTER(EdgeRecord)))
        if st:
            raise ValueError(STATUS_NAMES[st])
        return rec[:n_pairs]

    def shard_range(self, n_pairs, rank):
        lo, hi = C.c_int(), C.c_int()
        self._L.s3d_sweep_shard_range(n_pairs, self.ranks, rank, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def close(self):
        if getattr(self, "_h", None):
            for c in self._clouds:
                self._L.s3d_sweep_cloud_release(self._h, c)
            self._clouds = []
            self._L.s3d_sweep_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
