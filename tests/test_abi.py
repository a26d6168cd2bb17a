"""The C-ABI library loads and exports every symbol include/slam3d_hip.h declares (no compute)."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT


def declared_symbols(header="slam3d_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(s3d_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import slam3d_amd
    slam3d_amd.build()
    lib = ctypes.CDLL(slam3d_amd.lib_path())
    names = declared_symbols()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    L = slam3d_amd.load_library()
    assert sorted(L._s3d_symbols) == names     # the binding declares exactly the header's functions
    # the test hooks of include/slam3d_hip_debug.h (not API: the public header must not pull them in)
    hooks = declared_symbols("slam3d_hip_debug.h")
    assert hooks == ["s3d_debug_filtered_nn", "s3d_debug_fused_reruns", "s3d_debug_raise", "s3d_profile_nn_kernel"]
    for n in hooks:
        assert hasattr(lib, n), "missing export: " + n
    public = open(os.path.join(ROOT, "include", "slam3d_hip.h")).read()
    assert "#include \"slam3d_hip_debug.h\"" not in public and "#define S3D_DBG_" not in public


def test_struct_layouts_match_the_header(tmp_path):
    import slam3d_amd
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "slam3d_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(s3d_reg_params),sizeof(s3d_edge_record),sizeof(s3d_exec_options),sizeof(s3d_align_info),'
                   'sizeof(s3d_profile),offsetof(s3d_reg_params,rotation_epsilon),sizeof(s3d_map_profile));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes[0] == ctypes.sizeof(slam3d_amd.RegParams)
    assert sizes[1] == ctypes.sizeof(slam3d_amd.EdgeRecord) == 128       # the all-gathered unit
    assert sizes[2] == ctypes.sizeof(slam3d_amd.ExecOptions)
    assert sizes[3] == ctypes.sizeof(slam3d_amd.AlignInfo)
    assert sizes[4] == ctypes.sizeof(slam3d_amd.Profile)
    assert sizes[5] == slam3d_amd.RegParams.rotation_epsilon.offset
    assert sizes[6] == ctypes.sizeof(slam3d_amd.api.MapProfile)


def test_default_params_match_reference_defaults():
    import slam3d_amd
    p = slam3d_amd.default_params()      # RegistrationParameters.hpp:36-97
    assert (p.registration_algorithm, p.point_cloud_density, p.max_fitness_score) == (1, 0.2, 2.0)
    assert (p.max_translation, p.max_rotation, p.euclidean_fitness_epsilon) == (1.0, 1.0, 1.0)
    assert (p.transformation_epsilon, p.max_correspondence_distance, p.maximum_iterations) == (1e-5, 2.5, 50)
    assert (p.rotation_epsilon, p.correspondence_randomness, p.maximum_optimizer_iterations) == (2e-3, 20, 20)
    assert (p.resolution, p.step_size, p.outlier_ratio) == (1.0, 0.05, 0.35)


def test_no_cpu_fallback():
    """Without a HIP device the product refuses to run (it must never route to a CPU path)."""
    import torch
    import slam3d_amd
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(slam3d_amd.BackendError):
        slam3d_amd.Context(0)
    with pytest.raises(slam3d_amd.BackendError):
        slam3d_amd.backend_info(0)
    with pytest.raises(slam3d_amd.BackendError):
        slam3d_amd.Sweep([0])            # the multi-GPU sweep has no CPU path either
    with pytest.raises(slam3d_amd.BackendError):
        slam3d_amd.Sweep()


def test_sweep_shard_arithmetic_matches_the_python_helper():
    """s3d_sweep_shard_range (the C ABI's block arithmetic) == slam3d_amd.sweep.shard_range (what bench.py and the
    gloo test use): contiguous blocks of ceil(n / ranks), short or empty at the end."""
    import slam3d_amd
    from slam3d_amd import sweep
    L = slam3d_amd.load_library()
    lo, hi = ctypes.c_int(), ctypes.c_int()
    for n in (0, 1, 7, 8, 9, 512, 4096, 4097):
        for world in (1, 2, 3, 8):
            covered = []
            for r in range(world):
                L.s3d_sweep_shard_range(n, world, r, ctypes.byref(lo), ctypes.byref(hi))
                assert (lo.value, hi.value) == sweep.shard_range(n, r, world)
                covered += list(range(lo.value, hi.value))
            assert covered == list(range(n))


def test_product_does_not_reference_the_oracle():
    """oracle/ is test infrastructure: nothing under slam3d_amd/, cpp/ or include/ may mention it."""
    bad = []
    for base in ("slam3d_amd", "cpp", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".h", ".hpp", ".hip", ".cpp", ".c")) or fn == "Makefile":
                    text = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"(import\s+oracle|from\s+oracle|s3d_oracle|libs3d_oracle|s3o_)", text):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_product_reads_no_environment_on_the_registration_path():
    """Round 4: behaviour switches come through s3d_exec_options.debug_flags, never through getenv (the library is
    entered from two threads of the host application, ScanSensor.cpp:209-210).  One read-once debug print may remain."""
    import slam3d_amd
    n = 0
    for fn in os.listdir(os.path.join(ROOT, "slam3d_amd", "csrc")):
        if fn.endswith((".h", ".hip")):
            text = open(os.path.join(ROOT, "slam3d_amd", "csrc", fn)).read()
            text = re.sub(r"//.*", "", text)
            n += len(re.findall(r"\bgetenv\s*\(", text))
    assert n <= 1, n
    # the link policy and the options struct of the binding are the header's
    src = 'int main(){return 0;}'
    assert ctypes.sizeof(slam3d_amd.api.LinkPolicyC) == 20 and ctypes.sizeof(slam3d_amd.ExecOptions) == 32


def test_bulk_hand_over_rejects_bad_arguments_before_touching_the_device():
    """s3d_cloud_upload_many validates its arguments first (no context, a negative count, a stride below 3, a null
    array for a non-empty cloud): S3D_STATUS_INVALID_ARGUMENT, nothing allocated - callable without a GPU."""
    import numpy as np
    import slam3d_amd
    L = slam3d_amd.load_library()
    C = ctypes
    pts = np.zeros((4, 3), np.float32)
    ptrs = (C.POINTER(C.c_float) * 1)(pts.ctypes.data_as(C.POINTER(C.c_float)))
    ns = (C.c_int * 1)(4)
    out = (C.c_void_p * 1)()
    assert L.s3d_cloud_upload_many(None, 1, ptrs, ns, 3, out) == 7           # no context
    fake = C.c_void_p(1)      # (never dereferenced: every check below fails before the context is used)
    assert L.s3d_cloud_upload_many(fake, -1, ptrs, ns, 3, out) == 7
    assert L.s3d_cloud_upload_many(fake, 1, ptrs, ns, 2, out) == 7           # stride < 3
    assert L.s3d_cloud_upload_many(fake, 1, None, ns, 3, out) == 7
    null = (C.POINTER(C.c_float) * 1)()
    assert L.s3d_cloud_upload_many(fake, 1, null, ns, 3, out) == 7           # a null array for 4 points
    neg = (C.c_int * 1)(-3)
    assert L.s3d_cloud_upload_many(fake, 1, ptrs, neg, 3, out) == 7
    assert L.s3d_cloud_upload_many(fake, 0, None, None, 3, None) == 0        # nothing to do
