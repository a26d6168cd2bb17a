"""slam3d_amd — MI355X-native scan-pair registration behind the slam3d PointCloudSensor API.

The product is the C-ABI library ``slam3d_amd/lib/libslam3d_hip.so`` (HIP kernels for gfx950,
sources in ``slam3d_amd/csrc``, ABI in ``include/slam3d_hip.h``) and the C++ mirror of the
reference classes under ``cpp/``.  This Python package is a thin ctypes binding used by the
tests and by ``bench.py``; it never computes anything itself and it has no CPU fallback.
"""
from .api import (  # noqa: F401
    ALG_GICP, ALG_GICP_OMP, ALG_ICP, ALG_NDT, ALG_NDT_OMP, STATUS_NAMES, AlignInfo, BackendError, Cloud, Context,
    EdgeRecord, ExecOptions, Profile, RegParams, Sweep, backend_info, build, cu_masks, default_params, lib_path,
    load_library,
    EDGE_RECORD_DOUBLES)
from .synthetic import make_pair, make_scene_cloud  # noqa: F401
from . import api, posegraph  # noqa: F401
