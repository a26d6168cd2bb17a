"""The optional pin against the real PCL (test infrastructure, like the rest of oracle/).

oracle/pcl/pcl_gicp.cpp runs the reference's own call sequence (PointCloudSensor.cpp:52-82, :119-174) around the REAL
pcl::GeneralizedIterativeClosestPoint / pcl::VoxelGrid.  It needs libpcl-dev, which the image this repository is built
in does not have; `status()` says so, `build_and_pin()` compiles it and writes tests/golden/pcl_golden.json where PCL
exists (called by __graft_entry__.build()), `bench()` times one pair through it (bench.py's cpu_baseline_pcl).
No stand-in headers are ever written to make it compile: a host without PCL pins nothing."""
import json
import os
import subprocess
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
_PCL_DIR = os.path.join(_HERE, "pcl")
_EXE = os.path.join(_PCL_DIR, "pcl_gicp")
_GOLDEN = os.path.join(os.path.dirname(_HERE), "tests", "golden", "pcl_golden.json")


def available():
    """True when pkg-config knows a pcl_registration module (pcl_registration or pcl_registration-<version>)."""
    try:
        mods = subprocess.run(["pkg-config", "--list-all"], capture_output=True, text=True, timeout=20).stdout
    except (OSError, subprocess.SubprocessError):
        return False
    return any(line.split()[0].startswith("pcl_registration") for line in mods.splitlines() if line.strip())


def status():
    if not available():
        return "absent"
    return "pinned" if os.path.exists(_GOLDEN) else "present, not pinned yet"


def build_and_pin(verbose=True):
    """`make -C oracle/pcl golden` when PCL is installed; returns status().  Never raises: the pin is optional."""
    if not available():
        if verbose:
            print('[oracle] "pcl": "absent" - oracle stays unpinned (DESIGN.md 5); nothing built')
        return "absent"
    try:
        subprocess.check_call(["make", "-s", "-C", _PCL_DIR, "golden"])
    except (OSError, subprocess.CalledProcessError) as e:
        if verbose:
            print("[oracle] PCL found but oracle/pcl did not build / run: %s" % e)
        return "present, build failed"
    if verbose:
        print("[oracle] PCL found: wrote %s (tests/test_pcl_pin.py now pins the oracle against it)" % _GOLDEN)
    return status()


def bench(src_xyz, tgt_xyz, leaf, iters, reps=3):
    """One pair through real PCL GICP `reps` times; None without the built program."""
    if not os.path.exists(_EXE):
        return None
    import numpy as np
    with tempfile.TemporaryDirectory() as d:
        fs, ft = os.path.join(d, "s.bin"), os.path.join(d, "t.bin")
        np.ascontiguousarray(src_xyz, np.float32)[:, :3].tofile(fs)
        np.ascontiguousarray(tgt_xyz, np.float32)[:, :3].tofile(ft)
        r = subprocess.run([_EXE, "bench", fs, ft, str(leaf), str(int(iters)), str(int(reps))], capture_output=True,
                           text=True)
    if r.returncode != 0:
        return None
    try:
        return json.loads(r.stderr.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return None
