"""GPU tests of the cross-call pre-pass cache (s3d_exec_options.cache_prepass, include/slam3d_hip.h).

The reference re-filters both clouds in every align() (PointCloudSensor.cpp:127-131) although a mapper registers every
scan several times (ScanSensor.cpp:113 against the previous scan, :179-201 against its neighbours).  With the option
on, the voxel filter / search grid / k-NN normals of a device-resident cloud are kept in HBM per
(cloud, point_cloud_density, grid budget, correspondence_randomness).  The bar: results are identical bit for bit
with and without it, in every mix of cached and new clouds; entries die with their cloud."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _opts(s3d, cache, **kw):
    return s3d.ExecOptions(cache_prepass=1 if cache else 0, **kw)


def test_cached_equals_uncached_bit_for_bit(gpu_ctx, fixture_clouds):
    import slam3d_amd as s3d
    gpu_ctx.cache_control(clear=True)
    base = gpu_ctx.cache_control()
    cl = [gpu_ctx.upload(c) for c in fixture_clouds]
    try:
        chain = [(0, 1), (1, 2), (2, 3)]
        for alg in (s3d.ALG_GICP, s3d.ALG_ICP, s3d.ALG_NDT):
            p = s3d.default_params(registration_algorithm=alg, maximum_iterations=12)
            want = gpu_ctx.align_batch([cl[a] for a, _ in chain], [cl[b] for _, b in chain], None, p, _opts(s3d, False))
            # the mapper pattern: one new scan per call, registered against the previous one
            got = [gpu_ctx.align_batch([cl[a]], [cl[b]], None, p, _opts(s3d, True))[0] for a, b in chain]
            assert np.array_equal(np.array(got), want), alg
            # everything cached now: a batch of hits only, and a mix of cached clouds with a new one
            again = gpu_ctx.align_batch([cl[a] for a, _ in chain], [cl[b] for _, b in chain], None, p, _opts(s3d, True))
            assert np.array_equal(again, want), alg
            fresh = gpu_ctx.upload(fixture_clouds[3])
            mixed = gpu_ctx.align_batch([cl[2], cl[0]], [fresh, cl[1]], None, p, _opts(s3d, True))
            fresh.release()
            assert np.array_equal(mixed[0], want[2]) and np.array_equal(mixed[1], want[0]), alg
        st = gpu_ctx.cache_control()
        assert st["hits"] > base["hits"] and st["entries"] >= 4 and st["bytes"] > 0
        # the single-pair entry points on handles, coarse + fine (two densities -> two entries per cloud)
        fine = s3d.default_params(registration_algorithm=s3d.ALG_ICP)
        coarse = s3d.default_params(registration_algorithm=s3d.ALG_ICP, point_cloud_density=0.5,
                                    max_correspondence_distance=5.0, max_translation=3.0)
        ident = np.eye(4)
        ref = gpu_ctx.create_constraint_clouds(cl[0], ident, cl[3], ident, ident, True, fine, coarse, 1.0, _opts(s3d, False))
        for _ in range(2):
            got = gpu_ctx.create_constraint_clouds(cl[0], ident, cl[3], ident, ident, True, fine, coarse, 1.0, _opts(s3d, True))
            assert got[0] == ref[0] == 0 and np.array_equal(got[1], ref[1]) and got[3] == ref[3]
        # normals computed with another k are not reused
        p7 = s3d.default_params(correspondence_randomness=7, maximum_iterations=8)
        w7 = gpu_ctx.align_batch([cl[0]], [cl[1]], None, p7, _opts(s3d, False))
        assert np.array_equal(gpu_ctx.align_batch([cl[0]], [cl[1]], None, p7, _opts(s3d, True)), w7)
        assert np.array_equal(gpu_ctx.align_batch([cl[0]], [cl[1]], None, p7, _opts(s3d, True)), w7)
    finally:
        for c in cl:
            c.release()
    assert gpu_ctx.cache_control()["entries"] == 0        # released clouds take their entries with them


def test_cache_budget_and_eviction(gpu_ctx, fixture_clouds):
    import slam3d_amd as s3d
    gpu_ctx.cache_control(clear=True)
    cl = [gpu_ctx.upload(c) for c in fixture_clouds]
    try:
        p = s3d.default_params(registration_algorithm=s3d.ALG_ICP, maximum_iterations=5)
        want = gpu_ctx.align_batch([cl[0], cl[2]], [cl[1], cl[3]], None, p, _opts(s3d, False))
        gpu_ctx.align_batch([cl[0]], [cl[1]], None, p, _opts(s3d, True))
        one = gpu_ctx.cache_control()
        assert one["entries"] == 2
        per_entry = one["bytes"] // 2
        # room for about three entries: the fourth cloud pushes the least recently used one out
        gpu_ctx.cache_control(limit_bytes=int(3.5 * per_entry))
        got = gpu_ctx.align_batch([cl[2]], [cl[3]], None, p, _opts(s3d, True))
        st = gpu_ctx.cache_control()
        assert st["entries"] == 3 and st["bytes"] <= int(3.5 * per_entry)
        assert np.array_equal(got[0], want[1])
        assert np.array_equal(gpu_ctx.align_batch([cl[0]], [cl[1]], None, p, _opts(s3d, True))[0], want[0])
        # a budget below one entry: nothing is kept, results unchanged
        gpu_ctx.cache_control(limit_bytes=1024, clear=True)
        assert np.array_equal(gpu_ctx.align_batch([cl[0], cl[2]], [cl[1], cl[3]], None, p, _opts(s3d, True)), want)
        assert gpu_ctx.cache_control()["entries"] == 0
    finally:
        gpu_ctx.cache_control(limit_bytes=16 << 30, clear=True)
        for c in cl:
            c.release()


def test_host_buffer_entry_points_do_not_populate_the_cache(gpu_ctx, fixture_clouds):
    """s3d_align on host buffers uploads temporary clouds: nothing to keep."""
    import slam3d_amd as s3d
    gpu_ctx.cache_control(clear=True)
    p = s3d.default_params(registration_algorithm=s3d.ALG_ICP, maximum_iterations=5)
    gpu_ctx.align(fixture_clouds[0], fixture_clouds[1], np.eye(4), p, _opts(s3d, True))
    assert gpu_ctx.cache_control()["entries"] == 0


def test_checkpoint_round_trip_of_the_cached_products(gpu_ctx, fixture_clouds):
    """s3d_cloud_cache_export / _import: what a checkpoint keeps next to a measurement's .s3dm file
    (GraphSerialization.cpp:14-66 / :68-135).  A reloaded measurement is a NEW cloud handle; with its blob imported
    the first registration is served from the cache (no misses) and gives the uncached result bit for bit."""
    import slam3d_amd as s3d
    gpu_ctx.cache_control(clear=True)
    cl = [gpu_ctx.upload(c) for c in fixture_clouds[:3]]
    p = s3d.default_params(registration_algorithm=s3d.ALG_GICP, maximum_iterations=10)
    coarse = s3d.default_params(registration_algorithm=s3d.ALG_ICP, point_cloud_density=0.5, maximum_iterations=6)
    want = gpu_ctx.align_batch([cl[0], cl[1]], [cl[1], cl[2]], None, p, _opts(s3d, False))
    want_c = gpu_ctx.align_batch([cl[0]], [cl[1]], None, coarse, _opts(s3d, False))
    assert gpu_ctx.cache_export(cl[0]) == b""                         # nothing cached yet
    gpu_ctx.align_batch([cl[0], cl[1]], [cl[1], cl[2]], None, p, _opts(s3d, True))
    gpu_ctx.align_batch([cl[0]], [cl[1]], None, coarse, _opts(s3d, True))   # a second entry (another density) for 0 and 1
    blobs = [gpu_ctx.cache_export(c) for c in cl]
    assert all(len(b) > 0 for b in blobs) and len(blobs[0]) > len(blobs[2])
    assert gpu_ctx.cache_export(cl[0]) == blobs[0]                    # deterministic bytes
    for c in cl:
        c.release()
    assert gpu_ctx.cache_control()["entries"] == 0
    # "fromFolder": new handles for the same points, blobs handed back
    re = [gpu_ctx.upload(c) for c in fixture_clouds[:3]]
    try:
        for c, b in zip(re, blobs):
            assert gpu_ctx.cache_import(c, b) == 0
        st0 = gpu_ctx.cache_control()
        assert st0["entries"] == 5
        got = gpu_ctx.align_batch([re[0], re[1]], [re[1], re[2]], None, p, _opts(s3d, True))
        got_c = gpu_ctx.align_batch([re[0]], [re[1]], None, coarse, _opts(s3d, True))
        st1 = gpu_ctx.cache_control()
        assert np.array_equal(got, want) and np.array_equal(got_c, want_c)
        assert st1["misses"] == st0["misses"] and st1["hits"] == st0["hits"] + 5
        # a blob only fits the cloud it was made from; a damaged one is refused; nothing is installed in either case
        other = gpu_ctx.upload(fixture_clouds[3])
        try:
            assert gpu_ctx.cache_import(other, blobs[0]) == 7 and "different point cloud" in gpu_ctx.last_error()
            moved = gpu_ctx.upload(fixture_clouds[0][:, :3] + np.float32(1e-3))
            assert gpu_ctx.cache_import(moved, blobs[0]) == 7
            moved.release()
            assert gpu_ctx.cache_import(other, blobs[0][:100]) == 7
            bad = bytearray(gpu_ctx.cache_export(re[2])); bad[40] ^= 0xFF      # an entry header field
            assert gpu_ctx.cache_import(re[2], bytes(bad[:len(bad) // 2])) == 7
            # bit rot INSIDE a payload with every header intact: the per-entry checksum refuses it - a
            # flipped cell-table or index word would otherwise send the kernels out of bounds
            good = gpu_ctx.cache_export(re[2])
            for at in (len(good) // 2, len(good) - 8, 200):
                rot = bytearray(good); rot[at] ^= 0x10
                assert gpu_ctx.cache_import(re[2], bytes(rot)) == 7 and "damaged payload" in gpu_ctx.last_error(), at
            assert gpu_ctx.cache_import(re[2], good) == 0
            assert gpu_ctx.cache_control()["entries"] == 5
        finally:
            other.release()
        # importing over existing entries replaces them
        assert gpu_ctx.cache_import(re[0], blobs[0]) == 0 and gpu_ctx.cache_control()["entries"] == 5
        assert np.array_equal(gpu_ctx.align_batch([re[0]], [re[1]], None, coarse, _opts(s3d, True)), want_c)
    finally:
        for c in re:
            c.release()


def test_import_accepts_the_previous_blob_version(gpu_ctx, fixture_clouds):
    """Blob version 3 added `layout` and the fused grid to every entry; a checkpoint written by the revision before
    (version 2: the same entry without the two fields, always the two-sort layout) must still load.  The
    version-2 bytes are made here from a version-3 export of two-sort entries (NDT registrations take that path)."""
    import struct
    import slam3d_amd as s3d
    gpu_ctx.cache_control(clear=True)
    cl = [gpu_ctx.upload(c) for c in fixture_clouds[:2]]
    p = s3d.default_params(registration_algorithm=s3d.ALG_NDT, maximum_iterations=8)
    try:
        want = gpu_ctx.align_batch([cl[0]], [cl[1]], None, p, _opts(s3d, False))
        gpu_ctx.align_batch([cl[0]], [cl[1]], None, p, _opts(s3d, True))
        blobs3 = [gpu_ctx.cache_export(c) for c in cl]
        blobs2 = []
        for b in blobs3:
            magic, version, entries, n_raw, ph = struct.unpack_from("<IIIIQ", b, 0)
            assert (magic, version, entries) == (0x42443353, 3, 1)
            e = b[24:24 + 176]                                  # BlobEntry (s3d_api.hip): 124 bytes of common fields,
            assert struct.unpack_from("<I", e, 0)[0] == 0x45443353 and struct.unpack_from("<i", e, 124)[0] == 0   # layout 0
            nbytes, _ = struct.unpack_from("<QQ", e, 160)       # layout + fz (36 bytes, padded to 160), then size and hash
            assert 24 + 176 + nbytes == len(b)
            blobs2.append(struct.pack("<IIIIQ", magic, 2, entries, n_raw, ph) + e[:124] + b"\0" * 4 + e[160:176] + b[24 + 176:])
        gpu_ctx.cache_control(clear=True)
        assert gpu_ctx.cache_control()["entries"] == 0
        for c, b in zip(cl, blobs2):
            assert gpu_ctx.cache_import(c, b) == 0, gpu_ctx.last_error()
        st0 = gpu_ctx.cache_control()
        assert st0["entries"] == 2
        got = gpu_ctx.align_batch([cl[0]], [cl[1]], None, p, _opts(s3d, True))
        st1 = gpu_ctx.cache_control()
        assert np.array_equal(got, want) and st1["misses"] == st0["misses"] and st1["hits"] == st0["hits"] + 2
        # and what is exported from the imported entries is version 3 again, byte for byte
        assert [gpu_ctx.cache_export(c) for c in cl] == blobs3
        assert gpu_ctx.cache_import(cl[0], blobs2[0][:150]) == 7         # truncated inside a version-2 entry header
        # an unknown version is refused
        assert gpu_ctx.cache_import(cl[0], blobs3[0][:4] + struct.pack("<I", 9) + blobs3[0][8:]) == 7
    finally:
        for c in cl:
            c.release()
        gpu_ctx.cache_control(clear=True)


def test_cpp_mirror_device_cache_across_a_checkpoint(fixture_clouds, tmp_path):
    """cpp/example_checkpoint.cpp: PointCloudSensor::saveDeviceCache / loadDeviceCache around a save + reload of two
    measurements (new objects carrying the stored uuids, GraphSerialization.cpp:40-47 / :68-135): the edge after the
    reload equals the one before bit for bit and is served from the restored cache without a miss."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "cpp", "example_checkpoint")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "cpp")])
    files = []
    for i in (0, 1):
        f = tmp_path / ("scan%d.bin" % i)
        fixture_clouds[i].astype(np.float32).tofile(f)
        files.append(str(f))
    out = subprocess.check_output([exe, str(tmp_path)] + files, stderr=subprocess.DEVNULL).decode().splitlines()
    get = lambda key: [l[len(key) + 1:] for l in out if l.startswith(key + " ")]
    assert get("save before any registration") == ["0"]
    assert get("save 0") == ["1"] and get("save 1") == ["1"]
    assert get("entries after release") == ["0"]
    assert get("uuid kept") == ["1", "1"]
    assert get("preload") == ["2 0"]      # PointCloudSensor::preloadDeviceClouds: one bulk hand-over, duplicates / repeats skipped
    assert get("load wrong scan") == ["0"] and get("load missing file") == ["0"]
    assert get("load 0") == ["1"] and get("load 1") == ["1"]
    assert len(get("first")) == 1 and get("first") == get("again")
    assert get("entries")[-1] == "2 new hits 2 new misses 0"
    assert os.path.getsize(tmp_path / "0.s3dc") > 16 * len(fixture_clouds[0]) // 8
