"""Host-side arithmetic of bench.py that needs no GPU: the per-rank share of the host's CPUs (eight
ranks on a 16-CPU quota must not oversubscribe the host into the timed region)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_ranks_share_the_usable_cpus():
    import bench
    for usable in (1, 2, 8, 16, 64, 256):
        for world in (1, 2, 4, 8):
            t = bench._host_threads(usable, world)
            assert 1 <= t <= 16
            assert t * world <= max(usable, world), (usable, world, t)
    assert bench._host_threads(16, 8) == 2 and bench._host_threads(16, 1) == 16 and bench._host_threads(256, 8) == 16


def test_usable_cpus_respects_affinity_and_quota():
    import bench
    n, quota = bench._usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    assert quota is None or quota > 0
    if quota is not None:
        assert n <= max(1, int(quota + 0.5))
