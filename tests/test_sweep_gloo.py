"""N>1 path on CPU: the pair-sharded sweep + all-gather of edge records over gloo, world_size 2."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_records(lo, hi):
    """Stand-in for the GPU registration: a record that encodes its pair id (no GPU in this test)."""
    rec = np.zeros((hi - lo, 16))
    for i, p in enumerate(range(lo, hi)):
        rec[i, :12] = np.eye(4)[:3, :4].T.reshape(-1)
        rec[i, 9] = 0.1 * p            # tx
        rec[i, 12] = 1.0 / (p + 1)     # fitness
        rec[i, 13] = 20
        rec[i, 14] = 1000 + p
        rec[i, 15] = p % 5             # status
    return rec


def _worker(rank, world, port, n_pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from slam3d_amd.sweep import loop_closure_sweep, shard_range
    lo, hi = shard_range(n_pairs, rank, world)
    out = loop_closure_sweep(_fake_records, n_pairs)
    q.put((rank, lo, hi, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [8, 7, 1])
def test_sweep_world2_gloo(n_pairs):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = _fake_records(0, n_pairs)
    covered = []
    for rank, lo, hi, out in results:
        assert out.shape == (n_pairs, 16)
        assert np.array_equal(out, expect)          # every rank holds every edge, in pair order
        covered += list(range(lo, hi))
    assert sorted(covered) == list(range(n_pairs))  # blocks tile the pair list exactly once


def test_shard_range_tiles():
    from slam3d_amd.sweep import shard_range
    for n in (0, 1, 5, 4096, 4097):
        for w in (1, 2, 4, 8):
            blocks = [shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
