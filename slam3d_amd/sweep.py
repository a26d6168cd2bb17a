"""Pair-sharded loop-closure sweep over the GPUs of one node.

Candidate scan pairs (ScanSensor::linkToNeighbors, ScanSensor.cpp:179-201) are independent: each
rank registers a contiguous block of them on its own GPU with no data-path collective, then ONE
all-gather of the fixed-size 128-byte edge records (16 doubles, include/slam3d_registration_types.h)
returns every edge to every rank in pair order.  Backend "nccl" is RCCL over xGMI on ROCm; the CPU
tests run the same code over gloo.
"""
import numpy as np
import torch
import torch.distributed as dist

from .api import EDGE_RECORD_DOUBLES


def shard_range(n_pairs, rank, world):
    """Contiguous block [lo, hi) of rank `rank`: ceil(n/world) pairs per rank, last ranks may be short."""
    per = (n_pairs + world - 1) // world
    lo = min(rank * per, n_pairs)
    return lo, min(lo + per, n_pairs)


def all_gather_records(local_records, n_pairs, device=None):
    """local_records: (hi-lo, 16) float64 numpy of this rank's block.  Returns (n_pairs, 16) in pair order."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return np.asarray(local_records, np.float64).reshape(-1, EDGE_RECORD_DOUBLES)[:n_pairs]
    rank = dist.get_rank()
    per = (n_pairs + world - 1) // world
    lo, hi = shard_range(n_pairs, rank, world)
    buf = torch.zeros((per, EDGE_RECORD_DOUBLES), dtype=torch.float64)
    if hi > lo:
        buf[:hi - lo] = torch.from_numpy(np.ascontiguousarray(local_records, np.float64).reshape(hi - lo, -1))
    if device is not None:
        buf = buf.to(device)
    out = torch.empty((world * per, EDGE_RECORD_DOUBLES), dtype=torch.float64, device=buf.device)
    dist.all_gather_into_tensor(out, buf)
    return out.cpu().numpy()[:n_pairs] if per * world == n_pairs else _compact(out.cpu().numpy(), n_pairs, per, world)


def _compact(arr, n_pairs, per, world):
    rows = []
    for r in range(world):
        lo, hi = shard_range(n_pairs, r, world)
        rows.append(arr[r * per:r * per + (hi - lo)])
    return np.concatenate(rows, 0)


def loop_closure_sweep(register_block, n_pairs, device=None):
    """register_block(lo, hi) -> (hi-lo, 16) records of pairs lo..hi-1 on this rank's GPU.
    Returns all n_pairs records on every rank, in pair order (deterministic insertion order for the graph)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(n_pairs, rank, world)
    local = register_block(lo, hi) if hi > lo else np.zeros((0, EDGE_RECORD_DOUBLES))
    return all_gather_records(local, n_pairs, device)
