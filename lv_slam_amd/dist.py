"""Multi-GPU sharding of independent scan pairs + the one collective of the path.

Pairs are independent units: global pair index i is owned by rank i mod world (round-robin, as BASELINE.json
config 4 states).  Each rank registers its own pairs on its own GPU; the only exchange is ONE all-gather of
fixed 96-byte pose records per step -- RCCL over xGMI when the backend is "nccl", gloo in the CPU tests:

    int32 word   0..15   final transform, float32 bit patterns, column-major (Eigen::Matrix4f layout)
                 16      score, float32 bit pattern
                 17      iterations        (int32)
                 18      converged         (int32)
                 19      pair_id           (int32; -1 = padding row of a rank that owns fewer pairs than `capacity`)
                 20..23  pad

The records travel as int32 words (ids and counters are exact, floats keep their bit patterns); on the GPU they are packed
by the engine itself (mi355ndt_batch_pose_records) into the tensor that goes into the all-gather -- no host hop.  At <= 100 B
per pair the message is latency-bound; ring vs tree is irrelevant at this size.
"""
from __future__ import annotations

import numpy as np
import torch

REC_WORDS = 24           # 96 bytes


def shard_pairs(n_total: int, rank: int, world: int) -> list[int]:
    """Global pair indices owned by `rank` (round-robin)."""
    return list(range(rank, n_total, world))


def shard_capacity(n_total: int, world: int) -> int:
    """Records per rank in the gather: the largest shard (ranks that own one pair fewer pad with pair_id = -1)."""
    return (n_total + world - 1) // world


def pack_records(finals: np.ndarray, scores, iterations, converged, pair_ids, capacity: int | None = None) -> torch.Tensor:
    """Host-side packer (CPU tests, tools): [capacity, 24] int32 CPU tensor.  finals: [n,16] column-major or [n,4,4]."""
    n = len(pair_ids)
    cap = n if capacity is None else capacity
    rec = np.zeros((cap, REC_WORDS), np.int32)
    rec[:, 19] = -1
    if n:
        f = np.asarray(finals, np.float32)
        if f.ndim == 3:
            f = np.transpose(f, (0, 2, 1)).reshape(n, 16)
        rec[:n, :16] = np.ascontiguousarray(f).view(np.int32)
        rec[:n, 16] = np.asarray(scores, np.float32).view(np.int32)
        rec[:n, 17] = np.asarray(iterations, np.int32)
        rec[:n, 18] = np.asarray(converged, np.int32)
        rec[:n, 19] = np.asarray(pair_ids, np.int32)
    return torch.from_numpy(rec)


def gather_records(rec: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """All-gather the per-rank record blocks (same shape on every rank).  Returns [world*cap, 24] on rec.device."""
    import torch.distributed as dist
    if not dist.is_initialized():                  # no process group: the block is the result
        if out is None:
            return rec.clone()
        out.copy_(rec)
        return out
    world = dist.get_world_size()                  # (a group of one rank still goes through the collective)
    if out is None:
        out = torch.empty(world * rec.shape[0], rec.shape[1], dtype=rec.dtype, device=rec.device)
    if dist.get_backend() == "gloo":
        parts = list(out.chunk(world, dim=0))
        dist.all_gather(parts, rec.contiguous())
    else:
        dist.all_gather_into_tensor(out, rec.contiguous())
    return out


def unpack_records(gathered: torch.Tensor) -> dict[int, dict]:
    """pair_id -> {final[4,4] f32, score, iterations, converged}; padding rows (pair_id < 0) dropped."""
    g = np.ascontiguousarray(gathered.detach().cpu().numpy()).astype(np.int32, copy=False)
    out = {}
    for row in g:
        pid = int(row[19])
        if pid < 0:
            continue
        if pid in out:
            raise ValueError(f"pair {pid} gathered twice")
        out[pid] = dict(final=row[:16].view(np.float32).reshape(4, 4).T.copy(), score=float(row[16:17].view(np.float32)[0]),
                        iterations=int(row[17]), converged=bool(row[18]))
    return out
