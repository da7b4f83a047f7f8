"""Multi-GPU sharding of independent scan pairs + the one collective of the path.

Pairs are independent units: global pair index i is owned by rank i mod world (round-robin, as BASELINE.json
config 4 states).  Each rank registers its own pairs on its own GPU; the only exchange is ONE all-gather of
fixed 96-byte pose records {final[16] f32, score, iterations, converged, pair_id, pad[4]} per step --
RCCL over xGMI when the backend is "nccl", gloo in the CPU tests.  At <= 100 B per pair the message is
latency-bound; ring vs tree is irrelevant at this size.
"""
from __future__ import annotations

import numpy as np
import torch

REC_FLOATS = 24          # 96 bytes


def shard_pairs(n_total: int, rank: int, world: int) -> list[int]:
    """Global pair indices owned by `rank` (round-robin)."""
    return list(range(rank, n_total, world))


def pack_records(finals: np.ndarray, scores, iterations, converged, pair_ids, capacity: int | None = None) -> torch.Tensor:
    """[capacity, 24] float32 CPU tensor; unused rows carry pair_id = -1.  finals: [n,16] column-major or [n,4,4]."""
    n = len(pair_ids)
    cap = n if capacity is None else capacity
    rec = torch.zeros(cap, REC_FLOATS, dtype=torch.float32)
    rec[:, 19] = -1.0
    if n:
        f = np.asarray(finals, np.float32)
        if f.ndim == 3:
            f = np.transpose(f, (0, 2, 1)).reshape(n, 16)
        r = rec.numpy()
        r[:n, :16] = f
        r[:n, 16] = np.asarray(scores, np.float32)
        r[:n, 17] = np.asarray(iterations, np.float32)
        r[:n, 18] = np.asarray(converged, np.float32)
        r[:n, 19] = np.asarray(pair_ids, np.float32)
    return rec


def gather_records(rec: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """All-gather the per-rank record blocks (same shape on every rank).  Returns [world*cap, 24] on rec.device."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return rec.clone()
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
    g = gathered.detach().cpu().numpy()
    out = {}
    for row in g:
        pid = int(round(float(row[19])))
        if pid < 0:
            continue
        if pid in out:
            raise ValueError(f"pair {pid} gathered twice")
        out[pid] = dict(final=row[:16].reshape(4, 4).T.copy(), score=float(row[16]), iterations=int(round(float(row[17]))),
                        converged=bool(round(float(row[18]))))
    return out
