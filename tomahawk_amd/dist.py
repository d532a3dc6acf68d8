"""Multi-GPU plumbing for the sharded all-vs-all run: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The data path needs no collective (every rank owns a band of rows of the pair triangle and holds
all planes); the only exchange is the final gather of the surviving records to the writer rank,
the multi-GPU counterpart of the reference's per-thread flush into one shared writer
(lib/ld/ld_engine.cpp:1742-1802).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .hip import RECORD_DTYPE


def gather_records(recs: np.ndarray, dst: int = 0, device: torch.device | None = None):
    """Gather variable-length RECORD_DTYPE arrays to rank `dst`.

    all_gather of the counts (8 B per rank), then one gather of byte payloads padded to the
    largest count.  Returns the concatenation (rank order) on `dst`, None elsewhere.
    """
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    recs = np.ascontiguousarray(recs, dtype=RECORD_DTYPE)
    if world == 1:
        return recs
    device = device or torch.device("cpu")
    cnt = torch.tensor([len(recs)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt)
    counts = [int(c.item()) for c in counts]
    maxc = max(counts)
    if maxc == 0:
        return np.zeros(0, dtype=RECORD_DTYPE) if rank == dst else None
    item = RECORD_DTYPE.itemsize
    payload = torch.zeros(maxc * item, dtype=torch.uint8, device=device)
    if len(recs):
        payload[: len(recs) * item] = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).to(device)
    gl = [torch.empty_like(payload) for _ in range(world)] if rank == dst else None
    dist.gather(payload, gl, dst=dst)
    if rank != dst:
        return None
    parts = [g[: c * item].cpu().numpy().view(RECORD_DTYPE) for g, c in zip(gl, counts) if c]
    return np.concatenate(parts) if parts else np.zeros(0, dtype=RECORD_DTYPE)
