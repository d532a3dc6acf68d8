"""Multi-GPU plumbing for the sharded all-vs-all run: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The data path needs no collective (every rank owns a band of rows of the pair triangle and holds
all planes); the only exchange is the final gather of the surviving records to the writer rank,
the multi-GPU counterpart of the reference's per-thread flush into one shared writer
(lib/ld/ld_engine.cpp:1742-1802).  Message shapes: an all_gather of one int64 count per
rank, then count_r x 104 bytes from every rank r with survivors straight into rank dst's buffer -
grouped point-to-point transfers (one ncclGroup of send/recv over the direct xGMI links), exact
sizes, no padding to the largest rank.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .hip import RECORD_DTYPE


def gather_records(recs: np.ndarray, dst: int = 0, device: torch.device | None = None):
    """Gather variable-length RECORD_DTYPE arrays to rank `dst`.

    Returns the concatenation in rank order on `dst`, None elsewhere.
    """
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    recs = np.ascontiguousarray(recs, dtype=RECORD_DTYPE)
    if world == 1:
        return recs
    device = device or torch.device("cpu")
    item = RECORD_DTYPE.itemsize
    cnt = torch.tensor([len(recs)], dtype=torch.int64, device=device)
    gathered = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(gathered, cnt)
    counts = [int(c.item()) for c in gathered]
    total = sum(counts)
    if total == 0:
        return np.zeros(0, dtype=RECORD_DTYPE) if rank == dst else None

    ops, out = [], None
    if rank == dst:
        # one receive buffer for everything, every rank's slice at its final place
        out = torch.empty(total * item, dtype=torch.uint8, device=device)
        off = 0
        for r, c in enumerate(counts):
            n = c * item
            if c and r != dst:
                ops.append(dist.P2POp(dist.irecv, out[off:off + n], r))
            elif c:
                out[off:off + n] = torch.from_numpy(recs.view(np.uint8).reshape(-1)).to(device)
            off += n
    elif len(recs):
        payload = torch.from_numpy(recs.view(np.uint8).reshape(-1)).to(device)
        ops.append(dist.P2POp(dist.isend, payload, dst))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != dst:
        return None
    return out.cpu().numpy().view(RECORD_DTYPE)
