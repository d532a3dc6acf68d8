"""Multi-GPU plumbing for the sharded all-vs-all run: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The data path needs no collective (every rank owns a band of rows of the pair triangle and holds
all planes); the only exchange is the final gather of the surviving records to the writer rank,
the multi-GPU counterpart of the reference's per-thread flush into one shared writer
(lib/ld/ld_engine.cpp:1742-1802).  Message shapes: an all_gather of one int64 count per
rank, then count_r x 104 bytes from every rank r with survivors straight into rank dst's buffer -
grouped point-to-point transfers (one ncclGroup of send/recv over the direct xGMI links), exact
sizes, no padding to the largest rank.  Over RCCL the payload is the engine's own HBM buffer
(twk_hip_set_device_sink / HipLd.device_records_tensor): it goes GPU to GPU and only the writer rank
copies it to host memory, once.
"""
from __future__ import annotations

import datetime
import os

import numpy as np
import torch
import torch.distributed as dist

from .hip import RECORD_DTYPE

ITEM = RECORD_DTYPE.itemsize


def window_slab(n_variants: int, window_variants: int, part: int, n_parts: int):
    """Row band of shard `part` of a windowed run over variants at equal spacing, where every variant pairs with the
    `window_variants` variants after it: bands hold equal numbers of in-window pairs, boundaries on multiples of 64
    variants (the rule of twk_hip_ld_region's window mode), derived from the positions alone by every rank.
    -> (r0, r1, col_end, pairs): rows [r0, r1), the slab a rank loads is [r0, col_end) (its band plus the halo its
    window reaches), and the in-window pairs of the band.  Replaces the reference's square chunks
    (lib/ld/ld_balancing.h:23-80) for windowed multi-GPU runs."""
    cost = np.minimum(window_variants, n_variants - 1 - np.arange(n_variants, dtype=np.int64))
    cum = np.concatenate(([0], np.cumsum(cost)))

    def boundary(k):
        if k <= 0:
            return 0
        if k >= n_parts:
            return n_variants
        r = int(np.searchsorted(cum, cum[-1] * k // n_parts, side="left"))
        return min(n_variants, (r + 32) // 64 * 64)

    r0, r1 = boundary(part), boundary(part + 1)
    return r0, r1, min(n_variants, r1 + window_variants), int(cum[r1] - cum[r0])


def window_total_pairs(n_variants: int, window_variants: int) -> int:
    return int(np.minimum(window_variants, n_variants - 1 - np.arange(n_variants, dtype=np.int64)).sum())


def init_groups(backend: str, device: torch.device | None, force_rccl_failure: bool = False, timeout_s: int = 300):
    """Bring up torch.distributed for the bench: a gloo group for control traffic (barriers, the statistics
    all-reduce, the agreement below) and, for backend "nccl", an RCCL group for the gather of the survivors.

    Whether RCCL is usable is *agreed* before anyone depends on it: every rank tries to create the RCCL group and to
    run one small gather + barrier on it, then the outcomes are all-reduced over gloo.  Only if every rank succeeded is
    the RCCL group used; if every rank failed the gather runs over gloo (reported in the JSON line); ranks never end
    up in different backends.  A rank that hangs inside RCCL while others failed is ended by the group's timeout,
    which exits the job non-zero - an asymmetric failure is an error, not a fallback.
    -> (gather_group, tensor_device, description)"""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")       # one node: the hostname may not resolve, loopback always does
    dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=max(timeout_s, 600)))
    cpu = torch.device("cpu")
    if backend != "nccl":
        gather_records(np.zeros(1, dtype=RECORD_DTYPE), dst=0, device=cpu)
        dist.barrier()
        return None, cpu, "gloo"
    ok, err, group = 1, "", None
    try:
        if force_rccl_failure:          # test hook
            raise RuntimeError("forced by TWK_BENCH_FORCE_RCCL_FAILURE")
        group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=timeout_s), device_id=device)
    except Exception as e:              # noqa: BLE001 - whatever RCCL throws, the outcome is agreed below
        ok, err = 0, repr(e)
    flag = torch.tensor([ok], dtype=torch.int64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        # rings and the point-to-point channels of the gather are set up lazily on first use: once here, outside any step
        try:
            gather_records(np.zeros(1, dtype=RECORD_DTYPE), dst=0, device=device, group=group)
            dist.barrier(group=group)
        except Exception as e:          # noqa: BLE001
            ok, err = 0, repr(e)
        flag = torch.tensor([ok], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        return group, device, "nccl"
    # at least one rank could not use RCCL: everyone gathers over gloo, and says why
    msgs = [None] * dist.get_world_size()
    dist.all_gather_object(msgs, err)
    why = next((m for m in msgs if m), "unknown")
    gather_records(np.zeros(1, dtype=RECORD_DTYPE), dst=0, device=cpu)
    dist.barrier()
    return None, cpu, f"gloo (RCCL failed to initialise: {why[:120]})"


def gather_records(recs, dst: int = 0, device: torch.device | None = None, group=None, to_host: bool = True):
    """Gather variable-length arrays of 104-byte records to rank `dst`.

    recs: a RECORD_DTYPE numpy array, or a torch uint8 tensor of n x 104 bytes (HipLd.device_records_tensor: the
    engine's HBM buffer, sent as it is when `device` is that GPU).  `device` is where the transfer buffers live
    (the GPU for RCCL, the CPU for gloo).  Returns the concatenation in rank order on `dst` - a RECORD_DTYPE array,
    or with to_host=False the uint8 tensor on `device` - and None elsewhere.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    device = device or torch.device("cpu")
    if isinstance(recs, torch.Tensor):
        payload = recs.reshape(-1)
        assert payload.dtype == torch.uint8 and payload.numel() % ITEM == 0
        if payload.device != device:
            payload = payload.to(device)
    else:
        recs = np.ascontiguousarray(recs, dtype=RECORD_DTYPE)
        payload = torch.from_numpy(recs.view(np.uint8).reshape(-1))
        if device.type != "cpu":
            payload = payload.to(device)
    n_mine = payload.numel() // ITEM

    def deliver(t):
        if not to_host:
            return t
        return t.cpu().numpy().view(RECORD_DTYPE)

    if world == 1:
        return deliver(payload)
    cnt = torch.tensor([n_mine], dtype=torch.int64, device=device)
    gathered = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(gathered, cnt, group=group)
    counts = [int(c.item()) for c in gathered]
    total = sum(counts)
    if total == 0:
        return deliver(torch.empty(0, dtype=torch.uint8, device=device)) if rank == dst else None

    ops, out = [], None
    if rank == dst:
        # one receive buffer for everything, every rank's slice at its final place
        out = torch.empty(total * ITEM, dtype=torch.uint8, device=device)
        off = 0
        for r, c in enumerate(counts):
            n = c * ITEM
            if c and r != dst:
                ops.append(dist.P2POp(dist.irecv, out[off:off + n], r, group=group))
            elif c:
                out[off:off + n] = payload
            off += n
    elif n_mine:
        ops.append(dist.P2POp(dist.isend, payload.contiguous(), dst, group=group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != dst:
        return None
    return deliver(out)
