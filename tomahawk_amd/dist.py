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
import time

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


def _physical_device(device) -> str:
    """A name for the GPU behind a torch device that is the same in every process of the node, whatever each process
    was allowed to see: its UUID or PCI address where torch exposes them, else the visible index."""
    if device is None or device.type != "cuda":
        return ""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    try:
        p = torch.cuda.get_device_properties(idx)
        uuid = getattr(p, "uuid", None)
        if uuid is not None and str(uuid).strip("0-"):
            return f"uuid {uuid}"
        if hasattr(p, "pci_bus_id"):
            return f"pci {getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{getattr(p, 'pci_device_id', 0):02x}"
    except Exception:                   # noqa: BLE001
        pass
    return f"visible index {idx} ({os.environ.get('HIP_VISIBLE_DEVICES') or os.environ.get('CUDA_VISIBLE_DEVICES') or 'all'})"


def init_groups(backend: str, device: torch.device | None, force_rccl_failure: bool = False, timeout_s: int = 300):
    """Bring up torch.distributed for the bench: a gloo group for control traffic (barriers, the statistics
    all-reduce, the agreement below) and, for backend "nccl", an RCCL group for the gather of the survivors.

    Whether RCCL is usable is *agreed* before anyone depends on it, in three steps over gloo:
    1. a cheap precondition - every rank names the GPU it sits on, and RCCL is only attempted when no two ranks of the
       node share one (RCCL refuses a duplicate device; found here it costs nothing, found inside `new_group` it can leave
       the other ranks waiting for the group's timeout);
    2. every rank creates the RCCL group and the outcomes are all-reduced;
    3. every rank runs one small gather + barrier on it (rings and point-to-point channels are set up lazily) and the
       outcomes are all-reduced again.
    All ranks succeeded -> RCCL.  All ranks failed alike (or the precondition said no) -> the gather runs over gloo and
    the JSON line says why.  Some succeeded and some failed -> RuntimeError on every rank (non-zero exit): an
    asymmetric failure is an error, not a fallback - the ranks that did not fail have typically sat in RCCL until its
    timeout, and a run that silently continues five minutes later on another transport measures nothing.
    -> (gather_group, tensor_device, description)"""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")       # one node: the hostname may not resolve, loopback always does
    dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=max(timeout_s, 600)))
    cpu = torch.device("cpu")
    world = dist.get_world_size()

    def over_gloo(why):
        gather_records(np.zeros(1, dtype=RECORD_DTYPE), dst=0, device=cpu)
        dist.barrier()
        return None, cpu, "gloo" + (f" ({why[:160]})" if why else "")

    if backend != "nccl":
        return over_gloo("")
    # 1. one distinct GPU per rank?  (the physical device: a launcher may have narrowed every rank's visibility to "cuda:0")
    if not force_rccl_failure:          # (the test hook goes straight to step 2: its failure there stands in for RCCL's own)
        devs = [None] * world
        dist.all_gather_object(devs, _physical_device(device))
        if not all(devs):
            return over_gloo("RCCL not attempted: a rank has no GPU")
        if len(set(devs)) != world:
            dup = next(d for d in devs if devs.count(d) > 1)
            return over_gloo(f"RCCL not attempted: ranks {[r for r, d in enumerate(devs) if d == dup]} share one GPU ({dup})")

    def agreed(ok, err, stage):
        """-> True if every rank succeeded, False if every rank failed; raises if the ranks disagree."""
        flag = torch.tensor([ok, 1 - ok], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.SUM)
        n_ok, n_bad = int(flag[0].item()), int(flag[1].item())
        if n_bad == 0:
            return True
        msgs = [None] * world
        dist.all_gather_object(msgs, err)
        if n_ok:
            raise RuntimeError(f"RCCL {stage} failed on ranks {[r for r, m in enumerate(msgs) if m]} only "
                               f"({next(m for m in msgs if m)[:200]}): not falling back")
        agreed.why = next((m for m in msgs if m), "unknown")
        return False

    # 2. the group
    ok, err, group = 1, "", None
    try:
        if force_rccl_failure is True:  # test hook (any other true value only waives the precondition: CPU tests)
            raise RuntimeError("forced by TWK_BENCH_FORCE_RCCL_FAILURE")
        group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=timeout_s), device_id=device)
    except Exception as e:              # noqa: BLE001 - whatever RCCL throws, the outcome is agreed below
        ok, err = 0, repr(e)
    if not agreed(ok, err, "group creation"):
        return over_gloo(f"RCCL failed to initialise: {agreed.why}")
    # 3. first use, outside any step
    try:
        gather_records(np.zeros(1, dtype=RECORD_DTYPE), dst=0, device=device, group=group)
        dist.barrier(group=group)
        if device is not None and device.type == "cuda":
            torch.cuda.synchronize(device)
    except Exception as e:              # noqa: BLE001
        ok, err = 0, repr(e)
    if not agreed(ok, err, "first gather"):
        return over_gloo(f"RCCL failed to initialise: {agreed.why}")
    return group, device, "nccl"


# The last gather_records of this process: payload bytes that crossed between ranks (dst: received, others: sent) and the
# seconds from the moment every rank had arrived (the counts all-gather) to the end of the transfers.
LAST_GATHER = {"bytes": 0, "seconds": 0.0}


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
    t_counts = time.perf_counter()          # every rank has arrived: what follows is the transfer itself
    LAST_GATHER["bytes"], LAST_GATHER["seconds"] = (total - counts[dst]) * ITEM if rank == dst else n_mine * ITEM, 0.0
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
        if device.type == "cuda":
            # For RCCL wait() only orders torch's current stream behind the transfer; the host runs on.  A sender's
            # payload is the engine's own HBM buffer (HipLd.device_records_tensor), which the next compute call rewrites
            # on the engine's streams - memory torch's allocator knows nothing about - so the transfer must have
            # *finished* before this function returns, on senders and on the receiver alike.
            torch.cuda.current_stream(device).synchronize()
    LAST_GATHER["seconds"] = time.perf_counter() - t_counts
    if rank != dst:
        return None
    return deliver(out)
