"""The N > 1 path on CPU: the row-band partition (host arithmetic of the C ABI) and the gather of
survivor records over torch.distributed with the gloo backend, world size 2."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import tomahawk_amd as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("M", [1, 64, 700, 50_000, 200_000])
@pytest.mark.parametrize("n_parts", [1, 2, 3, 8])
def test_bands_partition_the_triangle(M, n_parts):
    total = M * (M - 1) // 2
    prev_end, pairs = 0, []
    for k in range(n_parts):
        r0, r1, n = T.shard_rows(M, k, n_parts)
        assert r0 == prev_end and r0 <= r1 <= M
        assert n == sum(M - 1 - i for i in range(r0, r1)) if M <= 700 else n >= 0
        prev_end = r1
        pairs.append(n)
    assert prev_end == M and sum(pairs) == total
    if M >= 50_000:
        assert max(pairs) <= 1.02 * total / n_parts          # balanced to 2 % at the headline size
        assert all(r % 64 == 0 for r in [T.shard_rows(M, k, n_parts)[0] for k in range(n_parts)])


def test_rectangle_bands():
    spans = [T.shard_rows(1000, k, 4, n_cols=333, triangle=False) for k in range(4)]
    assert spans[0][0] == 0 and spans[-1][1] == 1000 and sum(s[2] for s in spans) == 333_000
    with pytest.raises(T.HipError):
        T.shard_rows(10, 3, 3)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, M, q, two_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tomahawk_amd.dist import gather_records
    from tomahawk_amd import hostlib
    r0, r1, n = T.shard_rows(M, rank, world)
    # stand-in survivors: every 7th pair of this rank's band (no device in this test); ragged on purpose:
    # the middle rank has none at all, the last one five
    ij = [(i, j) for i in range(r0, r1) for j in range(i + 1, M)][::7]
    if rank == world - 1:
        ij = ij[:5]
    elif rank != 0:
        ij = []
    recs = np.zeros(len(ij), dtype=T.RECORD_DTYPE)
    recs["idxA"] = [p[0] for p in ij]; recs["idxB"] = [p[1] for p in ij]
    recs["R2"] = recs["idxA"] * 1e-3 + recs["idxB"] * 1e-6
    recs["flags"] = 3
    got = gather_records(recs, dst=0)
    empty = gather_records(np.zeros(0, dtype=T.RECORD_DTYPE), dst=0)
    if rank == 0:
        # the writer rank packs what it gathered into a real .two (forward + reverse blocks, flush rule, index)
        w = hostlib.TwoStream(two_path, 10, np.zeros(M, dtype=np.uint32), 1000 + 100 * np.arange(M, dtype=np.uint32), b_size=500)
        w.append(got[: len(got) // 2]); w.append(got[len(got) // 2:])
        n_written = w.close()
        q.put((got["idxA"].tolist(), got["idxB"].tolist(), got["R2"].tolist(), len(empty), (r0, r1, n), n_written))
    else:
        assert got is None and empty is None
        q.put((rank, len(ij), (r0, r1, n)))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_records_world3_gloo_and_writer_rank(tmp_path):
    from tomahawk_amd import hostlib
    M, world = 300, 3
    two_path = str(tmp_path / "gathered.two")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, M, q, two_path)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    root = next(o for o in outs if len(o) == 6)
    others = sorted(o for o in outs if len(o) == 3)
    idxA, idxB, r2, n_empty, band0, n_written = root
    (_, n1, band1), (_, n2, band2) = others
    assert n_empty == 0 and n1 == 0 and n2 == 5
    assert band0[1] == band1[0] and band1[1] == band2[0] and band0[2] + band1[2] + band2[2] == M * (M - 1) // 2
    # rank order is preserved; rank 0's records all lie in its band, the last rank's in its own
    n0 = len(idxA) - n2
    assert n0 > 100
    assert all(band0[0] <= a < band0[1] for a in idxA[:n0]) and all(band2[0] <= a < band2[1] for a in idxA[n0:])
    assert all(a < b for a, b in zip(idxA, idxB))
    np.testing.assert_allclose(r2, np.array(idxA) * 1e-3 + np.array(idxB) * 1e-6)
    # the .two the writer rank produced: every gathered pair once forward and once reversed
    recs, info = hostlib.read_two(two_path)
    assert n_written == len(recs) == 2 * len(idxA) and info["n_samples"] == 10
    fwd = {(1000 + 100 * a, 1000 + 100 * b) for a, b in zip(idxA, idxB)}
    seen = {(int(r["packA"]) >> 2, int(r["packB"]) >> 2) for r in recs}
    assert seen == fwd | {(b, a) for a, b in fwd}
    state, ent, _ = hostlib.two_index(two_path)
    assert state == 0 and int(ent[:, 2].sum()) == len(recs) and ent[:, 2].max() <= 500


@pytest.mark.parametrize("M,wv", [(200_000, 5000), (16_384, 5000), (4096, 300), (700, 5000), (64, 3)])
@pytest.mark.parametrize("n_parts", [1, 2, 3, 8])
def test_window_slabs_partition_the_band(M, wv, n_parts):
    """bench.py's configs[4] partition (tomahawk_amd.dist.window_slab): bands of rows with equal in-window pairs, each
    rank's slab = its band + the halo its window reaches; together they hold every in-window pair exactly once."""
    from tomahawk_amd.dist import window_slab, window_total_pairs
    total = window_total_pairs(M, wv)
    assert total == sum(min(wv, M - 1 - i) for i in range(M))
    prev, pairs = 0, []
    for k in range(n_parts):
        r0, r1, col_end, n = window_slab(M, wv, k, n_parts)
        assert r0 == prev and r0 <= r1 <= M and col_end == min(M, r1 + wv)
        assert n == sum(min(wv, M - 1 - i) for i in range(r0, r1))
        assert r0 % 64 == 0 or r0 == M
        prev = r1
        pairs.append(n)
    assert prev == M and sum(pairs) == total
    if M >= 200_000:
        assert max(pairs) <= 1.02 * total / n_parts


def _worker_groups(rank, world, port, q, backend, force_fail):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from tomahawk_amd.dist import gather_records, init_groups
    if isinstance(force_fail, int) and not isinstance(force_fail, bool):       # RCCL "fails" on this one rank only: every rank must raise
        if rank != force_fail:
            import torch.distributed as d
            real = d.new_group          # the other ranks "succeed" (no collective: the failing rank never calls new_group)
            d.new_group = lambda *a, **k: d.group.WORLD if k.get("backend") == "nccl" else real(*a, **k)
        try:
            init_groups(backend, None, force_rccl_failure=(rank == force_fail) or "others", timeout_s=60)
            q.put((rank, "no error"))
        except RuntimeError as e:
            q.put((rank, str(e)))
        dist.destroy_process_group()
        return
    group, xdev, desc = init_groups(backend, None, force_rccl_failure=force_fail, timeout_s=60)
    # the payload as bench.py hands it over: a uint8 tensor of n x 104 bytes (the engine's buffer on the GPU box)
    n = [3, 0, 7, 1][rank % 4]
    recs = np.zeros(n, dtype=T.RECORD_DTYPE)
    recs["idxA"] = rank; recs["idxB"] = 100 + np.arange(n)
    payload = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy())
    got = gather_records(payload, dst=0, device=xdev, group=group)
    raw = gather_records(payload, dst=0, device=xdev, group=group, to_host=False)
    me = torch.tensor([rank], dtype=torch.int64)
    seen = [torch.zeros_like(me) for _ in range(world)]
    dist.all_gather(seen, me, group=group)
    if rank == 0:
        assert raw.dtype == torch.uint8 and raw.numel() == len(got) * 104
        q.put((desc, got["idxA"].tolist(), got["idxB"].tolist(), sorted(int(x) for x in seen)))
    else:
        assert got is None and raw is None
        q.put((rank, desc))
    dist.barrier()
    dist.destroy_process_group()


def test_an_asymmetric_rccl_failure_is_an_error_on_every_rank():
    """RCCL group creation fails on rank 2 only: no rank falls back to gloo on its own, every rank raises (a non-zero
    exit of the job), and the message names the rank."""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_groups, args=(r, world, port, q, "nccl", 2)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert sorted(outs) == [0, 1, 2, 3]
    for r in range(world):
        assert "failed on ranks [2] only" in outs[r] and "not falling back" in outs[r], outs


@pytest.mark.parametrize("backend,force_fail", [("gloo", False), ("nccl", True), ("nccl", False)])
def test_backend_agreement_and_tensor_payload(backend, force_fail):
    """init_groups: a gloo control group, and the gather's backend agreed by all ranks.  With RCCL made to fail on
    every rank (there is no GPU here) every rank ends up on gloo together and says why; without the hook the
    precondition (one GPU per rank) already says no; the gather takes the uint8 tensor payload of the device-resident
    path."""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_groups, args=(r, world, port, q, backend, force_fail)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    root = next(o for o in outs if len(o) == 4)
    desc, idxA, idxB, seen = root
    assert seen == [0, 1, 2, 3]
    assert idxA == [0] * 3 + [2] * 7 + [3] * 1 and idxB == [100, 101, 102] + list(range(100, 107)) + [100]
    if force_fail:
        assert desc.startswith("gloo (RCCL failed to initialise") and "forced" in desc
    elif backend == "nccl":
        assert desc == "gloo (RCCL not attempted: a rank has no GPU)"
    else:
        assert desc == "gloo"
    assert all(o[1] == desc for o in outs if len(o) == 2)
