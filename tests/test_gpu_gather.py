"""The gather of a one-process multi-GPU run inside the C++ product (twk_hip_gather_records, include/twk_hip.h; `tomahawk calc`
with engine option gather = 1): every GPU keeps its survivors in HBM and they travel GPU to GPU over RCCL - ncclCommInitAll once
per device set, one group of exact-size ncclSend / ncclRecv - into one GPU's sink, which then leaves for the host once.  The north
star's "final RCCL gather of .two output blocks over xGMI"; reference analogue: every slave's flush of its output block into the
shared writer (lib/ld/ld_engine.cpp:1742-1802).  On a one-GPU box the same calls run as a loop from the sink to itself; the cases
that need two GPUs skip there, with the reason, and run the first time a multi-GPU box does."""
import os
import subprocess

import numpy as np
import pytest

import tomahawk_amd as T
from tests import util
from tomahawk_amd import hostlib

pytestmark = pytest.mark.gpu
ORDER = ["idxA", "idxB"]


def _same(a, b):
    return np.sort(a, order=ORDER).tobytes() == np.sort(b, order=ORDER).tobytes()


def test_rccl_moves_a_sink_to_itself_on_one_gpu(hip):
    assert T.gather_backend().startswith("rccl "), T.gather_backend()
    N, M = 2000, 1800
    al = util.mosaic_alleles(M, N, 17, n_founders=8, switch=0.03, mut=0.01)
    util.upload(hip, al)
    f = T.Filters(minR2=0.02)
    want, npairs, nrec = hip.ld_all(T.MODE_UNPHASED, f)
    assert nrec == len(want) > 20_000
    with pytest.raises(T.HipError):                       # the sink must be on
        T.gather_records([hip])
    hip.set_device_sink(True)
    try:
        _, _, nrec2 = hip.ld_all(T.MODE_UNPHASED, f)
        assert nrec2 == nrec
        n, ms = T.gather_records([hip])                   # one context: nothing to move
        assert n == nrec and ms == 0
        n, ms = T.gather_records([hip], self_loop=True)   # ... unless asked to send it round: ncclSend / ncclRecv to itself
        assert n == nrec and ms > 0
        ptr, n_dev = hip.device_records()
        assert n_dev == nrec and ptr
        got = hip.drain_device_sink()
        assert len(got) == nrec and _same(got, want)
        assert hip.device_records() == (0, 0)             # drained
        n, ms = T.gather_records([hip], self_loop=True)   # an empty sink goes round as well
        assert n == 0
    finally:
        hip.set_device_sink(False)


def test_cli_gather_option_writes_the_same_file_records(tmp_path):
    N, M = 1500, 1200
    al = util.mosaic_alleles(M, N, 23, n_founders=8, switch=0.04, mut=0.01, miss_rate=0.02, miss_variants=0.2)
    twk = str(tmp_path / "in.twk")
    pos = (1000 + 37 * np.arange(M)).astype(np.uint32)
    hostlib.write_twk(twk, al, pos, np.zeros(M, np.uint32), np.ones(M, np.uint8), block_size=100)
    outs = {}
    for tag, extra in (("stream", []), ("gather", ["--engine-option", "gather=1"]), ("gather_w", ["--engine-option", "gather=1", "-w", "9000"]), ("stream_w", ["-w", "9000"])):
        out = str(tmp_path / f"{tag}.two")
        env = {k: v for k, v in os.environ.items() if k != "NCCL_DEBUG"}       # (this pool's image exports NCCL_DEBUG=VERSION, which asks RCCL for its banner)
        r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", twk, "-o", out, "-r", "0.01"] + extra, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        if "gather" in tag:
            assert "Gathered" in r.stderr and " over rccl " in r.stderr, r.stderr[-800:]
        assert r.stdout == "", r.stdout[:300]            # (RCCL's start-up banner goes to stdout unless NCCL_DEBUG says NONE: the engine says so where the caller says nothing)
        recs, _ = hostlib.read_two(out)
        outs[tag] = np.sort(recs, order=["ridA", "packA", "ridB", "packB"])
    assert len(outs["stream"]) > 10_000 and outs["stream"].tobytes() == outs["gather"].tobytes()
    assert 500 < len(outs["stream_w"]) < len(outs["stream"]) and outs["stream_w"].tobytes() == outs["gather_w"].tobytes()


def test_gather_between_two_gpus_equals_one(tmp_path):
    if T.device_count() < 2:
        pytest.skip(f"the gather between GPUs needs >= 2 devices; this box shows {T.device_count()} (runs the first time a multi-GPU box does)")
    N, M = 3000, 2048
    al = util.mosaic_alleles(M, N, 29, n_founders=8, switch=0.03, mut=0.01)
    from oracle import oracle as O
    data, mask = O.bitvectors_from_alleles(al)
    variants = O.variants_from_alleles(al)
    f = T.Filters(minR2=0.02)
    with T.HipLd(0) as one:
        one.set_problem(N, M); one.upload(data, util.to_hip_meta(variants), mask)
        want, _, nrec = one.ld_all(T.MODE_UNPHASED, f)
    n_dev = min(T.device_count(), 8)
    engines = [T.HipLd(g) for g in range(n_dev)]
    try:
        held = []
        for g, e in enumerate(engines):
            e.set_problem(N, M); e.upload(data, util.to_hip_meta(variants), mask)
            e.set_device_sink(True)
            _, _, k = e.ld_all(T.MODE_UNPHASED, f, part=g, n_parts=n_dev)
            held.append(k)
        assert sum(held) == nrec and min(held) > 0
        n, ms = T.gather_records(engines, dst=0)
        assert n == nrec and ms > 0
        assert [e.device_records()[1] for e in engines] == [nrec] + [0] * (n_dev - 1)
        got = engines[0].drain_device_sink()
        assert _same(got, want)
    finally:
        for e in engines:
            e.close()
    # and the CLI: TWK_HIP_GPUS = n with the gather on writes the records of the one-GPU run
    twk = str(tmp_path / "in.twk")
    hostlib.write_twk(twk, al, (1000 + 10 * np.arange(M)).astype(np.uint32), np.zeros(M, np.uint32), np.ones(M, np.uint8), block_size=100)
    outs = []
    for env, extra in (({}, []), ({"TWK_HIP_GPUS": str(n_dev)}, ["--engine-option", "gather=1"])):
        out = str(tmp_path / f"g{len(outs)}.two")
        r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", twk, "-o", out, "-u", "-r", "0.02"] + extra, capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        if extra:
            assert f"of {n_dev} GPU(s) into GPU 0 over rccl" in r.stderr, r.stderr[-800:]
        outs.append(np.sort(hostlib.read_two(out)[0], order=["ridA", "packA", "ridB", "packB"]))
    assert len(outs[0]) > 10_000 and outs[0].tobytes() == outs[1].tobytes()
