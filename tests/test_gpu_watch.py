"""The count-launch log and the outlier watch (twk_hip_launch_log, twk_hip.hip watch_launch): measurement integrity - a run
that silently takes 1.7 x as long must be visible in the numbers it reports.  No reference counterpart."""
import numpy as np
import pytest

import tomahawk_amd as T
from tests import util

pytestmark = pytest.mark.gpu


def test_every_count_launch_is_logged_with_its_clock_and_samples_are_not(hip, opt):
    N, M = 300_000, 520
    rng = np.random.default_rng(9)
    al = (rng.random((M, N, 2)) < rng.uniform(0.05, 0.5, size=M)[:, None, None]).astype(np.int8)
    util.upload(hip, al)
    f = T.Filters(minR2=0.1)
    hip.timing_reset()
    assert hip.launch_log() == ([], 0)
    hip.ld_all(T.MODE_UNPHASED, f, tile_variants=128)          # long rows: three products through the matrix, every launch sampled first
    tm = hip.timing()
    stats, seen = hip.launch_log()
    assert seen == len(stats) == tm["count_launches"] >= 10 and tm["three_launches"] == tm["count_launches"]      # the samples are not in the log
    assert all(x["kind"] == 1 and x["words_per_row"] == (N + 31) // 32 and x["row_pairs"] > 0 and x["ms"] > 0 for x in stats)
    assert all(1000 < x["shader_mhz"] < 2600 for x in stats), [x["shader_mhz"] for x in stats]
    assert all(0 <= x["xcd_finish_spread_us"] < 5e4 for x in stats)
    assert sum(x["row_pairs"] for x in stats) == tm["row_pairs"] and abs(sum(x["ms"] for x in stats) - tm["count_ms"]) < 1e-3 * tm["count_ms"] + 1e-3
    assert tm["outlier_launches"] == sum(x["outlier"] for x in stats)
    # the four-product and the phased forms are kinds of their own; the log keeps the most recent launches and honours the capacity asked for
    opt.set("three", 0)
    hip.ld_all(T.MODE_UNPHASED, f, tile_variants=256)
    hip.ld_all(T.MODE_PHASED, f, tile_variants=256)
    stats2, seen2 = hip.launch_log()
    assert seen2 > seen and [x["kind"] for x in stats2[:len(stats)]] == [1] * len(stats) and {x["kind"] for x in stats2[len(stats):]} == {0}
    last3, seen3 = hip.launch_log(3)
    assert seen3 == seen2 and last3 == stats2[-3:]
    hip.timing_reset()
    assert hip.launch_log() == ([], 0)


def test_fused_launches_are_logged_by_kind(hip, opt):
    N, M = 2504, 3000
    rng = np.random.default_rng(10)
    al = (rng.random((M, N, 2)) < rng.uniform(0.05, 0.5, size=M)[:, None, None]).astype(np.int8)
    util.upload(hip, al)
    f = T.Filters(minR2=0.1)
    hip.timing_reset()
    hip.ld_all(T.MODE_PHASED, f)
    hip.ld_all(T.MODE_UNPHASED, f)
    opt.set("three", 0)
    hip.ld_all(T.MODE_UNPHASED, f)
    kinds = [x["kind"] for x in hip.launch_log()[0]]
    assert kinds and set(kinds) == {2, 3, 4} and kinds == sorted(kinds, key=lambda k: {2: 0, 4: 1, 3: 2}[k])


def test_delivery_thread_hands_over_the_same_records_in_the_same_order(hip, opt):
    """Option async_delivery (default 1): a finished launch's sorted survivors are copied aside on the device and a second
    thread of the engine takes them to the sink while the caller's thread enqueues the next launches.  Same records, same
    order, same totals as with the caller's thread doing both - for band launches (fused, sampled for their candidate density:
    the samples' own survivors must not reach the sink), matrix-sized tiles, a window, list / probe passes - and a sink that
    fails still fails the call, after every launch in front of it was delivered."""
    import ctypes as C
    from tomahawk_amd import hip as H
    N, M = 2504, 1500
    al = util.mosaic_alleles(M, N, 31, n_founders=10, switch=0.05, mut=0.01)
    util.upload(hip, al)
    cases = [(T.MODE_UNPHASED, dict(minR2=0.1), dict()), (T.MODE_UNPHASED, dict(minR2=0.004), dict(window=T.OPT_R2_SCREEN)),
             (T.MODE_PHASED, dict(minR2=0.02), dict()), (T.MODE_PHASED, dict(minR2=0.0), dict(tile_variants=256)),
             (T.MODE_AUTO, dict(minR2=0.1), dict(window=T.OPT_WINDOW, l_window=4000))]
    for mode, fk, kw in cases:
        got = {}
        for on in (0, 1):
            opt.set("async_delivery", on)
            got[on] = hip.ld_all(mode, T.Filters(**fk), **kw)
        assert got[0][1:] == got[1][1:] and got[0][2] == len(got[0][0]) > 100
        assert got[0][0].tobytes() == got[1][0].tobytes()            # the same records in the same order
    # a failing sink: the call reports it whichever thread ran the sink
    calls = []

    def bad_sink(_user, recs, n):
        calls.append(n)
        return 1 if len(calls) >= 2 else 0

    for on in (0, 1):
        opt.set("async_delivery", on)
        del calls[:]
        cb = H._SINK(bad_sink)
        f = T.Filters(minR2=0.0)._c()
        rc = hip._lib.twk_hip_ld_all(hip._ctx, T.MODE_PHASED, C.byref(f), 0, 1, 128, 0, 0, cb, None, None, None)
        assert rc != 0 and len(calls) >= 2


def test_delivery_thread_failures_backpressure_and_recovery(hip, opt):
    """The engine's second thread beyond the happy path (twk_delivery.h): launches big enough to be
    staged (>= 2^18 survivors), with
      * a sink that fails ON the delivery thread: the call returns the failure and its text, the thread is joined, the pool freed;
      * a staging allocation that fails (option deliver_fail_alloc_at): the calling thread hands that launch over itself, behind
        what is queued - same records, same order;
      * a staging copy that fails (deliver_fail_copy_at): the call fails, the buffer goes back to the pool;
      * one staging buffer and a sink slower than the launches (deliver_buffers = 1): launches wait for the buffer - same records;
    and after every failure the context runs the next call as if nothing had happened."""
    import ctypes as C
    import time
    from tomahawk_amd import hip as H
    N, M = 500, 1500
    al = util.random_alleles(M, N, 5)
    util.upload(hip, al)
    f = T.Filters(minR2=0.0)
    kw = dict(tile_variants=1024)                       # launches of 523,776 / 487,424 / 113,050 pairs: two staged, one handed over directly
    opt.set("async_delivery", 0)
    want, npairs, nrec = hip.ld_all(T.MODE_PHASED, f, **kw)
    opt.unset("async_delivery")
    assert nrec == len(want) > 1_100_000 and npairs == M * (M - 1) // 2

    def run_ok():
        got, p, r = hip.ld_all(T.MODE_PHASED, f, **kw)
        assert (p, r) == (npairs, nrec) and got.tobytes() == want.tobytes()

    run_ok()
    # the sink fails at its second call - on the delivery thread (the first launch is staged: 523,776 >= 2^18 records)
    import threading
    seen = []

    def bad_sink(_user, recs, n):
        seen.append((n, threading.get_ident()))
        return 1 if len(seen) >= 2 else 0

    cb = H._SINK(bad_sink)
    fc = f._c()
    rc = hip._lib.twk_hip_ld_all(hip._ctx, T.MODE_PHASED, C.byref(fc), 0, 1, 1024, 0, 0, cb, None, None, None)
    assert rc != 0 and len(seen) >= 2 and seen[0][0] >= 2 ** 18 and seen[0][1] != threading.get_ident()
    assert b"sink failed" in hip._lib.twk_hip_last_error(hip._ctx)
    run_ok()
    # a failing staging allocation: the launch is delivered by the calling thread; a failing staging copy: the call fails
    for nth in (1, 2):
        opt.set("deliver_fail_alloc_at", nth)
        run_ok()
        opt.unset("deliver_fail_alloc_at")
    opt.set("deliver_fail_copy_at", 2)
    with pytest.raises(T.HipError) as e:
        hip.ld_all(T.MODE_PHASED, f, **kw)
    assert "staging copy" in str(e.value)
    opt.unset("deliver_fail_copy_at")
    run_ok()
    # back-pressure: one buffer, a slow sink
    opt.set("deliver_buffers", 1)
    chunks = []

    def slow_sink(_user, recs, n):
        time.sleep(0.05)
        buf = (C.c_char * (n * T.RECORD_DTYPE.itemsize)).from_address(recs)
        chunks.append(np.frombuffer(buf, dtype=T.RECORD_DTYPE).copy())
        return 0

    cb = H._SINK(slow_sink)
    n_p, n_r = C.c_uint64(0), C.c_uint64(0)
    rc = hip._lib.twk_hip_ld_all(hip._ctx, T.MODE_PHASED, C.byref(fc), 0, 1, 512, 0, 0, cb, None, C.byref(n_p), C.byref(n_r))
    assert rc == 0 and n_r.value == nrec
    got = np.concatenate(chunks)
    assert np.sort(got, order=["idxA", "idxB"]).tobytes() == np.sort(want, order=["idxA", "idxB"]).tobytes()
