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
