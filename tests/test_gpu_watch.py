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
