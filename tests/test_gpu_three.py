"""The three-product form of the unphased contraction (k_count3_list_t / k_count3_screen_unphased_t + k_screen3_pairs +
k_recount_unphased, ld_count.hip.h): per word and variant pair HH = popc(H_A & H_B) and S = popc(Q_A & C_B) + popc(C_A & Q_B)
with C = H | Q - all that UnphasedMath's r2 screen reads (lib/ld/ld_engine.cpp:1363-1375: minhap / maxhap from n11 and the
double hets) - instead of HH, HQ, QH, QQ; pairs that pass the screen get their four products counted afresh from their rows.
The reference's list kernel has the same shape: one popcount, the other cells from the margins (ld_engine.cpp:244-246).
The records must be those of the four-product forms, byte for byte, and those of the oracle."""
import numpy as np
import pytest

import tomahawk_amd as T
from oracle import oracle as O
from tests import util
from tests.test_gpu_configs import _cohort_alleles

pytestmark = pytest.mark.gpu
ORDER = ["idxA", "idxB"]


def _both(hip, opt, call, expect_three=True, on=1):
    """`call` with the three-product form off and on -> (four-product result, three-product result, timing of the latter)."""
    opt.set("three", 0)
    hip.timing_reset()
    four = call()
    assert hip.timing()["three_launches"] == 0
    opt.set("three", on)
    hip.timing_reset()
    three = call()
    tm = hip.timing()
    opt.unset("three")
    if expect_three:
        assert tm["three_launches"] > 0
    return four, three, tm


def _same(a, b):
    return np.sort(a, order=ORDER).tobytes() == np.sort(b, order=ORDER).tobytes()


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("N", [64, 1000, 2504, 16_000, 50_000])
def test_three_product_equals_four_product_and_oracle(hip, opt, N, fused):
    """`-u` on data without missing genotypes: whole triangle with and without the allele-count band, cut-offs incl. ones
    placed on existing r2 values, a window, shards, small tiles, odd tile origins - fused (short rows: candidates straight
    from the count kernel's epilogue) and through the (HH, S) matrix (fused = 0: k_screen3_pairs)."""
    M = 1500 if N <= 2504 else 600
    al = _cohort_alleles(M, N, 900 + N)
    data, mask, variants = util.upload(hip, al)
    opt.set("fused", fused)
    mode = T.MODE_UNPHASED
    # (the engine samples every launch's candidate density first and keeps to four products on data as rich in LD as this cohort -
    # test_sampling_decides_...; three = 2 takes the form regardless)
    on = 2
    _b = _both
    def _both_on(hip, opt, call, expect_three=True):
        return _b(hip, opt, call, expect_three, on=on)
    for minR2 in (0.1, 0.6, 0.004):
        for wopt in (0, T.OPT_R2_SCREEN):
            f = T.Filters(minR2=minR2)
            (p, np0, _), (q, np1, nr1), tm = _both_on(hip, opt, lambda: hip.ld_all(mode, f, window=wopt), expect_three=(minR2 >= 0.1))
            assert np0 == np1 == M * (M - 1) // 2 and nr1 == len(q) == len(p) > 20
            assert _same(p, q), (minR2, wopt)
            if minR2 >= 0.1:                        # the screen screens, and every survivor was recounted
                assert len(q) <= tm["recount_candidates"] < 0.5 * np1
    base, _, _ = hip.ld_all(mode, T.Filters(minR2=0.05))
    r2 = np.unique(base["R2"]); r2 = r2[r2 < 1]
    for x in r2[:: max(1, len(r2) // 4)][:4]:
        for cut in (np.nextafter(x, 0.0), x, np.nextafter(x, 1.0)):
            f = T.Filters(minR2=float(cut))
            (p, _, _), (q, _, _), _ = _both_on(hip, opt, lambda: hip.ld_all(mode, f))
            assert _same(p, q), cut
    f = T.Filters(minR2=0.1)
    (p, np0, _), (q, np1, _), _ = _both_on(hip, opt, lambda: hip.ld_all(mode, f, window=T.OPT_WINDOW, l_window=30_000))
    assert np0 == np1 and len(p) > 20 and _same(p, q)
    opt.set("three", on)
    parts = [hip.ld_all(mode, f, part=k, n_parts=3) for k in range(3)]
    whole, _, _ = hip.ld_all(mode, f)
    assert _same(np.concatenate([x[0] for x in parts]), whole)
    (p, _, _), (q, _, _), tm = _both_on(hip, opt, lambda: hip.ld_all(mode, f, tile_variants=256))
    assert tm["three_launches"] > 3 and _same(p, q)
    for a0, nA, b0, nB, diag in ((0, M, 0, M, True), (3, M // 5 + 1, M // 4 + 2, M // 3 + 11, False), (129, 200, 129, 333, True)):
        (p, _), (q, _), _ = _both_on(hip, opt, lambda: hip.ld_tile(mode, a0, nA, b0, nB, diag, f))
        assert _same(p, q), (a0, nA, b0, nB)
    # and the oracle agrees (sampled: it is scalar)
    sub = np.sort(np.random.default_rng(N).choice(M, size=220, replace=False))
    hip.set_problem(N, len(sub))
    hip.upload(data[sub], util.to_hip_meta(variants[sub]), None)
    want = O.all_pairs(data[sub], None, variants[sub], N, O.settings(minR2=0.1, unphased=True), vector_only=False)
    opt.set("three", on)
    hip.timing_reset()
    got, _, _ = hip.ld_all(mode, T.Filters(minR2=0.1))
    tm = hip.timing()
    assert tm["three_launches"] > 0 and (tm["fused_launches"] > 0) == bool(fused) and len(want) > 20
    util.assert_records_match(got, want, variants[sub], double_root=util.double_root_vetter(data[sub], None, variants[sub], N))


@pytest.mark.parametrize("min_chunks", [8, 1])
def test_three_product_on_long_rows_with_tiles_split_along_k(hip, opt, min_chunks):
    """N = 300,000: rows of 293 chunks - beyond the fused form, and the last tiles of a launch are cut along K, their (HH, S)
    added into the matrix with atomics (StoreCounts3, k_zero_tiles over 64-row tiles).  count_min_chunks = 1 cuts them finer."""
    N, M = 300_000, 420
    al = _cohort_alleles(M, N, 4242)
    util.upload(hip, al)
    opt.set("count_min_chunks", min_chunks)
    for minR2, wopt in ((0.1, 0), (0.3, T.OPT_R2_SCREEN), (0.02, 0)):
        f = T.Filters(minR2=minR2)
        (p, np0, _), (q, np1, nr), tm = _both(hip, opt, lambda: hip.ld_all(T.MODE_UNPHASED, f, window=wopt), expect_three=(minR2 >= 0.1), on=2)
        assert tm["fused_launches"] == 0 and np0 == np1 and nr == len(q) == len(p) > 20 and _same(p, q), (minR2, wopt)
    (p, _, _), (q, _, _), tm = _both(hip, opt, lambda: hip.ld_all(T.MODE_UNPHASED, T.Filters(minR2=0.1), tile_variants=128), on=2)
    assert tm["three_launches"] >= 6 and _same(p, q)


@pytest.mark.parametrize("N", [40_000, 131_072 + 64])
def test_matrix_form_on_rows_that_end_inside_their_last_chunk(hip, opt, N):
    """The three-product kernel through the count matrix (fused = 0 sends rows this short that way) on rows whose last chunk is partly
    filled: the rolled loop over its live half-slots reads a variant's H and Q words with one ds_read2_b64, like the unrolled one.  N = 40,000: 1,250 words = 39 chunks + 2 words (1 live half-slot); 131,136: 4,098 words = 128 chunks + 2 words; and
    rows whose last chunk holds 9 .. 12 half-slots: N = 40,000 + 32 * 18 .. """
    M = 500
    for n in (N, N + 32 * 18, N + 32 * 23):
        al = _cohort_alleles(M, n, 31 + n)
        util.upload(hip, al)
        opt.set("fused", 0)
        f = T.Filters(minR2=0.1)
        (p, np0, _), (q, np1, nr), tm = _both(hip, opt, lambda: hip.ld_all(T.MODE_UNPHASED, f), on=2)
        assert tm["fused_launches"] == 0 and tm["three_launches"] > 0
        assert np0 == np1 and nr == len(q) == len(p) > 50 and _same(p, q), n
        opt.set("skip_pad", 0)
        r, _, _ = hip.ld_all(T.MODE_UNPHASED, f)
        opt.unset("skip_pad")
        assert _same(p, r)


def test_candidate_rich_launches_fall_back_to_four_products(hip, opt):
    """A cut-off that nearly every pair passes: the matrix form's candidate list (1/128 of the pairs) overflows and the tile is
    redone with four products; the fused form keeps its candidates but the call stops using three products after the first
    launch (three = 2 keeps to them).  The records are the same every way."""
    N, M = 1200, 700
    al = util.mosaic_alleles(M, N, 77, n_founders=4, switch=0.004, mut=0.001)
    util.upload(hip, al)
    f = T.Filters(minR2=0.002)
    for fused in (0, 1):
        opt.set("fused", fused)
        (p, _, _), (q, _, nr), tm = _both(hip, opt, lambda: hip.ld_all(T.MODE_UNPHASED, f, tile_variants=256), expect_three=False)
        assert nr == len(q) == len(p) > 50_000 and _same(p, q)
        n_default = tm["three_launches"]
        opt.set("three", 2)
        hip.timing_reset()
        r, _, _ = hip.ld_all(T.MODE_UNPHASED, f, tile_variants=256)
        tm2 = hip.timing()
        opt.unset("three")
        assert _same(p, r) and tm2["three_launches"] >= n_default and tm2["three_launches"] > 3
        if fused:
            assert n_default < tm2["three_launches"]            # the default gave the form up after a rich launch
    # phased math, masked planes and a zero cut-off never take the form
    opt.unset("fused")
    for mode, ff in ((T.MODE_PHASED, T.Filters(minR2=0.1)), (T.MODE_UNPHASED, T.Filters(minR2=0.0))):
        hip.timing_reset()
        hip.ld_all(mode, ff)
        assert hip.timing()["three_launches"] == 0


def test_three_product_band_launches_and_probe_zone(hip, opt):
    """Band launches (candidate count known only after the count kernel: the recount is enqueued with the deferred math) at
    2,504 samples, several launches per region."""
    N, M = 2504, 6000
    al = _cohort_alleles(M, N, 31337)
    util.upload(hip, al)
    opt.set("band_work_log2", 12)
    f = T.Filters(minR2=0.2)
    for wopt, lw in ((0, 0), (T.OPT_R2_SCREEN, 0), (T.OPT_WINDOW, 60_000)):
        (p, np0, _), (q, np1, nr), tm = _both(hip, opt, lambda: hip.ld_all(T.MODE_UNPHASED, f, window=wopt, l_window=lw), on=2)      # (2: this cohort is rich in LD)
        assert tm["three_launches"] >= (2 if wopt == 0 else 1) and tm["fused_launches"] == tm["three_launches"], (wopt, tm)
        assert np0 == np1 and nr == len(q) == len(p) > 100 and _same(p, q), wopt


def test_sampling_decides_between_three_and_four_products_on_long_rows(hip, opt):
    """Through the count matrix (rows too long to fuse) a region first samples its candidate density (a sub-tile
    of its own middle for every launch) and takes the three-product form only where candidates are few: unlinked variants
    (the headline's synthetic input) -> three products in every launch; a cohort rich in LD -> four, with no launch wasted.
    Same records either way.  Fused launches (short rows) are sampled the same way."""
    N = 300_000
    rng = np.random.default_rng(5)
    M = 700
    iid = (rng.random((M, N, 2)) < rng.uniform(0.05, 0.5, size=M)[:, None, None]).astype(np.int8)
    iid[5] = iid[4]; iid[300] = iid[299]; iid[650, :, 0] = iid[649, :, 1]; iid[650, :, 1] = iid[649, :, 0]       # a few pairs in perfect LD
    util.upload(hip, iid)
    f = T.Filters(minR2=0.1)
    (p, np0, _), (q, np1, nr), tm = _both(hip, opt, lambda: hip.ld_all(T.MODE_UNPHASED, f, tile_variants=256))
    assert tm["three_launches"] == tm["count_launches"] >= 6 and tm["fused_launches"] == 0
    assert np0 == np1 and nr == len(q) == len(p) >= 3 and _same(p, q) and tm["recount_candidates"] >= 3
    al = _cohort_alleles(420, N, 4242)
    util.upload(hip, al)
    (p, _, _), (q, _, nr), tm = _both(hip, opt, lambda: hip.ld_all(T.MODE_UNPHASED, f, tile_variants=128), expect_three=False)
    assert tm["three_launches"] == 0 and tm["count_launches"] >= 6 and nr == len(q) == len(p) > 100 and _same(p, q)
    # short rows (fused launches): the same decision
    N2 = 2504
    iid2 = (rng.random((3000, N2, 2)) < rng.uniform(0.05, 0.5, size=3000)[:, None, None]).astype(np.int8)
    util.upload(hip, iid2)
    (p, _, _), (q, _, nr), tm = _both(hip, opt, lambda: hip.ld_all(T.MODE_UNPHASED, f))
    assert tm["fused_launches"] == tm["three_launches"] == tm["count_launches"] >= 1 and _same(p, q)
    util.upload(hip, _cohort_alleles(3000, N2, 99))
    (p, _, _), (q, _, nr), tm = _both(hip, opt, lambda: hip.ld_all(T.MODE_UNPHASED, f), expect_three=False)
    assert tm["three_launches"] == 0 and tm["fused_launches"] == tm["count_launches"] >= 1 and nr == len(q) == len(p) > 100 and _same(p, q)


@pytest.mark.parametrize("N", [2080, 2208, 2272, 2336, 2400, 2504, 2816])
def test_three_product_rows_that_end_inside_their_last_chunk(hip, opt, N):
    """The fused three-product kernel on rows whose last chunk is partly filled (the padding behind the last live 8 bytes is not
    contracted): 1, 3, 4, 5, 6, 8 and 12 live half-slots - odd and even counts, the shortest and the longest the skip is taken
    for.  Records against the four-product kernel's, and against the run that contracts the padding too."""
    live_halves = (((N + 31) // 32 - 1) % 32 + 2) // 2
    assert 1 <= live_halves <= 12
    M = 900
    al = _cohort_alleles(M, N, 77 + N)
    util.upload(hip, al)
    f = T.Filters(minR2=0.15)
    (p, np0, _), (q, np1, nr), tm = _both(hip, opt, lambda: hip.ld_all(T.MODE_UNPHASED, f), on=2)
    assert tm["fused_launches"] == tm["three_launches"] >= 1 and np0 == np1 and nr == len(q) == len(p) > 50 and _same(p, q), (N, live_halves)
    opt.set("skip_pad", 0)
    r, _, _ = hip.ld_all(T.MODE_UNPHASED, f)
    assert _same(p, r)
