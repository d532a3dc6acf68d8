"""The fused count -> r2 screen -> candidate list form of the count kernel (k_count_screen_t + k_ld_stats_list,
ld_count.hip.h / ld_math.hip.h): for rows of <= 128 K-chunks (N <= 65,536 phased) the block that counted a tile screens
it in registers and only candidate pairs reach the math kernel - no count matrix in HBM, no one-thread-per-pair math
front end.  Reference shape: the per-pair loop count -> math of lib/ld/ld_engine.cpp:1898-2015 with PhasedMath
(:1162-1310).  The records must be those of the plain path, bit for bit, and those of the oracle."""
import numpy as np
import pytest

import tomahawk_amd as T
from oracle import oracle as O
from tests import util
from tests.test_gpu_configs import _cohort_alleles

pytestmark = pytest.mark.gpu
ORDER = ["idxA", "idxB"]


def _both(hip, opt, call):
    """Run `call` with the fused form off and on -> (plain records, fused records, fused launches, candidates)."""
    opt.set("fused", 0)
    hip.timing_reset()
    plain = call()
    assert hip.timing()["fused_launches"] == 0
    opt.set("fused", 1)
    hip.timing_reset()
    fused = call()
    tm = hip.timing()
    opt.unset("fused")
    return plain, fused, tm["fused_launches"], tm["candidates"]


@pytest.mark.parametrize("N", [64, 1000, 2504, 8192, 40_000])
def test_fused_equals_plain_and_oracle(hip, opt, N):
    """Every phased-math route that qualifies: -p, default mode without missing data, with and without the allele-count
    band (r2 screen), a window, shards, several cut-offs incl. one placed on existing r2 values."""
    M = 1700 if N <= 2504 else 700
    al = _cohort_alleles(M, N, 500 + N)
    data, mask, variants = util.upload(hip, al)
    n_fused = 0
    for mode in (T.MODE_PHASED, T.MODE_AUTO):
        for minR2 in (0.1, 0.6, 0.004):
            for wopt in (0, T.OPT_R2_SCREEN):
                f = T.Filters(minR2=minR2)
                (p, np0, _), (q, np1, nr1), nf, ncand = _both(hip, opt, lambda: hip.ld_all(mode, f, window=wopt))
                assert nf > 0 and np0 == np1 == M * (M - 1) // 2 and nr1 == len(q) == len(p) > 20
                assert ncand >= len(q)                          # every survivor was a candidate
                assert ncand < 0.5 * np1 or minR2 < 0.01         # and the screen did screen
                assert np.sort(p, order=ORDER).tobytes() == np.sort(q, order=ORDER).tobytes()
                n_fused += nf
    # cut-offs on, just below and just above r2 values that exist
    f0 = T.Filters(minR2=0.05)
    base, _, _ = hip.ld_all(T.MODE_PHASED, f0)
    r2 = np.unique(base["R2"]); r2 = r2[r2 < 1]
    for x in r2[:: max(1, len(r2) // 5)][:5]:
        for cut in (np.nextafter(x, 0.0), x, np.nextafter(x, 1.0)):
            f = T.Filters(minR2=float(cut))
            (p, _, _), (q, _, _), nf, _ = _both(hip, opt, lambda: hip.ld_all(T.MODE_PHASED, f))
            assert nf > 0 and np.sort(p, order=ORDER).tobytes() == np.sort(q, order=ORDER).tobytes()
    # window mode (positions 100 bp apart) and shards of it
    f = T.Filters(minR2=0.1)
    (p, np0, _), (q, np1, _), nf, _ = _both(hip, opt, lambda: hip.ld_all(T.MODE_PHASED, f, window=T.OPT_WINDOW, l_window=30_000))
    assert nf > 0 and np0 == np1 and len(p) > 20 and np.sort(p, order=ORDER).tobytes() == np.sort(q, order=ORDER).tobytes()
    parts = [hip.ld_all(T.MODE_PHASED, f, part=k, n_parts=3) for k in range(3)]
    whole, _, _ = hip.ld_all(T.MODE_PHASED, f)
    assert np.sort(np.concatenate([x[0] for x in parts]), order=ORDER).tobytes() == np.sort(whole, order=ORDER).tobytes()
    # small super-tiles: diagonal + rectangle launches, two-deep pipeline
    (p, _, _), (q, _, _), nf, _ = _both(hip, opt, lambda: hip.ld_all(T.MODE_PHASED, f, tile_variants=256))
    assert nf > 3 and np.sort(p, order=ORDER).tobytes() == np.sort(q, order=ORDER).tobytes()
    # and the oracle agrees (sampled: it is scalar)
    sub = np.sort(np.random.default_rng(N).choice(M, size=240, replace=False))
    hip.set_problem(N, len(sub))
    hip.upload(data[sub], util.to_hip_meta(variants[sub]), None)
    want = O.all_pairs(data[sub], None, variants[sub], N, O.settings(minR2=0.1, phased=True), vector_only=False)
    hip.timing_reset()
    got, _, _ = hip.ld_all(T.MODE_PHASED, T.Filters(minR2=0.1))
    assert hip.timing()["fused_launches"] > 0 and len(want) > 20
    util.assert_records_match(got, want, variants[sub])


def test_fused_first_pass_of_default_mode_with_missing_data(hip, opt):
    """Default mode with missing genotypes: the plain phased planes decide the pairs without missing data (fused), the
    masked unphased planes the rest (through C as before); single tiles run both passes into one survivor buffer."""
    N, M = 1500, 1300
    al = _cohort_alleles(M, N, 41, miss=True)
    data, mask, variants = util.upload(hip, al)
    f = T.Filters(minR2=0.2)
    for call in (lambda: hip.ld_all(T.MODE_AUTO, f), lambda: hip.ld_all(T.MODE_AUTO, f, window=T.OPT_R2_SCREEN),
                 lambda: hip.ld_tile(T.MODE_AUTO, 0, M, 0, M, True, f) + (0,),
                 lambda: hip.ld_tile(T.MODE_AUTO, 130, 300, 430, 513, False, f) + (0,)):      # (a rectangle next to the diagonal: LD lives there)
        p, q, nf, _ = _both(hip, opt, call)
        assert nf > 0 and len(p[0]) > 20
        assert np.sort(p[0], order=ORDER).tobytes() == np.sort(q[0], order=ORDER).tobytes()
    sub = np.sort(np.random.default_rng(3).choice(M, size=230, replace=False))
    hip.set_problem(N, len(sub))
    hip.upload(data[sub], util.to_hip_meta(variants[sub]), mask[sub])
    want = O.all_pairs(data[sub], mask[sub], variants[sub], N, O.settings(minR2=0.2), vector_only=False)
    got, _, _ = hip.ld_all(T.MODE_AUTO, f)
    util.assert_records_match(got, want, variants[sub], double_root=util.double_root_vetter(data[sub], mask[sub], variants[sub], N))


def test_fused_candidate_overflow_falls_back_to_the_plain_path(hip, opt):
    """The candidate list lives in the slot's count-matrix buffer (a third of the pairs fit).  Haplotype-block data
    with a cut-off just above the screen's floor puts most pairs on it: the tile is redone through C, the rest of the
    call runs plain, the records are the same."""
    N, M = 300, 900
    al = util.mosaic_alleles(M, N, 5, n_founders=3, switch=0.002, mut=0.0005)
    util.upload(hip, al)
    f = T.Filters(minR2=2e-6)
    for tile in (0, 256):
        (p, _, _), (q, _, nr), nf, ncand = _both(hip, opt, lambda: hip.ld_all(T.MODE_PHASED, f, tile_variants=tile))
        assert nf >= 1 and ncand > M * (M - 1) // 2 // 3 or tile      # the first fused tile overflowed ...
        assert len(p) == nr > 0.6 * M * (M - 1) // 2                   # ... (most pairs survive this cut-off)
        assert np.sort(p, order=ORDER).tobytes() == np.sort(q, order=ORDER).tobytes()
    # the next call starts fused again
    hip.timing_reset()
    hip.ld_all(T.MODE_PHASED, T.Filters(minR2=0.5))
    assert hip.timing()["fused_launches"] > 0


def test_fused_forced_on_long_rows(hip, opt):
    """option "fused" = 2 (test hook): the fused form whatever the row length - N = 70,000 phased is 137 K-chunks per
    tile, which the default policy would split into units near the end of a launch."""
    N, M = 70_000, 400
    al = util.mosaic_alleles(M, N, 8, n_founders=6, switch=0.01, mut=0.001)
    util.upload(hip, al)
    f = T.Filters(minR2=0.3)
    opt.set("fused", 1)
    hip.timing_reset()
    plain, _, _ = hip.ld_all(T.MODE_PHASED, f)
    assert hip.timing()["fused_launches"] == 0           # too long for the default policy
    opt.set("fused", 2)
    hip.timing_reset()
    forced, _, _ = hip.ld_all(T.MODE_PHASED, f)
    assert hip.timing()["fused_launches"] > 0 and len(plain) > 50
    assert np.sort(plain, order=ORDER).tobytes() == np.sort(forced, order=ORDER).tobytes()


def test_fisher_pipeline_agrees_with_the_oracle_in_any_order(hip):
    """twk_hip_fisher_exact: the engine's Fisher kernels (starting points -> walk-length bins -> the reference's walk, one
    table per lane; ld_math.hip.h) against kt_fisher_exact as the oracle restates it (fisher_math.cpp:231-267): rare and
    common margins at n = 5,008, P from 1 down through the underflow region to exactly 0, tables beyond the log-factorial
    table, and balanced tables at n = 2,000,000 whose P runs through 1e-270 .. 1e-320.  The walks start on cells whose
    index is a multiple of 11 - where the reference's own recurrence is re-synchronised - so every term they add is the
    reference's own number: the agreement does not depend on the size of P.  Binned and unbinned give the same bits."""
    rng = np.random.default_rng(12)
    hip.set_problem(2504, 8)
    n = 5008
    tabs = []
    for _ in range(6000):
        r1 = int(rng.integers(1, n)); c1 = int(rng.choice([rng.integers(1, 60), rng.integers(1, n)]))
        lo, hi = max(0, r1 + c1 - n), min(r1, c1)
        mean = r1 * c1 / n
        n11 = int(np.clip(round(mean + rng.normal() * rng.choice([0.5, 3, 12, 40]) * max(1.0, np.sqrt(mean * (1 - r1 / n) * (1 - c1 / n)))), lo, hi))
        tabs.append([n11, r1 - n11, c1 - n11, n - r1 - c1 + n11])
    tabs += [[3741, 794, 8, 465], [3763, 10, 775, 460], [1792, 1208, 1994, 14], [2504, 0, 0, 2504], [0, 2504, 2504, 0], [5008, 0, 0, 0],
             [1, 0, 0, 5007], [2500, 4, 4, 2500], [20_000, 5, 7, 30_000], [7_000, 6_000, 6_500, 7_200]]        # the last two: beyond the table
    tabs = np.array(tabs, dtype=np.int32)
    assert (tabs >= 0).all()
    binned, _ = hip.fisher_exact(tabs)
    as_given, _ = hip.fisher_exact(tabs, ordered=False)
    assert binned.tobytes() == as_given.tobytes()
    want = np.array([O.fisher(*[int(x) for x in t])[2] for t in tabs])
    assert ((want < 1e-250) & (want > 0)).sum() > 20 and (want > 0.05).sum() > 100 and (want == 0).sum() > 10
    assert np.allclose(binned, want, rtol=1e-8, atol=1e-322)
    assert (binned[want == 0] == 0).all()
    # the largest problem whose log-factorial table goes into LDS, and the first one beyond it
    for N in (4024, 4025):
        hip.set_problem(N, 8)
        t2 = np.array([[a, N - a, b, N - b] for a, b in zip(rng.integers(0, N, 3000), rng.integers(0, N, 3000))], dtype=np.int32)
        got2, _ = hip.fisher_exact(t2)
        want2 = np.array([O.fisher(*[int(x) for x in t]) [2] for t in t2])
        assert np.allclose(got2, want2, rtol=1e-8, atol=1e-322), N
    # n = 2,000,000: q crosses the smallest normal double between these tables
    hip.set_problem(1_000_000, 8)
    ks = np.arange(12_600, 14_300, 50)
    big = np.array([[500_000 + k, 500_000 - k, 500_000 - k, 500_000 + k] for k in ks], dtype=np.int32)
    got, _ = hip.fisher_exact(np.tile(big, (40, 1)))          # (enough tables for the bins to engage)
    assert (got.reshape(40, -1) == got[: len(big)]).all()
    want = np.array([O.fisher(*[int(x) for x in t])[2] for t in big])
    assert (want > 1e-290).sum() > 5 and ((want > 0) & (want < 1e-300)).sum() >= 1 and (want == 0).sum() > 3
    assert np.allclose(got[: len(big)], want, rtol=1e-6, atol=1e-322)


@pytest.mark.parametrize("N", [2504, 6000])
def test_fisher_order_and_table_placement_do_not_change_a_record(hip, opt, N):
    """The records of a survivor-heavy call with the Fisher walks binned by length (default) and in the order the survivors
    were appended, with the log-factorial table in LDS (N <= 4,024) and in global memory: the same bytes."""
    M = 1500
    al = _cohort_alleles(M, N, 77 + N)
    util.upload(hip, al)
    f = T.Filters(minR2=0.05)
    for mode in (T.MODE_PHASED, T.MODE_UNPHASED):
        base, _, _ = hip.ld_all(mode, f)
        assert len(base) > 2000 and (base["P"] < 1e-20).sum() > 100
        for env in ({"fisher_order": 0}, {"fisher_lds": 0}, {"fisher_order": 0, "fisher_lds": 0}):
            for k, v in env.items():
                opt.set(k, v)
            got, _, _ = hip.ld_all(mode, f)
            for k in env:
                opt.unset(k)
            assert np.sort(base, order=ORDER).tobytes() == np.sort(got, order=ORDER).tobytes(), env


@pytest.mark.parametrize("env", [{"seg": 64}, {"seg": 32, "xcd_queues": 8}, {"xcd_queues": 8},
                                 {"patch_rows": 16, "patch_cols": 32, "seg": 128, "xcd_queues": 4}])
def test_count_kernel_work_orders_give_the_same_counts(hip, opt, env):
    """The measured options of the count kernel's work order (K segments per patch, one unit queue per XCD,
    patch shape) are not the default, but they stay in the kernel: contingency cells bit-exact against the oracle and
    records equal to the default order's, on rows long enough for them to engage (N = 70,000 phased: 137 K-chunks)."""
    N, M = 70_000, 700
    al = util.mosaic_alleles(M, N, 21, n_founders=6, switch=0.01, mut=0.002)
    data, mask, variants = util.upload(hip, al)
    base, _, _ = hip.ld_all(T.MODE_PHASED, T.Filters(minR2=0.2))
    for k, v in env.items():
        opt.set(k, v)
    cells = hip.count_tile(T.MODE_PHASED, 0, M, 0, M)
    got, _, _ = hip.ld_all(T.MODE_PHASED, T.Filters(minR2=0.2))
    for k in env:
        opt.unset(k)
    rng = np.random.default_rng(2)
    for i, j in zip(rng.integers(0, M, 40), rng.integers(0, M, 40)):
        assert np.array_equal(cells[i, j], O.count_phased(data[i], None, data[j], None, N)), (i, j)
    assert len(base) > 50 and np.sort(base, order=ORDER).tobytes() == np.sort(got, order=ORDER).tobytes()


@pytest.mark.parametrize("N", [64, 1000, 2504, 16_000, 50_000])
def test_fused_unphased_equals_plain_and_oracle(hip, opt, N):
    """The unphased form (k_count_screen_unphased_t + k_ld_stats_list_unphased): a variant pair's four products are
    gathered from four lanes with DPP moves, the screen is the interval test on the admissible haplotype frequencies
    (UnphasedMath, ld_engine.cpp:1312-1560), candidates carry HH, HQ, QH, QQ.  `-u` on data without missing genotypes, with
    and without the allele-count band, windows, shards, small tiles, cut-offs on existing r2 values; up to rows of 49
    chunks (the default policy fuses up to 128 chunks of N bits)."""
    M = 1500 if N <= 2504 else 600
    al = _cohort_alleles(M, N, 700 + N)
    data, mask, variants = util.upload(hip, al)
    mode = T.MODE_UNPHASED
    for minR2 in (0.1, 0.6, 0.004):
        for wopt in (0, T.OPT_R2_SCREEN):
            f = T.Filters(minR2=minR2)
            (p, np0, _), (q, np1, nr1), nf, ncand = _both(hip, opt, lambda: hip.ld_all(mode, f, window=wopt))
            assert nf > 0 and np0 == np1 == M * (M - 1) // 2 and nr1 == len(q) == len(p) > 20
            assert ncand >= len(q) or minR2 < 0.01                 # (at 0.004 the list may overflow and the tile be redone plain)
            assert ncand < 0.5 * np1 or minR2 < 0.05
            assert np.sort(p, order=ORDER).tobytes() == np.sort(q, order=ORDER).tobytes()
    base, _, _ = hip.ld_all(mode, T.Filters(minR2=0.05))
    r2 = np.unique(base["R2"]); r2 = r2[r2 < 1]
    for x in r2[:: max(1, len(r2) // 4)][:4]:
        for cut in (np.nextafter(x, 0.0), x, np.nextafter(x, 1.0)):
            f = T.Filters(minR2=float(cut))
            (p, _, _), (q, _, _), nf, _ = _both(hip, opt, lambda: hip.ld_all(mode, f))
            assert nf > 0 and np.sort(p, order=ORDER).tobytes() == np.sort(q, order=ORDER).tobytes()
    f = T.Filters(minR2=0.1)
    (p, np0, _), (q, np1, _), nf, _ = _both(hip, opt, lambda: hip.ld_all(mode, f, window=T.OPT_WINDOW, l_window=30_000))
    assert nf > 0 and np0 == np1 and len(p) > 20 and np.sort(p, order=ORDER).tobytes() == np.sort(q, order=ORDER).tobytes()
    parts = [hip.ld_all(mode, f, part=k, n_parts=3) for k in range(3)]
    whole, _, _ = hip.ld_all(mode, f)
    assert np.sort(np.concatenate([x[0] for x in parts]), order=ORDER).tobytes() == np.sort(whole, order=ORDER).tobytes()
    (p, _, _), (q, _, _), nf, _ = _both(hip, opt, lambda: hip.ld_all(mode, f, tile_variants=256))
    assert nf > 3 and np.sort(p, order=ORDER).tobytes() == np.sort(q, order=ORDER).tobytes()
    # odd tile origins through the single-tile entry point (plane rows start even whatever the variant index)
    for a0, nA, b0, nB, diag in ((0, M, 0, M, True), (3, M // 5 + 1, M // 4 + 2, M // 3 + 11, False), (129, 200, 129, 333, True)):
        (p, _), (q, _), nf, _ = _both(hip, opt, lambda: hip.ld_tile(mode, a0, nA, b0, nB, diag, f))
        assert nf > 0 and np.sort(p, order=ORDER).tobytes() == np.sort(q, order=ORDER).tobytes()
    sub = np.sort(np.random.default_rng(N).choice(M, size=220, replace=False))
    hip.set_problem(N, len(sub))
    hip.upload(data[sub], util.to_hip_meta(variants[sub]), None)
    want = O.all_pairs(data[sub], None, variants[sub], N, O.settings(minR2=0.1, unphased=True), vector_only=False)
    hip.timing_reset()
    got, _, _ = hip.ld_all(mode, T.Filters(minR2=0.1))
    assert hip.timing()["fused_launches"] > 0 and len(want) > 20
    util.assert_records_match(got, want, variants[sub], double_root=util.double_root_vetter(data[sub], None, variants[sub], N))


@pytest.mark.parametrize("mode", [T.MODE_PHASED, T.MODE_UNPHASED])
def test_fused_tiles_that_end_inside_a_block_tile(hip, opt, mode):
    """Tiles whose variant counts are not multiples of the 128-row block tile, in the middle of an LD-rich matrix: the
    columns (and rows) of the last block tile that lie beyond the tile's own variants belong to other tiles and must not
    become candidates (found by the overflow path, which cuts a tile into one-row strips)."""
    N, M = 800, 900
    al = util.mosaic_alleles(M, N, 15, n_founders=4, switch=0.004, mut=0.001)
    util.upload(hip, al)
    f = T.Filters(minR2=0.3)
    for a0, nA, b0, nB, diag in ((265, 1, 265, 1, True), (265, 1, 266, 100, False), (100, 70, 100, 200, True), (300, 129, 429, 131, False), (5, 3, 400, 2, False)):
        (p, np0), (q, np1), nf, _ = _both(hip, opt, lambda: hip.ld_tile(mode, a0, nA, b0, nB, diag, f))
        assert np0 == np1 and np.sort(p, order=ORDER).tobytes() == np.sort(q, order=ORDER).tobytes(), (a0, nA, b0, nB)
        assert len(q) == 0 or ((q["idxA"] >= a0) & (q["idxA"] < a0 + nA) & (q["idxB"] >= b0) & (q["idxB"] < b0 + nB)).all()
    assert len(hip.ld_tile(mode, 100, 70, 100, 200, True, f)[0]) > 100
    # and the whole run with a survivor buffer that overflows everywhere (strips of one row)
    whole, _, _ = hip.ld_all(mode, f)
    opt.set("record_cap", 300)
    strips, _, nrec = hip.ld_all(mode, f)
    opt.unset("record_cap")
    assert nrec == len(whole) > 5000 and np.sort(whole, order=ORDER).tobytes() == np.sort(strips, order=ORDER).tobytes()


@pytest.mark.parametrize("N", [10, 40, 656, 752, 1100, 2504, 100_010])
def test_padding_behind_the_last_chunk_is_skipped_without_changing_a_count(hip, opt, N):
    """Rows are padded to whole 32-word K-chunks; the contraction of a row's last chunk stops at the last half-slot
    (2 words) that carries data (CountWork::last_halves).  The sample counts cover 1..15 live half-slots in both
    layouts (2N bits phased, N bits unphased), with and without missing genotypes (the mask planes are padded alike):
    contingency cells equal to the full contraction's and the oracle's, records of the plain and the fused path equal."""
    M = 300 if N < 50_000 else 140
    for miss in (0.0, 0.02):
        al = util.mosaic_alleles(M, N, N + int(miss * 100), n_founders=6, switch=0.02, mut=0.003, miss_rate=miss, miss_variants=0.5 if miss else 0.0)
        data, mask, variants = util.upload(hip, al)
        rng = np.random.default_rng(N)
        for mode, count in ((T.MODE_PHASED, O.count_phased), (T.MODE_UNPHASED, O.count_unphased)):
            opt.set("skip_pad", 0)
            full = hip.count_tile(mode, 0, M, 0, M)
            rec_full, _, _ = hip.ld_all(mode, T.Filters(minR2=0.1))
            opt.unset("skip_pad")
            cells = hip.count_tile(mode, 0, M, 0, M)
            assert np.array_equal(cells, full)
            for i, j in zip(rng.integers(0, M, 60), rng.integers(0, M, 60)):
                mi = mask[i] if mask is not None and variants["gt_missing"][i] else None
                mj = mask[j] if mask is not None and variants["gt_missing"][j] else None
                assert np.array_equal(cells[i, j], count(data[i], mi, data[j], mj, N)), (mode, i, j)
            for fused in ("0", "1"):
                opt.set("fused", int(fused))
                rec, _, _ = hip.ld_all(mode, T.Filters(minR2=0.1))
                opt.unset("fused")
                assert len(rec) > 10 and np.sort(rec, order=ORDER).tobytes() == np.sort(rec_full, order=ORDER).tobytes()


@pytest.mark.parametrize("mode", [T.MODE_PHASED, T.MODE_UNPHASED])
def test_candidate_slot_windows_of_any_size_give_the_same_records(hip, opt, mode):
    """A wave of the fused kernels reserves candidate slots several tiles ahead (ScreenWork::chunk) and marks what it does not
    use; the list kernels skip marked slots.  One atomic per wave and tile (0), tiny windows (3), the default and windows far
    larger than any tile needs (100,000: most of the list is unused slots, and the list overflows where it is short - the
    tile is then redone the plain way): the same records, and the slot counter never below the number of survivors."""
    N, M = 2504, 1900
    al = _cohort_alleles(M, N, 4242)
    util.upload(hip, al)
    f = T.Filters(minR2=0.05)
    opt.set("fused", 0)
    base, _, _ = hip.ld_all(mode, f, window=T.OPT_WINDOW, l_window=60_000)
    opt.set("fused", 1)
    assert len(base) > 5000
    for chunk in ("0", "3", None, "700", "100000"):
        if chunk is None:
            opt.unset("cand_chunk")
        else:
            opt.set("cand_chunk", int(chunk))
        for tv in (0, 256):
            hip.timing_reset()
            got, _, _ = hip.ld_all(mode, f, window=T.OPT_WINDOW, l_window=60_000, tile_variants=tv)
            tm = hip.timing()
            assert tm["fused_launches"] > 0 and tm["candidates"] >= len(got), (chunk, tv)
            assert np.sort(base, order=ORDER).tobytes() == np.sort(got, order=ORDER).tobytes(), (chunk, tv)
    opt.unset("cand_chunk")
    opt.unset("fused")


@pytest.mark.parametrize("mode", [T.MODE_PHASED, T.MODE_UNPHASED])
def test_band_launches_equal_matrix_sized_tiles(hip, opt, mode):
    """A fused launch leaves no count matrix behind, so the engine sizes it by its work: a band of rows over every column
    the rows reach, at most 8 launches per region (twk_hip.hip region_impl; option band_launch).  Same records, bit for
    bit, as the matrix-sized tiles of round 3 - all pairs, the allele-count band, a window, shards, a rectangle - as one
    launch and (band_work_log2 small) as several; and when a launch outgrows its candidate list or its survivor buffer
    (band_list_entries / record_cap small) its rows are redone as matrix-sized tiles.  The reference's shape: one pass
    over the block pairs with nothing stored per pair (lib/ld/ld_engine.cpp:1898-2015)."""
    N, M = 2504, 6000
    al = _cohort_alleles(M, N, 4242)
    util.upload(hip, al)
    f = T.Filters(minR2=0.1)
    calls = {
        "all": lambda: hip.ld_all(mode, f),
        "band": lambda: hip.ld_all(mode, f, window=T.OPT_R2_SCREEN),
        # a Fisher cut-off that splits the survivors: the dropped ones sort behind the kept (with the unused slots of a band launch's sort)
        "band, P cut-off": lambda: hip.ld_all(mode, T.Filters(minR2=0.1, minP=1e-300), window=T.OPT_R2_SCREEN),
        "window": lambda: hip.ld_all(mode, f, window=T.OPT_WINDOW, l_window=40_000),
        "shard": lambda: hip.ld_all(mode, f, part=1, n_parts=3),
        "window shard": lambda: hip.ld_all(mode, f, part=2, n_parts=3, window=T.OPT_WINDOW, l_window=25_000),
        "rectangle": lambda: hip.ld_region(mode, f, 100, 2000, 2500, 3300, False),
    }
    sizes = {}
    for name, call in calls.items():
        opt.set("band_launch", 0)
        hip.timing_reset()
        base, np0, nr0 = call()
        sizes[name] = nr0
        t0 = hip.timing()
        assert t0["fused_launches"] == t0["count_launches"] >= 1 and nr0 == len(base) > 100, name
        want = np.sort(base, order=ORDER).tobytes()
        opt.set("band_launch", 1)
        for log2, n_launch in ((19, 1), (10, None), (4, None)):
            opt.set("band_work_log2", log2)
            hip.timing_reset()
            got, np1, nr1 = call()
            t1 = hip.timing()
            assert np1 == np0 and nr1 == nr0 and np.sort(got, order=ORDER).tobytes() == want, (name, log2)
            assert t1["fused_launches"] == t1["count_launches"] and 1 <= t1["count_launches"] <= 8, (name, log2, t1)
            if log2 < 19 and name in ("all", "band", "band, P cut-off", "rectangle"):
                assert t1["count_launches"] > 1, (name, log2, t1)
            if n_launch is not None and name in ("all", "window", "rectangle"):
                assert t1["count_launches"] == n_launch, (name, log2, t1)
            if name == "window":       # in (idxA, idxB) order launch after launch: the order the records reach the writer in
                key = got["idxA"].astype(np.uint64) << np.uint64(32) | got["idxB"].astype(np.uint64)
                assert (np.diff(key.astype(np.int64)) > 0).all(), (name, log2)
        opt.set("band_work_log2", 19)
        if name.startswith("band"):        # allele-count order: last band first by default, first band first gives the same records
            opt.set("band_reverse", 0)
            opt.set("band_work_log2", 10)
            got, np1, nr1 = call()
            opt.unset("band_reverse")
            opt.set("band_work_log2", 19)
            assert np1 == np0 and nr1 == nr0 and np.sort(got, order=ORDER).tobytes() == want, (name, "band_reverse=0")
        for key, value in (("band_list_entries", 64), ("record_cap", 50)):
            opt.set(key, value)
            hip.timing_reset()
            got, np1, nr1 = call()
            t1 = hip.timing()
            opt.unset(key)
            assert np1 == np0 and nr1 == nr0 and np.sort(got, order=ORDER).tobytes() == want, (name, key)
            assert t1["count_launches"] > 1, (name, key, t1)          # the band launch, then the tiles that redid it
    assert 100 < sizes["band, P cut-off"] < sizes["band"]          # the cut-off dropped some and kept some
    # more survivors than one piece of the host staging buffer (2^20 records): they reach the sink piece by piece, in order
    f_low = T.Filters(minR2=0.0004)
    opt.set("band_launch", 0)
    base, np0, nr0 = hip.ld_all(mode, f_low)
    opt.set("band_launch", 1)
    opt.set("band_list_entries", 1 << 25)
    hip.timing_reset()
    got, np1, nr1 = hip.ld_all(mode, f_low)
    assert hip.timing()["count_launches"] == 1 and nr1 == nr0 == len(got) > (1 << 20) + 1000
    assert np.sort(got, order=ORDER).tobytes() == np.sort(base, order=ORDER).tobytes()
    key = got["idxA"].astype(np.uint64) << np.uint64(32) | got["idxB"].astype(np.uint64)
    assert (np.diff(key.astype(np.int64)) > 0).all()
