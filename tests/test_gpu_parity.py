"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bar: contingency counts bit-exact; D, D', r, r2, chi-squared and Fisher P within 1e-6 relative
(P with an absolute floor where it underflows).
"""
import numpy as np
import pytest

import tomahawk_amd as T
from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu

CASES = [  # (N, M, seed)  N=100 is not a multiple of 32/64: exercises the padding paths
    (64, 40, 1), (100, 150, 2), (1000, 200, 3), (2049, 131, 4),
]


@pytest.mark.parametrize("N,M,seed", CASES)
def test_counts_phased_bitexact(hip, N, M, seed):
    al = util.random_alleles(M, N, seed, low_ac=6)
    data, mask, _ = util.upload(hip, al)
    got = hip.count_tile(T.MODE_PHASED, 0, M, 0, M, diag=False)
    for i in range(0, M, 7):
        for j in range(0, M, 5):
            want = O.count_phased(data[i], None, data[j], None, N)
            assert np.array_equal(got[i, j], want), (i, j, got[i, j], want)


@pytest.mark.parametrize("N,M,seed", CASES)
def test_counts_unphased_bitexact(hip, N, M, seed):
    al = util.random_alleles(M, N, seed, low_ac=6)
    data, mask, _ = util.upload(hip, al)
    got = hip.count_tile(T.MODE_UNPHASED, 0, M, 0, M, diag=False)
    for i in range(0, M, 7):
        for j in range(0, M, 5):
            want = O.count_unphased(data[i], None, data[j], None, N)
            assert np.array_equal(got[i, j], want), (i, j, got[i, j], want)


@pytest.mark.parametrize("N,M,seed", [(64, 60, 11), (128, 140, 12), (192, 90, 13)])
def test_counts_with_missing_bitexact(hip, N, M, seed):
    al = util.random_alleles(M, N, seed, miss_rate=0.1, miss_variants=0.4)
    data, mask, variants = util.upload(hip, al)
    assert mask is not None
    gp = hip.count_tile(T.MODE_PHASED, 0, M, 0, M)
    gu = hip.count_tile(T.MODE_UNPHASED, 0, M, 0, M)
    for i in range(0, M, 3):
        for j in range(0, M, 4):
            mi = mask[i] if variants["gt_missing"][i] else None
            mj = mask[j] if variants["gt_missing"][j] else None
            assert np.array_equal(gp[i, j], O.count_phased(data[i], mi, data[j], mj, N)), (i, j)
            assert np.array_equal(gu[i, j], O.count_unphased(data[i], mi, data[j], mj, N)), (i, j)


def test_count_tile_offsets_and_diag(hip):
    N, M = 300, 400
    al = util.random_alleles(M, N, 21)
    data, _, _ = util.upload(hip, al)
    sq = hip.count_tile(T.MODE_UNPHASED, 130, 100, 257, 143, diag=False)
    for i, j in [(0, 0), (99, 142), (50, 77), (3, 140)]:
        assert np.array_equal(sq[i, j], O.count_unphased(data[130 + i], None, data[257 + j], None, N))
    dg = hip.count_tile(T.MODE_PHASED, 128, 200, 128, 200, diag=True)
    assert not dg[10, 10].any() and not dg[20, 5].any()          # excluded by diag -> zero-filled
    assert np.array_equal(dg[5, 20], O.count_phased(data[133], None, data[148], None, N))
    assert np.array_equal(dg[0, 199], O.count_phased(data[128], None, data[327], None, N))


@pytest.mark.parametrize("mode,phased", [(T.MODE_PHASED, True), (T.MODE_UNPHASED, False)])
@pytest.mark.parametrize("N,M,seed", CASES)
def test_records_match_oracle(hip, mode, phased, N, M, seed):
    al = util.random_alleles(M, N, seed, low_ac=6)
    data, mask, variants = util.upload(hip, al, phase=int(phased))
    st = O.settings(minR2=0.0, phased=phased, unphased=not phased)
    want = O.all_pairs(data, mask, variants, N, st, vector_only=False)
    got, npairs, nrec = hip.ld_all(mode, T.Filters(minR2=0.0))
    assert npairs == M * (M - 1) // 2
    assert nrec == len(got) == len(want)
    util.assert_records_match(got, want, variants)


@pytest.mark.parametrize("minR2,minP,minD", [(0.1, 1.0, 0.0), (0.02, 1e-3, 0.0), (0.0, 1.0, 0.5)])
def test_filters(hip, minR2, minP, minD):
    N, M = 500, 160
    rng = np.random.default_rng(5)
    al = util.random_alleles(M, N, 5)
    for v in range(1, M, 2):        # correlated neighbours so that some pairs pass r2 >= 0.1
        flip = rng.random((N, 2)) < 0.15
        al[v] = np.where(flip, al[v], al[v - 1])
    data, mask, variants = util.upload(hip, al)
    for mode, phased in ((T.MODE_PHASED, True), (T.MODE_UNPHASED, False)):
        st = O.settings(minR2=minR2, minP=minP, minDprime=minD, phased=phased, unphased=not phased)
        want = O.all_pairs(data, mask, variants, N, st)
        got, _, _ = hip.ld_all(mode, T.Filters(minR2=minR2, minP=minP, minDprime=minD))
        assert len(want) > 0
        util.assert_records_match(got, want, variants)


@pytest.mark.parametrize("mode", [T.MODE_PHASED, T.MODE_UNPHASED, T.MODE_AUTO])
def test_records_with_missing(hip, mode):
    N, M = 128, 120
    al = util.random_alleles(M, N, 31, miss_rate=0.08, miss_variants=0.3, low_ac=4)
    data, mask, variants = util.upload(hip, al)
    st = O.settings(minR2=0.0, phased=(mode == T.MODE_PHASED), unphased=(mode == T.MODE_UNPHASED))
    want = O.all_pairs(data, mask, variants, N, st, vector_only=False)
    got, _, _ = hip.ld_all(mode, T.Filters(minR2=0.0))
    util.assert_records_match(got, want, variants)


@pytest.mark.parametrize("window", [0, 1])
def test_default_mode_regrouped_equals_per_tile_two_pass(hip, window):
    """Default mode with missing genotypes: the whole-triangle run (plain phased products for every
    pair + 3-plane products for the pairs that involve a variant with missing data, on the regrouped
    plane set) gives the records of the per-tile two-pass path (sub-regions), and of the oracle."""
    N, M, W = 128, 700, 6000
    al = util.random_alleles(M, N, 77, miss_rate=0.05, miss_variants=0.2, low_ac=3)
    rid = np.repeat([0, 1], [400, 300]).astype(np.uint32)
    pos = np.concatenate([np.arange(400) * 100 + 1000, np.arange(300) * 100 + 500])
    data, mask, variants = util.upload(hip, al, pos=pos, rid=rid)
    assert 50 < int((variants["an"] > 0).sum()) < 300
    f = T.Filters(minR2=0.0)
    kw = dict(window=window, l_window=W)
    whole, npairs, nrec = hip.ld_all(T.MODE_AUTO, f, **kw)
    assert nrec == len(whole)
    key = lambda r: list(zip(r["idxA"].tolist(), r["idxB"].tolist()))
    assert len(set(key(whole))) == len(whole)                  # every pair at most once
    assert all(a < b for a, b in key(whole))                   # A is the variant that comes first in the file
    # (1) sub-regions go through the per-tile two-pass path
    regs = [hip.ld_region(T.MODE_AUTO, f, 0, 300, 0, 300, True, **kw),
            hip.ld_region(T.MODE_AUTO, f, 0, 300, 300, 400, False, **kw),
            hip.ld_region(T.MODE_AUTO, f, 300, 400, 300, 400, True, **kw)]
    ref = np.concatenate([r[0] for r in regs])
    assert sum(r[1] for r in regs) == npairs
    a = np.sort(whole, order=["idxA", "idxB"]); b = np.sort(ref, order=["idxA", "idxB"])
    assert a.tobytes() == b.tobytes()
    # (2) shards of the regrouped run partition it
    parts = [hip.ld_all(T.MODE_AUTO, f, part=k, n_parts=3, **kw) for k in range(3)]
    assert sum(p[1] for p in parts) == npairs and sum(p[2] for p in parts) == nrec
    c = np.sort(np.concatenate([p[0] for p in parts]), order=["idxA", "idxB"])
    assert a.tobytes() == c.tobytes()
    # (3) the oracle
    if not window:
        want = O.all_pairs(data, mask, variants, N, O.settings(minR2=0.0), vector_only=False)
        util.assert_records_match(whole, want, variants)


def test_sharded_union_equals_whole(hip):
    """The multi-GPU partition: shards are disjoint and their union is the whole triangle."""
    N, M = 200, 700
    al = util.random_alleles(M, N, 41)
    data, mask, variants = util.upload(hip, al)
    f = T.Filters(minR2=0.0)
    whole, npairs, _ = hip.ld_all(T.MODE_PHASED, f, tile_variants=128)
    parts = [hip.ld_all(T.MODE_PHASED, f, part=k, n_parts=3, tile_variants=128) for k in range(3)]
    assert sum(p[1] for p in parts) == npairs == M * (M - 1) // 2
    keys = [set(zip(p[0]["idxA"].tolist(), p[0]["idxB"].tolist())) for p in parts]
    assert not (keys[0] & keys[1]) and not (keys[0] & keys[2]) and not (keys[1] & keys[2])
    assert set().union(*keys) == set(zip(whole["idxA"].tolist(), whole["idxB"].tolist()))
    loads = [p[1] for p in parts]
    assert max(loads) < 1.35 * min(loads)


def test_window_mode_sharded_and_slab(hip):
    """calc -w: in-window pairs only; shards partition them; a slab (rows band + halo) gives its band's records."""
    N, M, W = 96, 1500, 2500                      # positions 1000 + 100 v  ->  25 partners each side
    hip.set_problem(N, M)
    hip.generate_synthetic(5)
    f = T.Filters(minR2=0.0)
    whole, npairs_all, _ = hip.ld_all(T.MODE_PHASED, f)
    keyset = lambda r: set(zip(r["idxA"].tolist(), r["idxB"].tolist()))
    want = {(a, b) for (a, b) in keyset(whole) if (b - a) * 100 <= W}
    win, npw, nrw = hip.ld_all(T.MODE_PHASED, f, window=1, l_window=W)
    assert keyset(win) == want and nrw == len(want)
    assert npw == sum(min(25, M - 1 - i) for i in range(M))           # pairs inside the window
    parts = [hip.ld_all(T.MODE_PHASED, f, part=k, n_parts=4, window=1, l_window=W) for k in range(4)]
    assert sum(p[1] for p in parts) == npw and set().union(*[keyset(p[0]) for p in parts]) == want
    assert max(p[1] for p in parts) < 1.35 * min(p[1] for p in parts)  # by in-window pairs (64-row granularity), not triangle area
    # slab: rows [600, 900) + halo of 25 columns, generated with the global variant ids
    r0, r1, halo = 600, 900, 25
    hip.set_problem(N, r1 - r0 + halo)
    hip.generate_synthetic(5, first_variant=r0)
    slab, nps, _ = hip.ld_region(T.MODE_PHASED, f, 0, r1 - r0, 0, r1 - r0 + halo, True, window=1, l_window=W)
    got = {(a + r0, b + r0) for (a, b) in keyset(slab)}
    assert got == {(a, b) for (a, b) in want if r0 <= a < r1}
    bykey = {(int(r["idxA"]), int(r["idxB"])): r for r in whole}
    for r in slab[:: max(1, len(slab) // 50)]:
        w = bykey[(int(r["idxA"]) + r0, int(r["idxB"]) + r0)]
        assert r["R2"] == w["R2"] and np.array_equal(r["cnt"], w["cnt"])


def test_many_variants_beyond_grid_limits(hip):
    """More variants than a launch grid's y extent (65535): prep kernels loop, tiles cover everything."""
    N, M = 40, 70_000
    hip.set_problem(N, M)
    hip.generate_synthetic(3)
    ac, het, hom, _ = hip.marginals()
    for v in (0, 65534, 65535, 65536, 69_999):
        bv, a = T.synth_bitvector(3, N, v)
        assert a == ac[v]
    c = hip.count_tile(T.MODE_UNPHASED, 65500, 100, 69_900, 100)
    assert (c.sum(axis=2) == N).all() and (c[:, :, 3:6].sum(axis=2) == het[65500:65600, None]).all()
    recs, npairs, nrec = hip.ld_all(T.MODE_PHASED, T.Filters(minR2=0.9), collect=False)
    assert npairs == M * (M - 1) // 2


def test_overflow_is_reported_and_recovered(hip):
    N, M = 64, 300
    al = util.random_alleles(M, N, 51)
    util.upload(hip, al)
    with pytest.raises(T.HipError) as e:
        hip.ld_tile(T.MODE_PHASED, 0, M, 0, M, True, T.Filters(minR2=0.0), capacity=10)
    assert e.value.code == -4
    recs, npairs = hip.ld_tile(T.MODE_PHASED, 0, M, 0, M, True, T.Filters(minR2=0.0))
    assert npairs == M * (M - 1) // 2 and len(recs) > 10


def test_ld_all_recovers_from_survivor_overflow(hip, opt):
    """Dense output larger than the device record buffer: the tile is redone in row strips."""
    N, M = 64, 333
    al = util.random_alleles(M, N, 52)
    util.upload(hip, al)
    f = T.Filters(minR2=0.0)
    whole, npairs, nrec = hip.ld_all(T.MODE_PHASED, f)
    for cap in ("1000", "777", "50000"):
        opt.set("record_cap", int(cap))
        got, np2, nrec2 = hip.ld_all(T.MODE_PHASED, f, tile_variants=256)
        assert np2 == npairs and nrec2 == nrec == len(got)
        key = lambda r: np.lexsort((r["idxB"], r["idxA"]))
        a, b = whole[key(whole)], got[key(got)]
        assert np.array_equal(a["idxA"], b["idxA"]) and np.array_equal(a["idxB"], b["idxB"]) and np.array_equal(a["R2"], b["R2"])
    opt.unset("record_cap")


@pytest.mark.parametrize("N,M", [(1, 2), (3, 5), (64, 1), (31, 129)])
def test_degenerate_shapes(hip, N, M):
    """Tiny problems: one or two variants, a single sample, odd sizes."""
    al = util.random_alleles(M, N, 61, maf_lo=0.3, maf_hi=0.7)
    data, mask, variants = util.upload(hip, al)
    for mode, phased in ((T.MODE_PHASED, True), (T.MODE_UNPHASED, False)):
        st = O.settings(minR2=0.0, phased=phased, unphased=not phased)
        want = O.all_pairs(data, mask, variants, N, st) if M > 1 else np.zeros(0, dtype=O.RECORD_DTYPE)
        got, npairs, nrec = hip.ld_all(mode, T.Filters(minR2=0.0))
        assert npairs == M * (M - 1) // 2 and nrec == len(want)
        util.assert_records_match(got, want, variants)


def test_randomised_shapes_counts_bitexact(hip):
    """Sweep of random shapes (sample counts around the 32/64/128-bit and KC = 32-word boundaries, row
    counts around the 128-row tile): every contingency cell of every pair equals the oracle's."""
    rng = np.random.default_rng(2026)
    specials = [31, 32, 33, 63, 64, 65, 127, 128, 129, 511, 512, 513, 1023, 1024, 1025, 2047, 2049]
    for it in range(24):
        N = int(rng.choice(specials)) if it % 2 == 0 else int(rng.integers(1, 700))
        M = int(rng.choice([2, 3, 64, 127, 128, 129, 130, 255, 257])) if it % 3 == 0 else int(rng.integers(2, 200))
        miss = it % 4 == 1
        al = util.random_alleles(M, N, 1000 + it, maf_lo=0.02, maf_hi=0.98,
                                 miss_rate=0.2 if miss else 0.0, miss_variants=0.5 if miss else 0.0)
        data, mask, variants = util.upload(hip, al)
        a0 = int(rng.integers(0, M)); b0 = int(rng.integers(0, M))
        nA = int(rng.integers(1, M - a0 + 1)); nB = int(rng.integers(1, M - b0 + 1))
        for mode, counter in ((T.MODE_PHASED, O.count_phased), (T.MODE_UNPHASED, O.count_unphased)):
            got = hip.count_tile(mode, a0, nA, b0, nB)
            for i in rng.choice(nA, size=min(nA, 6), replace=False):
                for j in rng.choice(nB, size=min(nB, 6), replace=False):
                    A, B = a0 + int(i), b0 + int(j)
                    mA = mask[A] if mask is not None and variants["gt_missing"][A] else None
                    mB = mask[B] if mask is not None and variants["gt_missing"][B] else None
                    want = counter(data[A], mA, data[B], mB, N)
                    assert np.array_equal(got[i, j], want), (it, N, M, mode, A, B)


def test_fisher_p_values_large_tables(hip):
    """Fisher's P at a sample count where the reference walks ~1e5 tail terms per record and the device
    starts its walk at a verified e^-50 point instead: same P (1e-6), from P ~ 1 down to the underflow region."""
    N, M = 60000, 48
    rng = np.random.default_rng(12)
    al = util.random_alleles(M, N, 12)
    for k, noise in enumerate([0.0005, 0.05, 0.3, 0.45, 0.47, 0.48, 0.485, 0.49]):     # partners in LD with variant 0..7
        flip = rng.random((N, 2)) < noise
        al[M - 1 - k] = np.where(flip, 1 - al[k], al[k])
    data, mask, variants = util.upload(hip, al)
    got, _, _ = hip.ld_all(T.MODE_PHASED, T.Filters(minR2=0.0))
    want = O.all_pairs(data, mask, variants, N, O.settings(minR2=0.0, phased=True), vector_only=False)
    util.assert_records_match(got, want, variants)
    P = np.sort(want["P"])
    assert P[0] < 1e-250 and (P > 0.5).sum() > 100 and ((P > 1e-200) & (P < 1e-6)).sum() >= 2       # the whole range is exercised


@pytest.mark.parametrize("N,seed,founders,switch,mut,miss", [(64, 5001, 4, 0.02, 0.002, False), (250, 5004, 7, 0.02, 0.002, False),
                                                            (128, 5003, 6, 0.005, 0.0, True), (1000, 5006, 3, 0.005, 0.0, False)])
def test_haplotype_block_data_all_modes(hip, N, seed, founders, switch, mut, miss):
    """Real LD structure (mosaics of a few founder haplotypes): identical / complementary variants, D' = 1,
    double roots of the unphased cubic.  Pair sets and records equal the oracle's in every mode."""
    M = 140
    al = util.mosaic_alleles(M, N, seed, n_founders=founders, switch=switch, mut=mut,
                             miss_rate=0.05 if miss else 0.0, miss_variants=0.3 if miss else 0.0)
    data, mask, variants = util.upload(hip, al)
    for mode, ph in ((T.MODE_UNPHASED, False), (T.MODE_PHASED, True), (T.MODE_AUTO, None)):
        st = O.settings(minR2=0.0, phased=bool(ph), unphased=(ph is False))
        want = O.all_pairs(data, mask, variants, N, st, vector_only=False)
        got, _, _ = hip.ld_all(mode, T.Filters(minR2=0.0))
        assert len(want) > 1000
        util.assert_records_match(got, want, variants)


@pytest.mark.parametrize("N,seed,miss", [(64, 901, True), (7, 903, False), (320, 904, False), (33, 915, True), (2504, 908, False)])
def test_hostile_genotypes_all_modes(hip, N, seed, miss):
    """Allele frequencies at the extremes, all-het / complementary / identical variants, 30-95 % missing
    samples (the compiled reference and the oracle agree bit for bit on such data, 146 k records checked).
    N = 33 / seed 915 contains the one table found where glibc's pow(d2, 3) is not correctly rounded and the
    double-root pair goes the other way: allowed through double_root_vetter, nothing else is."""
    al = util.extreme_alleles(70, N, seed, miss)
    data, mask, variants = util.upload(hip, al)
    vet = util.double_root_vetter(data, mask, variants, N)
    for mode, ph in ((T.MODE_UNPHASED, False), (T.MODE_PHASED, True), (T.MODE_AUTO, None)):
        st = O.settings(minR2=0.0, phased=bool(ph), unphased=(ph is False))
        want = O.all_pairs(data, mask, variants, N, st, vector_only=False)
        got, _, _ = hip.ld_all(mode, T.Filters(minR2=0.0))
        util.assert_records_match(got, want, variants, double_root=vet)


def test_records_at_the_headline_sample_count(hip):
    """Record-level parity (not only table properties) at N ~ 1 M: haplotype-block genotypes with missing
    samples, N not a multiple of anything, every mode, against the scalar oracle."""
    N, M = 1_000_003, 14
    al = util.mosaic_alleles(M, N, 2, n_founders=5, switch=0.05, mut=0.01, miss_rate=0.01, miss_variants=0.3)
    data, mask, variants = util.upload(hip, al)
    vet = util.double_root_vetter(data, mask, variants, N)
    for mode, ph in ((T.MODE_UNPHASED, False), (T.MODE_PHASED, True), (T.MODE_AUTO, None)):
        st = O.settings(minR2=0.0, phased=bool(ph), unphased=(ph is False))
        want = O.all_pairs(data, mask, variants, N, st, vector_only=False)
        got, npairs, _ = hip.ld_all(mode, T.Filters(minR2=0.0))
        assert npairs == M * (M - 1) // 2 and len(want) > 60
        util.assert_records_match(got, want, variants, double_root=vet)


def test_invalid_arguments_are_rejected(hip):
    hip.set_problem(10, 20)
    hip.generate_synthetic(1)
    with pytest.raises(T.HipError):
        hip.count_tile(T.MODE_PHASED, 0, 30, 0, 5)             # rows beyond the problem
    with pytest.raises(T.HipError):
        hip.count_tile(T.MODE_AUTO, 0, 5, 0, 5)                # raw cells need an explicit mode
    with pytest.raises(T.HipError):
        hip.ld_tile(T.MODE_PHASED, 0, 5, 1, 5, True, T.Filters())   # diag needs the same origin
    with pytest.raises(T.HipError):
        hip.ld_all(7, T.Filters())
    with pytest.raises(T.HipError):
        hip.ld_all(T.MODE_PHASED, T.Filters(), part=2, n_parts=2)
    with pytest.raises(T.HipError):
        hip.set_problem(0, 5)


def test_synthetic_generator_matches_host_twin(hip):
    N, M = 1000, 64
    hip.set_problem(N, M)
    hip.generate_synthetic(42)
    ac, het, hom, miss = hip.marginals()
    data = np.zeros((M, O.words64(N)), dtype=np.uint64)
    for v in range(M):
        data[v], a = T.synth_bitvector(42, N, v)
        assert a == ac[v]
    got = hip.count_tile(T.MODE_UNPHASED, 0, M, 0, M)
    for i, j in [(0, 1), (5, 60), (63, 2)]:
        assert np.array_equal(got[i, j], O.count_unphased(data[i], None, data[j], None, N))
    assert 0.04 * 2 * N < ac.min() and ac.max() < 0.56 * 2 * N and not miss.any()


def test_large_sample_axis_roundtrip_property(hip):
    """Full-size rows (N = 1M samples): size-independent invariants of the tables."""
    N, M = 1_000_000, 256
    hip.set_problem(N, M)
    hip.generate_synthetic(7)
    ac, het, hom, _ = hip.marginals()
    cu = hip.count_tile(T.MODE_UNPHASED, 0, M, 0, M)
    cp = hip.count_tile(T.MODE_PHASED, 0, M, 0, M)
    assert (cu.sum(axis=2) == N).all() and (cp.sum(axis=2) == 2 * N).all()
    # row / column marginals of the 3x3 table are the per-variant genotype counts
    assert (cu[:, :, 3:6].sum(axis=2) == het[:, None]).all() and (cu[:, :, 6:9].sum(axis=2) == hom[:, None]).all()
    assert (cu[:, :, [1, 4, 7]].sum(axis=2) == het[None, :]).all()
    # phased: ALTALT + c[1] = ac_A ; table of (i,j) is the transpose of (j,i)
    assert (cp[:, :, 3] + cp[:, :, 1] == ac[:, None]).all()
    assert np.array_equal(cp[:, :, 1], cp[:, :, 2].T) and np.array_equal(cu[:, :, 5], cu[:, :, 7].T)
    # self-pair: every sample is on the diagonal of the table
    d = np.arange(M)
    assert (cu[d, d][:, [1, 2, 3, 5, 6, 7]] == 0).all()
