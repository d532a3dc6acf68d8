"""GPU parity at the sample counts and launch shapes of BASELINE.json's single-GPU configs that the small
cases in test_gpu_parity.py do not reach, record for record against the scalar oracle:

  configs[1]  N = 100,000 phased, several 128-row tiles (diagonal + rectangle launches)
  configs[2]  N ~ 1,000,000 unphased: Fisher's tail-start shortcut on the cubic path's rounded tables
  configs[4]  N = 10,000,000, windowed, Fisher P <= 1e-6, missing samples, every mode

plus the two review findings of round 1 on window mode over the regrouped plane set and on the r2 screen.
"""
import numpy as np
import pytest

import tomahawk_amd as T
from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu


def _in_window(want, W):
    d = np.abs(want["Apos"].astype(np.int64) - want["Bpos"].astype(np.int64))
    return want[(want["ridA"] == want["ridB"]) & (d <= W)]


def test_config2_phased_100k_samples_multi_tile(hip):
    """configs[1]'s sample count with enough variants for several row / column tiles: the default launch
    (one diagonal super-tile of 3 x 3 blocks) and 128-variant super-tiles (diagonal + rectangle launches,
    two-deep pipeline) both equal the oracle, counts bit-exact."""
    N, M = 100_000, 333
    al = util.mosaic_alleles(M, N, 71, n_founders=8, switch=0.02, mut=0.003)
    data, mask, variants = util.upload(hip, al)
    want = O.all_pairs(data, mask, variants, N, O.settings(minR2=0.0, phased=True), vector_only=False)
    assert len(want) > 50_000
    for tile in (0, 128):
        got, npairs, nrec = hip.ld_all(T.MODE_PHASED, T.Filters(minR2=0.0), tile_variants=tile)
        assert npairs == M * (M - 1) // 2 and nrec == len(got)
        util.assert_records_match(got, want, variants)
    # the default filter (r2 >= 0.1) keeps the same pairs as the oracle's
    want01 = O.all_pairs(data, mask, variants, N, O.settings(minR2=0.1, phased=True), vector_only=False)
    got01, _, _ = hip.ld_all(T.MODE_PHASED, T.Filters(minR2=0.1), tile_variants=128)
    assert 100 < len(want01) < len(want)
    util.assert_records_match(got01, want01, variants)
    # raw 2x2 tables of a rectangle that crosses tile boundaries
    c = hip.count_tile(T.MODE_PHASED, 100, 150, 120, 213)
    for i, j in [(0, 0), (149, 212), (27, 8), (28, 136), (140, 9)]:
        assert np.array_equal(c[i, j], O.count_phased(data[100 + i], None, data[120 + j], None, N)), (i, j)


def test_fisher_shortcut_unphased_1m_samples(hip):
    """Fisher's exact test behind UnphasedMath at N = 1 M: tables are round()ed expected haplotype counts
    near 2e6 (ld_engine.cpp:1656), the reference walks ~1e6 tail terms per record, the device starts at
    a verified e^-50 point.  P must agree from ~1 down to the underflow region."""
    N, M = 1_000_000, 26
    rng = np.random.default_rng(99)
    al = util.random_alleles(M, N, 99, maf_lo=0.1, maf_hi=0.5)
    for k, noise in enumerate([0.0003, 0.02, 0.2, 0.4, 0.45, 0.47, 0.48, 0.485, 0.49, 0.495, 0.497, 0.499]):
        flip = rng.random((N, 2)) < noise
        al[M - 1 - k] = np.where(flip, 1 - al[k], al[k])
    data, mask, variants = util.upload(hip, al, phase=0)
    vet = util.double_root_vetter(data, mask, variants, N)
    want = O.all_pairs(data, mask, variants, N, O.settings(minR2=0.0, unphased=True), vector_only=False)
    got, npairs, _ = hip.ld_all(T.MODE_UNPHASED, T.Filters(minR2=0.0))
    assert npairs == M * (M - 1) // 2 and len(want) > 300
    assert (want["controller"] & 1).sum() == 0            # every record came through the cubic, none through PhasedMath
    util.assert_records_match(got, want, variants, double_root=vet)
    P = np.sort(want["P"])
    assert P[0] < 1e-250 and (P > 0.3).sum() > 50 and ((P > 1e-200) & (P < 1e-6)).sum() >= 2
    # and the P cut-off drops the same records
    want_p = O.all_pairs(data, mask, variants, N, O.settings(minR2=0.0, minP=1e-6, unphased=True), vector_only=False)
    got_p, _, _ = hip.ld_all(T.MODE_UNPHASED, T.Filters(minR2=0.0, minP=1e-6))
    assert 0 < len(want_p) < len(want)
    util.assert_records_match(got_p, want_p, variants, double_root=vet)


@pytest.fixture(scope="module")
def ten_million():
    """configs[4]'s sample count: 14 haplotype-block variants, 10,000,000 diploid samples, a third of the
    variants with 0.5 % missing samples.  Counts reach 2e7 (the int arguments of kt_fisher_exact,
    fisher_math.cpp:231), a row is 19,532 K-chunks of the count kernel."""
    N, M = 10_000_000, 14
    rng = np.random.default_rng(4)
    founders = (rng.random((5, M)) < rng.uniform(0.15, 0.6, size=M)[None, :]).astype(np.int8)
    al = np.empty((M, N, 2), dtype=np.int8)
    cur = rng.integers(0, 5, size=2 * N).astype(np.int8)
    for v in range(M):
        sw = rng.random(2 * N) < 0.06
        cur = np.where(sw, rng.integers(0, 5, size=2 * N), cur).astype(np.int8)
        h = founders[cur, v] ^ (rng.random(2 * N) < 0.008)
        al[v] = h.reshape(N, 2)
    for v in (1, 4, 7, 12):
        ms = rng.random(N) < 0.005
        al[v, ms, :] = 2
    al[9, 17, 0] = 2                                     # one half-missing genotype (an = 1)
    data, mask = O.bitvectors_from_alleles(al)
    variants = O.variants_from_alleles(al)
    return N, M, data, mask, variants


@pytest.mark.parametrize("mode,ph", [(T.MODE_UNPHASED, False), (T.MODE_PHASED, True), (T.MODE_AUTO, None)])
def test_config5_ten_million_samples_all_modes(hip, ten_million, mode, ph):
    N, M, data, mask, variants = ten_million
    hip.set_problem(N, M)
    hip.upload(data, util.to_hip_meta(variants), mask)
    vet = util.double_root_vetter(data, mask, variants, N)
    st = O.settings(minR2=0.0, phased=bool(ph), unphased=(ph is False))
    want = O.all_pairs(data, mask, variants, N, st, vector_only=False)
    got, npairs, _ = hip.ld_all(mode, T.Filters(minR2=0.0))
    assert npairs == M * (M - 1) // 2 and len(want) > 60
    assert want["cnt"].max() > 1e7
    util.assert_records_match(got, want, variants, double_root=vet)
    # configs[4] as specified: window + Fisher cut-off (positions 1000 + 100 v: +-4 partners at 400 bp)
    W = 400
    st = O.settings(minR2=0.0, minP=1e-6, phased=bool(ph), unphased=(ph is False))
    want_w = _in_window(O.all_pairs(data, mask, variants, N, st, vector_only=False), W)
    got_w, npw, _ = hip.ld_all(mode, T.Filters(minR2=0.0, minP=1e-6), window=T.OPT_WINDOW, l_window=W)
    assert npw == sum(min(4, M - 1 - i) for i in range(M))
    assert 10 < len(want_w) < len(want)
    util.assert_records_match(got_w, want_w, variants, double_root=vet)
    # raw cells, bit-exact, on the widest rows the engine sees
    if mode != T.MODE_AUTO:
        counter = O.count_phased if ph else O.count_unphased
        c = hip.count_tile(mode, 0, M, 0, M)
        for i, j in [(0, 13), (1, 4), (4, 12), (9, 1), (6, 5)]:
            mi = mask[i] if variants["gt_missing"][i] else None
            mj = mask[j] if variants["gt_missing"][j] else None
            assert np.array_equal(c[i, j], counter(data[i], mi, data[j], mj, N)), (i, j)


@pytest.mark.parametrize("min_chunks", [8, 1])
def test_config5_ten_million_samples_three_product_form(hip, opt, min_chunks):
    """configs[4]'s sample count without missing genotypes - the planes on which bench.py's extra.cfg5_shard runs the three-product
    form (k_count3_list_t: rows of 312,500 words, 9,766 K chunks, every tile cut along K and added with atomics): 20 variants from
    the planted generator (copies next to their sources, flip probabilities from ~0 to 0.45), three = 2 against three = 0 byte for
    byte and against the oracle, at the default cut-off, at the lowest cut-off the form takes (2e-6; unlinked pairs lie near
    5e-8 at this N: the noisiest copies pass, nothing else), and as configs[4] specifies it: window + Fisher cut-off."""
    N, M, seed = 10_000_000, 20, 5
    plant = T.Plant.near(M, 1, max_eps=0.45)
    hip.set_problem(N, M)
    hip.generate_synthetic(seed, plant=plant)
    data, _ = hip.download()
    for v in (0, 7, 19):                                   # the device's rows are the host twin's
        assert np.array_equal(data[v], T.synth_bitvector(seed, N, v, plant)[0]), v
    variants = np.zeros(M, dtype=O.VARIANT_DTYPE)
    variants["ac"] = hip.marginals()[0]; variants["pos"] = 1000 + 100 * np.arange(M); variants["hwe"] = 1.0; variants["gt_phase"] = 1
    vet = util.double_root_vetter(data, None, variants, N)
    opt.set("count_min_chunks", min_chunks)
    order = ["idxA", "idxB"]
    for minR2, minP, W in ((0.1, 1.0, 0), (2e-6, 1.0, 0), (0.05, 1e-6, 400)):
        st = O.settings(minR2=minR2, minP=minP, unphased=True)
        want = O.all_pairs(data, None, variants, N, st, vector_only=False)
        if W:
            want = _in_window(want, W)
        f = T.Filters(minR2=minR2, minP=minP)
        res = {}
        for three in (0, 2):
            opt.set("three", three)
            hip.timing_reset()
            res[three], _, _ = hip.ld_all(T.MODE_UNPHASED, f, window=T.OPT_WINDOW if W else 0, l_window=W)
            tm = hip.timing()
            assert (tm["three_launches"] > 0) == (three == 2) and tm["fused_launches"] == 0, (three, tm)
        assert np.sort(res[0], order=order).tobytes() == np.sort(res[2], order=order).tobytes(), (minR2, minP, W)
        assert len(want) == len(res[2]) >= 3, (minR2, len(want), len(res[2]))
        util.assert_records_match(res[2], want, variants, double_root=vet)
    opt.unset("three")


@pytest.mark.parametrize("tile", [0, 128])
def test_window_mode_regrouped_rows_straddle_contigs(hip, tile):
    """Default mode + missing data + `-w` on several contigs: the rows of the regrouped rectangle (variants
    with missing data) straddle contig boundaries and later contigs start at small positions again, so a
    column tile can lie wholly *before* the row tile's reach.  Every same-contig pair with |dpos| <= w must
    still come out (round-1 review: a tile skip assumed columns always follow rows)."""
    N, W = 128, 6000
    sizes = [350, 300, 250, 120]
    M = sum(sizes)
    al = util.random_alleles(M, N, 78, miss_rate=0.05, miss_variants=0.3, low_ac=3)
    rid = np.repeat(np.arange(len(sizes)), sizes).astype(np.uint32)
    pos = np.concatenate([np.arange(n) * 100 + off for n, off in zip(sizes, (1000, 500, 200, 50))])
    data, mask, variants = util.upload(hip, al, pos=pos, rid=rid)
    f = T.Filters(minR2=0.0)
    everything, _, _ = hip.ld_all(T.MODE_AUTO, f)
    d = np.abs(pos[everything["idxA"]].astype(np.int64) - pos[everything["idxB"]].astype(np.int64))
    want = everything[(rid[everything["idxA"]] == rid[everything["idxB"]]) & (d <= W)]
    got, npairs, nrec = hip.ld_all(T.MODE_AUTO, f, tile_variants=tile, window=T.OPT_WINDOW, l_window=W)
    assert nrec == len(got)
    a = np.sort(want, order=["idxA", "idxB"]); b = np.sort(got, order=["idxA", "idxB"])
    assert len(a) == len(b) and a.tobytes() == b.tobytes()
    # the pair count reported for window mode is the number of in-window pairs
    same = rid[:, None] == rid[None, :]
    near = np.abs(pos[:, None].astype(np.int64) - pos[None, :].astype(np.int64)) <= W
    assert npairs == int(np.triu(same & near, 1).sum())
    # shards still partition it
    parts = [hip.ld_all(T.MODE_AUTO, f, part=k, n_parts=3, tile_variants=tile, window=T.OPT_WINDOW, l_window=W) for k in range(3)]
    c = np.sort(np.concatenate([p[0] for p in parts]), order=["idxA", "idxB"])
    assert a.tobytes() == c.tobytes()


@pytest.mark.parametrize("N,M,seed", [(64, 120, 301), (1_000_000, 12, 302)])
def test_r2_screen_agrees_at_the_cutoff(hip, N, M, seed):
    """PhasedMath's integer screen (ld_math.hip.h) against the reference's rounded test, with the r2 cut-off
    placed exactly on, one ulp below and one ulp above the r2 of existing pairs: the survivors must be
    exactly the records whose (unscreened, minR2 = 0) R2 is >= the cut-off."""
    al = util.mosaic_alleles(M, N, seed, n_founders=5, switch=0.03, mut=0.004)
    util.upload(hip, al)
    base, _, _ = hip.ld_all(T.MODE_PHASED, T.Filters(minR2=0.0))
    r2 = np.unique(base["R2"])
    r2 = r2[(r2 > 1e-5) & (r2 < 1.0)]
    picks = r2[:: max(1, len(r2) // 9)][:9]
    assert len(picks) >= 5
    keys = lambda r: set(zip(r["idxA"].tolist(), r["idxB"].tolist()))
    for x in picks:
        for cut in (np.nextafter(x, 0.0), x, np.nextafter(x, 1.0)):
            got, _, _ = hip.ld_all(T.MODE_PHASED, T.Filters(minR2=float(cut)))
            want = base[base["R2"] >= cut]
            assert keys(got) == keys(want), (x, cut)


def test_option_bits_pass_through_unchanged(hip):
    """ld_all / ld_tile hand the TWK_HIP_OPT_* bits to the C ABI as given: OPT_KEEP_LOW_AC alone keeps the
    singleton pairs and does not switch window mode on."""
    N, M = 64, 40
    al = util.random_alleles(M, N, 17, low_ac=8)
    al[:8] = 0
    for v in range(8):
        al[v, v, 0] = 1                                   # eight singletons: ac_A + ac_B = 2 for their pairs
    util.upload(hip, al)
    f = T.Filters(minR2=0.0)
    plain, _, _ = hip.ld_all(T.MODE_PHASED, f)
    keep, _, _ = hip.ld_all(T.MODE_PHASED, f, window=T.OPT_KEEP_LOW_AC, l_window=1)
    win, _, _ = hip.ld_all(T.MODE_PHASED, f, window=T.OPT_WINDOW, l_window=1)
    keys = lambda r: set(zip(r["idxA"].tolist(), r["idxB"].tolist()))
    assert keys(plain) <= keys(keep)
    assert not any(a < 8 and b < 8 for a, b in keys(plain))
    # singleton x singleton tables have < 5 observations off the main cell: PhasedMath drops them anyway,
    # but the pairs with the other variants are unaffected and l_window = 1 did not filter anything
    assert len(keep) >= len(plain) > len(win)
    t_keep, _ = hip.ld_tile(T.MODE_PHASED, 0, M, 0, M, True, f, window=T.OPT_KEEP_LOW_AC, l_window=1)
    assert keys(t_keep) == keys(keep)


def test_inconsistent_missing_flags_are_rejected_without_side_effects(hip):
    """gt_missing without missing alleles (or the reverse) cannot come from the reference's importer; the
    upload refuses it before touching device state (round-1 review)."""
    N, M = 64, 20
    al = util.random_alleles(M, N, 19, miss_rate=0.1, miss_variants=0.5)
    data, mask, variants = util.upload(hip, al)
    before, _, _ = hip.ld_all(T.MODE_AUTO, T.Filters(minR2=0.0))
    bad = util.to_hip_meta(variants)
    k = int(np.nonzero(bad["missing"])[0][0])
    bad["an"][k] = 0
    with pytest.raises(T.HipError) as e:
        hip.upload(np.zeros_like(data), bad, mask)
    assert e.value.code == -1
    after, _, _ = hip.ld_all(T.MODE_AUTO, T.Filters(minR2=0.0))
    order = ["idxA", "idxB"]                    # survivors are compacted in no particular order
    assert np.sort(before, order=order).tobytes() == np.sort(after, order=order).tobytes()


# ---- T1 on the device: run-length genotypes -> bitvector + mask -------------------------------------------------
def _rle_encode(al, width, missing):
    """Run words of one variant (al int8 [N, 2] in {0,1,2}) the way a .twk stores them: length << (2+2m) | A << (1+m) | B,
    `width` bytes each, runs split at the longest length the word can hold (lib/genotype_encoder.h:277-343)."""
    m = 1 if missing else 0
    limit = (1 << (8 * width - 2 - 2 * m)) - 1
    code = (al[:, 0].astype(np.int64) << (m + 1)) | al[:, 1].astype(np.int64)
    starts = np.concatenate(([0], np.nonzero(np.diff(code))[0] + 1, [len(code)]))
    words, lens, As, Bs = [], [], [], []
    for s0, s1 in zip(starts[:-1], starts[1:]):
        left = int(s1 - s0)
        while left:
            ln = min(left, limit)
            words.append(ln << (2 + 2 * m) | int(code[s0]))
            lens.append(ln); As.append(int(al[s0, 0])); Bs.append(int(al[s0, 1]))
            left -= ln
    raw = np.array(words, dtype={1: "<u1", 2: "<u2", 4: "<u4"}[width]).view(np.uint8)
    return raw, np.array(lens, np.uint32), np.array(As, np.uint8), np.array(Bs, np.uint8)


@pytest.mark.parametrize("N", [1, 15, 16, 17, 127, 128, 129, 1000, 4099, 200_003])
def test_rle_inflate_kernel_equals_the_oracle(hip, N):
    """twk_hip_upload_rle (HIP kernel) against orc_build_bitvector (twk_igt_vec::Build, core.cpp:349-391), bit for
    bit, data and mask: every run-word width, with and without missing genotypes, all-ref / all-alt / alternating
    variants, long runs that are split at the word's length limit, sample counts around the word boundaries."""
    rng = np.random.default_rng(N)
    M = 24
    al = util.random_alleles(M, N, 500 + N, maf_lo=0.01, maf_hi=0.6, miss_rate=0.1, miss_variants=0.4)
    al[0] = 0; al[1] = 1                                  # one run each (split at the limit of narrow words)
    al[2, :, 0] = 0; al[2, :, 1] = 1                        # all het
    al[3] = (np.arange(N)[:, None] % 2).astype(np.int8)     # alternates every sample: N runs
    if N > 3:
        al[4] = 0; al[4, N // 2] = (2, 1)                   # a single half-missing genotype
        blocks = (np.arange(N) // max(1, N // 7)) % 3       # a handful of long runs incl. missing ones
        al[5, :, 0] = blocks; al[5, :, 1] = blocks
    variants = O.variants_from_alleles(al)
    chunks, desc, want_d, want_m = [], np.zeros(M, dtype=T.RLE_DESC_DTYPE), [], []
    off = 3                                                  # run words need no alignment
    for v in range(M):
        missing = bool((al[v] == 2).any())
        width = int(rng.choice([1, 2, 4]))
        raw, lens, As, Bs = _rle_encode(al[v], width, missing)
        chunks.append((off, raw))
        desc[v] = (off, len(lens), width, int(missing), 0)
        off += raw.size + int(rng.integers(0, 5))
        d = np.zeros(O.words64(N), np.uint64); mk = np.zeros(O.words64(N), np.uint64)
        assert O.lib().orc_build_bitvector(lens.ctypes.data, As.ctypes.data, Bs.ctypes.data, len(lens), N, d.ctypes.data, mk.ctypes.data) == 0
        want_d.append(d); want_m.append(mk)
    buf = np.zeros(off + 8, np.uint8)
    for o, raw in chunks:
        buf[o:o + raw.size] = raw
    hip.set_problem(N, M + 3)
    hip.upload_rle(buf, desc, util.to_hip_meta(variants), first=2)          # into the middle of the problem
    data, mask = hip.download(2, M)
    assert np.array_equal(data, np.array(want_d)) and np.array_equal(mask, np.array(want_m))
    # the same rows through the bitvector upload give the same engine state
    d2, m2 = O.bitvectors_from_alleles(al)
    assert np.array_equal(data, d2) and (m2 is None or np.array_equal(mask, m2))
    # corrupt runs (lengths do not add up) are refused
    bad = desc.copy(); bad["n_runs"][3] -= 1
    if N > 1:
        with pytest.raises(T.HipError) as e:
            hip.upload_rle(buf, bad, util.to_hip_meta(variants), first=2)
        assert e.value.code == -1
    bad = desc.copy(); bad["offset"][0] = buf.size                             # runs beyond the buffer
    with pytest.raises(T.HipError):
        hip.upload_rle(buf, bad, util.to_hip_meta(variants), first=2)


def test_split_units_on_short_rows(hip, opt):
    """The count kernel's split-unit path (K-ranges of a tile added with atomics) is only taken on long rows by
    default; forced onto short ones, every cell still equals the oracle's."""
    N, M = 3000, 300                                  # unphased planes: 94 words -> 3 chunks of 32
    al = util.random_alleles(M, N, 8, miss_rate=0.05, miss_variants=0.3)
    data, mask, variants = util.upload(hip, al)
    opt.set("count_min_chunks", 1)
    for mode, counter in ((T.MODE_PHASED, O.count_phased), (T.MODE_UNPHASED, O.count_unphased)):
        got = hip.count_tile(mode, 0, M, 0, M)
        rng = np.random.default_rng(3)
        for i, j in zip(rng.integers(0, M, 60), rng.integers(0, M, 60)):
            mi = mask[i] if variants["gt_missing"][i] else None
            mj = mask[j] if variants["gt_missing"][j] else None
            assert np.array_equal(got[i, j], counter(data[i], mi, data[j], mj, N)), (mode, i, j)
    whole, _, _ = hip.ld_all(T.MODE_AUTO, T.Filters(minR2=0.0))
    opt.unset("count_min_chunks")
    plain, _, _ = hip.ld_all(T.MODE_AUTO, T.Filters(minR2=0.0))
    order = ["idxA", "idxB"]
    assert np.sort(whole, order=order).tobytes() == np.sort(plain, order=order).tobytes()


# ---- the r2 screen (TWK_HIP_OPT_R2_SCREEN): same records, most of the contraction never done on cohort-shaped data ----
def _cohort_alleles(M, N, seed, miss=False):
    """Founder mosaics for a third of the variants, the rest rare with a 1/x allele-count spectrum carried by
    haplotypes of one founder (LD among rare variants and with their founder's common ones)."""
    rng = np.random.default_rng(seed)
    al = util.mosaic_alleles(M, N, seed, n_founders=6, switch=0.01, mut=0.002)
    H = 2 * N
    rare = rng.random(M) < 0.65
    flat = al.reshape(M, H)
    for v in np.nonzero(rare)[0]:
        ac = max(1, int(np.exp(rng.random() * np.log(0.02 * H))))
        donors = np.nonzero(flat[max(0, v - 1)] == 1)[0] if v and rng.random() < 0.7 else np.arange(H)
        if len(donors) < ac:
            donors = np.arange(H)
        flat[v] = 0
        flat[v, rng.choice(donors, size=ac, replace=False)] = 1
    al = flat.reshape(M, N, 2)
    if miss:
        for v in np.nonzero(rng.random(M) < 0.15)[0]:
            al[v, rng.random(N) < 0.03, :] = 2
            if not (al[v] == 2).any():
                al[v, 0, :] = 2
    return al


@pytest.mark.parametrize("mode,miss", [(T.MODE_PHASED, False), (T.MODE_UNPHASED, False), (T.MODE_AUTO, False), (T.MODE_AUTO, True)])
def test_r2_screen_gives_the_same_records(hip, mode, miss):
    N, M = 1500, 2600
    al = _cohort_alleles(M, N, 31 + int(miss), miss)
    data, mask, variants = util.upload(hip, al)
    order = ["idxA", "idxB"]
    for minR2 in (0.1, 0.5, 0.004):
        f = T.Filters(minR2=minR2)
        hip.timing_reset()
        plain, np0, _ = hip.ld_all(mode, f)
        work0 = hip.timing()["row_pairs"]
        hip.timing_reset()
        scr, np1, nr1 = hip.ld_all(mode, f, window=T.OPT_R2_SCREEN)
        work1 = hip.timing()["row_pairs"]
        assert np1 == np0 == M * (M - 1) // 2 and nr1 == len(scr) == len(plain) > 50
        a, b = np.sort(plain, order=order), np.sort(scr, order=order)
        assert a.tobytes() == b.tobytes()
        if minR2 >= 0.1 and not miss:
            assert work1 < 0.75 * work0                   # tiles outside the band were not contracted
    # the cut-off placed exactly on the r2 of existing pairs: the band is wide enough for the rounded test
    r2 = np.unique(plain["R2"]); r2 = r2[(r2 > 0.004) & (r2 < 1)]
    for x in r2[:: max(1, len(r2) // 6)][:6]:
        for cut in (np.nextafter(x, 0.0), x, np.nextafter(x, 1.0)):
            f = T.Filters(minR2=float(cut))
            p2, _, _ = hip.ld_all(mode, f)
            s2, _, _ = hip.ld_all(mode, f, window=T.OPT_R2_SCREEN)
            assert np.sort(p2, order=order).tobytes() == np.sort(s2, order=order).tobytes()
    # shards of the screened run partition it
    f = T.Filters(minR2=0.1)
    whole, _, _ = hip.ld_all(mode, f, window=T.OPT_R2_SCREEN)
    parts = [hip.ld_all(mode, f, part=k, n_parts=3, window=T.OPT_R2_SCREEN) for k in range(3)]
    assert sum(p[1] for p in parts) == M * (M - 1) // 2
    assert np.sort(np.concatenate([p[0] for p in parts]), order=order).tobytes() == np.sort(whole, order=order).tobytes()
    # and the oracle agrees with both (sampled: the oracle is scalar)
    sub = np.sort(np.random.default_rng(5).choice(M, size=260, replace=False))
    hip.set_problem(N, len(sub))
    hip.upload(data[sub], util.to_hip_meta(variants[sub]), None if mask is None else mask[sub])
    st = O.settings(minR2=0.1, phased=(mode == T.MODE_PHASED), unphased=(mode == T.MODE_UNPHASED))
    want = O.all_pairs(data[sub], None if mask is None else mask[sub], variants[sub], N, st, vector_only=False)
    got, _, _ = hip.ld_all(mode, T.Filters(minR2=0.1), window=T.OPT_R2_SCREEN)
    util.assert_records_match(got, want, variants[sub], double_root=util.double_root_vetter(data[sub], None if mask is None else mask[sub], variants[sub], N))


@pytest.mark.parametrize("mode", [T.MODE_PHASED, T.MODE_UNPHASED])
def test_r2_screen_ignores_a_wrong_ac_field(hip, mode):
    """The screen's bound is about the margins of the counted table, so it must come from the bits on the device: a
    .twk whose `ac` header field disagrees with its genotypes (here: every variant claims 7 ALT alleles, which would
    put the common variants' partners outside any band) gives the same records with and without the screen."""
    N, M = 1200, 1500
    al = _cohort_alleles(M, N, 77)
    data, mask, variants = util.upload(hip, al)
    liar = variants.copy()
    liar["ac"] = 7
    hip.set_problem(N, M)
    hip.upload(data, util.to_hip_meta(liar), mask)
    order = ["idxA", "idxB"]
    f = T.Filters(minR2=0.3)
    plain, _, _ = hip.ld_all(mode, f)
    scr, _, nrec = hip.ld_all(mode, f, window=T.OPT_R2_SCREEN)
    assert nrec == len(plain) > 200
    assert np.sort(plain, order=order).tobytes() == np.sort(scr, order=order).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("M", [40, 700, 2300])
def test_survivors_leave_the_device_in_pair_order(hip, M):
    """The survivors of a tile are compacted with an atomic counter, then put in (idxA, idxB) order on the device
    (key sort + gather) - the order the writer keeps, which makes a one-GPU run's file deterministic.  Records the
    Fisher cut-off drops are only marked by the math kernel: they sort behind the rest and are cut off.  Same
    records as the oracle either way."""
    N = 96
    al = util.random_alleles(M, N, 900 + M, miss_rate=0.04, miss_variants=0.25)
    data, mask, variants = util.upload(hip, al)
    for mode, filt in ((T.MODE_PHASED, T.Filters(minR2=0.0)), (T.MODE_UNPHASED, T.Filters(minR2=0.0, minP=0.05)),
                       (T.MODE_AUTO, T.Filters(minR2=0.02))):
        recs, npairs = hip.ld_tile(mode, 0, M, 0, M, True, filt)
        assert npairs == M * (M - 1) // 2
        key = recs["idxA"].astype(np.int64) << 32 | recs["idxB"].astype(np.int64)
        assert (np.diff(key) > 0).all()
        assert (recs["idxA"] < recs["idxB"]).all() and recs["idxB"].max() < M
        if M <= 700:
            st = O.settings(minR2=filt.minR2, minP=filt.minP, phased=mode == T.MODE_PHASED, unphased=mode == T.MODE_UNPHASED)
            want = O.all_pairs(data, mask, variants, N, st, vector_only=False)
            assert len(want) > 30
            util.assert_records_match(recs, want, variants, double_root=util.double_root_vetter(data, mask, variants, N))


@pytest.mark.gpu
def test_cubic_record_with_cancelling_d(hip):
    """The worst cubic-path record of every sweep so far (tests/sweeps/haplotype_block_sweep_large_n.py, data set 2):
    N = 20,000 with missing genotypes, a pair with r2 = 3e-9 whose D = -2.4e-7 is what is left of a cancelling
    difference - device and oracle agree on D to 3.6e-12, which is 1.5e-5 *relative*, and D' = D / dmax = 0.0063 and r
    carry that.  3.6e-12 is 0.42 of what one unit of rounding in the cubic's terms moves this root by (8.6e-12: the cubic's
    coefficients are O(n) = 8e4 here, its slope at the root 1.0), which is what the checker allows such a record - a few
    times its own root's uncertainty, propagated through its own dmax (tests/util.py cubic_floors); the whole data set is
    compared in both modes that reach the cubic."""
    N, M = 20000, 90
    al = util.mosaic_alleles(M, N, 7002, n_founders=6, switch=0.005, mut=0.0, miss_rate=0.02, miss_variants=0.3)
    data, mask, variants = util.upload(hip, al)
    for mode, st in ((T.MODE_UNPHASED, O.settings(minR2=0.0, unphased=True)), (T.MODE_AUTO, O.settings(minR2=0.0))):
        want = O.all_pairs(data, mask, variants, N, st, vector_only=False)
        got, _, _ = hip.ld_all(mode, T.Filters(minR2=0.0))
        vet = util.double_root_vetter(data, mask, variants, N)
        with pytest.raises(AssertionError):          # without the pair's own conditioning nothing excuses it
            util.assert_records_match(got, want, variants.copy(), count=False)      # (a copy: not the data set upload() remembers)
        util.assert_records_match(got, want, variants, double_root=vet)
        g = util.records_by_pair(got, "idxA", "idxB")[(33, 89)]
        assert g["R2"] < 1e-8 and abs(g["D"]) < 1e-6 and 1e-3 < abs(g["Dprime"]) < 1e-2


@pytest.mark.gpu
def test_bench_orchestration_with_two_ranks_on_one_gpu(tmp_path):
    """bench.py as the driver launches it for N > 1 (python -m torch.distributed.run, one process per rank), with the
    gloo backend so that both ranks can share this box's one GPU: equal-area row bands, survivors gathered to rank 0
    (counts all-gather + exact-size point-to-point transfers, tomahawk_amd/dist.py), rank 0 packs them into a real .two
    inside the timed region, one JSON line.  (RCCL itself needs one GPU per rank: the 8-GPU run is the driver's.)"""
    import json
    import os
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--variants", "4096",
           "--samples", "50000", "--min-r2", "0.00005", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    M = 4096
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["survivors_per_step"] > 1000
    assert d["config"]["two_records_written_per_step"] == 2 * d["config"]["survivors_per_step"]
    assert f"{M * (M - 1) // 2} pairs/step" in d["config"]["workload"]
    assert d["roofline"]["bound"] == "valu" and 0 < d["roofline"]["frac"] < 1
    assert d["config"]["collective_backend"] == "gloo"
    # the default backend with RCCL made to fail on every rank: the run falls back to gloo and says so
    cmd2 = [c for c in cmd if c not in ("--backend", "gloo")]
    cmd2[cmd2.index("--master-port") + 1] = str(port + 1 if port < 65000 else port - 1)
    r = subprocess.run(cmd2, capture_output=True, text=True, timeout=600, cwd=root, env=dict(os.environ, TWK_BENCH_FORCE_RCCL_FAILURE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    d2 = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d2["config"]["collective_backend"].startswith("gloo (RCCL failed") and d2["config"]["survivors_per_step"] == d["config"]["survivors_per_step"]


def _run_bench(tmp_path, world, extra, tag, backend="gloo"):
    """bench.py as the driver launches it (python -m torch.distributed.run for N > 1; gloo so that the ranks can share
    this box's one GPU, nccl = RCCL with one GPU per rank) -> (JSON line, records of the .two rank 0 wrote in the last
    timed step)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from tomahawk_amd import hostlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    two = str(tmp_path / f"{tag}_{world}.two")
    common = ["--gpus", str(world), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--no-traffic", "--keep-two", two] + extra
    if world == 1:
        cmd = [sys.executable, os.path.join(root, "bench.py")] + common
    else:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(root, "bench.py"), "--backend", backend] + common
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]          # ONE JSON line and nothing else on stdout, whatever the libraries print
    recs, info = hostlib.read_two(two)
    return json.loads(lines[0]), recs


@pytest.mark.parametrize("config,extra", [
    ("cfg3", ["--samples", "20000", "--variants", "4096", "--min-r2", "0.0002"]),
    # configs[4]'s slab path: +-500 kb at 100 bp spacing = 5,000 partners per variant, i.e. more than two of the eight
    # bands of 16,384 variants wide - every rank's halo reaches beyond its neighbour's band
    ("cfg5", ["--samples", "20000", "--variants", "16384", "--min-r2", "0.0002", "--min-p", "1e-3"]),
])
def test_bench_with_eight_ranks_on_one_gpu_equals_one_rank(tmp_path, config, extra):
    """The N > 1 code of bench.py - equal-area row bands (cfg3) and window-mode slabs with halos (cfg5), survivors kept
    in HBM (twk_hip_set_device_sink), gathered to rank 0 and packed into one .two - with the world size of the driver's
    8-GPU run, on this box's one GPU over gloo: the file rank 0 writes holds exactly the records of the 1-rank file.
    (What this cannot show is RCCL itself: that needs one GPU per rank.)"""
    args = ["--config", config] + extra
    d1, one = _run_bench(tmp_path, 1, args, config)
    d8, eight = _run_bench(tmp_path, 8, args, config)
    assert d8["n_gpus"] == 8 and d8["ranks_seen"] == list(range(8)) and len(d8["per_rank_ms"]) == 8
    assert d8["gather_ms"] >= 0 and d8["write_ms"] > 0 and all(x > 0 for x in d8["per_rank_ms"])
    assert 0 < d8["gather_bytes"] <= 104 * d8["config"]["survivors_per_step"] and d8["gather_GBps"] > 0
    assert d8["config"]["collective_backend"] == "gloo"
    # the survivor-rich self-check that follows the timed region of every N > 1 run (bench.py gather_check): what the driver's
    # multi-GPU run will show of the gather even though the headline workload has no survivors to send
    gc = d8["extra"]["gather_check"]
    assert gc["equal"] is True and gc["equal_per_rank"] == [True] * 8 and gc["ranks_seen"] == list(range(8)), gc
    assert gc["records"] == sum(gc["records_per_rank"]) > 10_000 and gc["bytes"] == 104 * (gc["records"] - gc["records_per_rank"][0]) and gc["GBps"] > 0
    assert gc["pairs"] == 4096 * 4095 // 2 and gc["wall_s"] < 20 and gc["backend"] == "gloo"
    assert "gather_check" not in d1.get("extra", {})
    # the survivor-bearing steps on the same problem with planted LD (extra.<config>_planted): same pairs found by one rank and by eight,
    # the three-product form's records the four-product form's, and the eight ranks' survivors went through the gather
    p1, p8 = d1["extra"][f"{config}_planted"], d8["extra"][f"{config}_planted"]
    for pp in (p1, p8):
        assert pp["records_equal_four_product"] is True and pp["planted_found"] == pp["planted_expected"] > 100, pp
        assert pp["four_product_step"]["three_product_launches"] == 0
    assert p1["planted_found"] == p8["planted_found"] and p1["survivors_per_step"] == p8["survivors_per_step"] > 0
    assert p8["gather_bytes_per_step"] > 0 and p1["gather_bytes_per_step"] == 0
    assert d1["config"]["survivors_per_step"] == d8["config"]["survivors_per_step"] > 1000
    assert len(one) == len(eight) == 2 * d1["config"]["survivors_per_step"]
    order = ["ridA", "packA", "ridB", "packB"]
    assert np.sort(one, order=order).tobytes() == np.sort(eight, order=order).tobytes()


@pytest.mark.parametrize("config,extra", [
    ("cfg3", ["--samples", "20000", "--variants", "4096", "--min-r2", "0.0002"]),
    ("cfg5", ["--samples", "20000", "--variants", "16384", "--min-r2", "0.0002", "--min-p", "1e-3"]),
    # survivor-rich: tens of MB per rank through the gather (the sender's payload is the engine's own HBM buffer, which
    # the next step rewrites - torn records here would mean the transfer was not finished when gather_records returned)
    ("cfg3", ["--samples", "2000", "--variants", "4096", "--min-r2", "0.0005", "--steps", "3"]),
])
def test_rccl_gather_between_gpus_equals_one_rank(tmp_path, config, extra):
    """Runs wherever the box shows >= 2 GPUs (skipped, with the reason, on the one-GPU boxes of this pool): bench.py under
    torch.distributed.run with the nccl backend (= RCCL), world = min(8, GPUs), one GPU per rank - the survivors leave every
    rank's HBM over RCCL (counts all-gather + grouped send/recv of exact sizes) straight into rank 0's buffer.  The line
    must say RCCL carried the gather and connected every rank, and the .two rank 0 wrote must hold exactly the records
    of the 1-rank file.  Partition: SURVEY 8(e); reference analogue lib/ld/ld_balancing.h:23-80."""
    import torch
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip(f"RCCL between GPUs needs >= 2 devices; this box shows {n_dev} (the check runs the first time a multi-GPU box does)")
    world = min(8, n_dev)
    args = ["--config", config] + extra
    d1, one = _run_bench(tmp_path, 1, args, config)
    dn, many = _run_bench(tmp_path, world, args, config, backend="nccl")
    assert dn["n_gpus"] == world and dn["config"]["collective_backend"] == "nccl", dn["config"]["collective_backend"]
    assert dn["ranks_seen"] == list(range(world)) and len(dn["per_rank_ms"]) == world
    assert "RCCL gather" in dn["config"]["partition"]
    assert d1["config"]["survivors_per_step"] == dn["config"]["survivors_per_step"] > 1000
    assert dn["gather_bytes"] > 0 and dn["gather_GBps"] and dn["gather_GBps"] > 0
    gc = dn["extra"]["gather_check"]
    assert gc["equal"] is True and gc["backend"] == "nccl" and gc["ranks_seen"] == list(range(world)) and gc["bytes"] > 0, gc
    assert len(one) == len(many) == 2 * d1["config"]["survivors_per_step"]
    order = ["ridA", "packA", "ridB", "packB"]
    assert np.sort(one, order=order).tobytes() == np.sort(many, order=order).tobytes()


def test_rccl_itself_on_this_box(tmp_path):
    """What one GPU can show of RCCL.  (1) A world of one: the bench's own init_groups("nccl") creates the RCCL group on
    cuda:0 and the collectives of the N > 1 path (all_gather of counts / rank ids, barrier) run on it.  (2) Two ranks on
    the one GPU with the default backend: the precondition of init_groups (one GPU per rank, agreed over the control group
    before anyone enters RCCL, which would refuse the duplicate device) sends both ranks to gloo together, the JSON line
    says why, and the run completes with both ranks seen.  (On a box with several GPUs the two ranks get one each and
    RCCL carries the gather.)"""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "sweeps", "rccl_probe.py")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, MASTER_PORT=str(port)))
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "init_groups -> nccl cuda:0" in r.stdout and "collectives on the RCCL group ok: 0 5" in r.stdout, r.stdout[-1000:]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--variants", "4096", "--samples", "50000", "--min-r2", "0.00005", "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    import torch
    if torch.cuda.device_count() >= 2:
        assert d["config"]["collective_backend"] == "nccl" and d["ranks_seen"] == [0, 1]
    else:
        assert d["config"]["collective_backend"].startswith("gloo (RCCL not attempted: ranks [0, 1] share one GPU") and d["ranks_seen"] == [0, 1]
    assert d["config"]["survivors_per_step"] > 1000 and d["config"]["two_records_written_per_step"] == 2 * d["config"]["survivors_per_step"]
