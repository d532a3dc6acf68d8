"""Rare variants as carrier lists (ld_list.hip.h: k_build_lists + k_list_screen): the device's answer to the reference's
twk_igt_list / PhasedListVector (include/core.h:517-672, lib/ld/ld_engine.cpp:185-267) for very large sample counts.  The
pairs of the list zone are intersected instead of contracted; the records must be those of the dense path, bit for bit,
and those of the oracle."""
import numpy as np
import pytest

import tomahawk_amd as T
from oracle import oracle as O
from tests import util
from tests.test_gpu_configs import _cohort_alleles

pytestmark = pytest.mark.gpu
ORDER = ["idxA", "idxB"]


def _with_flips(al, seed):
    """Turn a few rare variants into their complements (ALT frequency near 1: the minor allele is REF)."""
    rng = np.random.default_rng(seed)
    M = al.shape[0]
    ac = (al == 1).sum(axis=(1, 2))
    rare = np.nonzero((ac > 0) & (ac < 40))[0]
    for v in rng.choice(rare, size=min(len(rare) // 6, 60), replace=False):
        al[v] = 1 - al[v]
    return al


def _run(hip, opt, lists_env, data, mask, variants, calls):
    opt.set("lists", int(lists_env))
    opt.set("probe_zone", 0)            # these tests are about the merges: the zone's own pairs stay with them
                                        # (test_probe_pass_equals_dense_and_merges_only covers the default)
    N = variants_n = None
    hip.set_problem(_run.N, len(variants))
    hip.upload(data, util.to_hip_meta(variants), mask)
    out = []
    for call in calls:
        hip.timing_reset()
        recs = call()
        out.append((recs, hip.timing()))
    opt.unset("lists"); opt.unset("probe_zone")
    return out


@pytest.mark.parametrize("N,forced", [(66_000, False), (1500, True)])
def test_list_zone_equals_dense_and_oracle(hip, opt, N, forced):
    """N = 66,000: rows of 4,128 words, lists of up to 32 carriers are kept by default.  N = 1,500 with option "lists" = 2:
    the list pass next to the fused count kernel (short rows), whose epilogue must leave the zone's pairs alone."""
    M = 1500 if not forced else 2200
    al = _with_flips(_cohort_alleles(M, N, 900 + N), 5)
    data, mask = O.bitvectors_from_alleles(al)
    variants = O.variants_from_alleles(al)
    _run.N = N
    calls = [lambda: hip.ld_all(T.MODE_PHASED, T.Filters(minR2=0.1), window=T.OPT_R2_SCREEN)[0],
             lambda: hip.ld_all(T.MODE_AUTO, T.Filters(minR2=0.5), window=T.OPT_R2_SCREEN)[0],
             lambda: hip.ld_all(T.MODE_PHASED, T.Filters(minR2=0.004), window=T.OPT_R2_SCREEN)[0],
             lambda: np.concatenate([hip.ld_all(T.MODE_PHASED, T.Filters(minR2=0.1), part=k, n_parts=3, window=T.OPT_R2_SCREEN)[0] for k in range(3)]),
             lambda: hip.ld_all(T.MODE_PHASED, T.Filters(minR2=0.1, minP=1e-8), window=T.OPT_R2_SCREEN)[0]]
    dense = _run(hip, opt, "0", data, None, variants, calls)
    lists = _run(hip, opt, "2" if forced else "1", data, None, variants, calls)
    for k, ((a, ta), (b, tb)) in enumerate(zip(dense, lists)):
        assert ta["list_launches"] == 0 and tb["list_launches"] > 0 and tb["list_pairs"] > 10_000, (k, tb)
        assert len(a) == len(b) > 20, k
        assert np.sort(a, order=ORDER).tobytes() == np.sort(b, order=ORDER).tobytes(), k
        if k == 0:
            assert tb["row_pairs"] < ta["row_pairs"]          # the zone's tiles were not contracted
    # the oracle on the rarest variants (the zone) plus a sample of the rest
    ac = np.minimum(variants["ac"], 2 * N - variants["ac"])
    rare = np.argsort(ac, kind="stable")[:300]          # (a zone needs two tiles of variants: 256)
    sub = np.sort(np.concatenate([rare, np.random.default_rng(1).choice(np.setdiff1d(np.arange(M), rare), size=60, replace=False)]))
    opt.set("lists", 2)
    hip.set_problem(N, len(sub))
    hip.upload(data[sub], util.to_hip_meta(variants[sub]), None)
    hip.timing_reset()
    got, _, _ = hip.ld_all(T.MODE_PHASED, T.Filters(minR2=0.05), window=T.OPT_R2_SCREEN)
    assert hip.timing()["list_launches"] + hip.timing()["probe_launches"] > 0      # (the zone's pairs: probes where the row's list is short, merges else)
    opt.unset("lists")
    want = O.all_pairs(data[sub], None, variants[sub], N, O.settings(minR2=0.05, phased=True), vector_only=False)
    assert len(want) > 20
    util.assert_records_match(got, want, variants[sub])


def test_list_zone_with_missing_data_in_default_mode(hip, opt):
    """Default mode with missing genotypes: the screened stage runs over the missing-free head of the sorted set - lists
    there - and the variants with missing data go through the masked unphased planes as before."""
    N, M = 66_000, 1900
    al = _cohort_alleles(M, N, 77, miss=True)
    data, mask = O.bitvectors_from_alleles(al)
    variants = O.variants_from_alleles(al)
    _run.N = N
    calls = [lambda: hip.ld_all(T.MODE_AUTO, T.Filters(minR2=0.2), window=T.OPT_R2_SCREEN)[0]]
    (a, ta), = _run(hip, opt, "0", data, mask, variants, calls)
    (b, tb), = _run(hip, opt, "1", data, mask, variants, calls)
    assert tb["list_launches"] > 0 and ta["list_launches"] == 0 and len(a) == len(b) > 20
    assert np.sort(a, order=ORDER).tobytes() == np.sort(b, order=ORDER).tobytes()


def test_list_pass_with_a_tiny_survivor_buffer(hip, opt):
    """option "record_cap" (test hook) makes every buffer overflow: the list pass then takes fewer rows per launch, down to one
    row with a buffer grown to the zone's width, and the dense tiles go through their strip redo - the same records."""
    N, M = 1500, 1600
    al = _cohort_alleles(M, N, 33)
    data, mask = O.bitvectors_from_alleles(al)
    variants = O.variants_from_alleles(al)
    _run.N = N
    calls = [lambda: hip.ld_all(T.MODE_PHASED, T.Filters(minR2=0.05), window=T.OPT_R2_SCREEN)[0]]
    (a, ta), = _run(hip, opt, "2", data, None, variants, calls)
    opt.set("record_cap", 40)
    (b, tb), = _run(hip, opt, "2", data, None, variants, calls)
    opt.unset("record_cap")
    assert ta["list_launches"] > 0 and tb["list_launches"] > ta["list_launches"] and len(a) == len(b) > 500
    assert np.sort(a, order=ORDER).tobytes() == np.sort(b, order=ORDER).tobytes()


@pytest.mark.parametrize("N,forced", [(66_000, False), (1500, True)])
def test_unphased_list_zone_equals_dense_and_oracle(hip, opt, N, forced):
    """`-u`: the lists hold samples with their genotype (het / rare homozygote), the merge sorts common samples into four
    counters, HH / HQ / QH / QQ follow by which of the two variants has ALT as its major allele; candidates go through the
    unphased list math kernel.  Same checks as the phased zone, incl. variants turned into their complements."""
    M = 1500 if not forced else 2200
    al = _with_flips(_cohort_alleles(M, N, 400 + N), 6)
    data, mask = O.bitvectors_from_alleles(al)
    variants = O.variants_from_alleles(al)
    _run.N = N
    mode = T.MODE_UNPHASED
    calls = [lambda: hip.ld_all(mode, T.Filters(minR2=0.1), window=T.OPT_R2_SCREEN)[0],
             lambda: hip.ld_all(mode, T.Filters(minR2=0.5), window=T.OPT_R2_SCREEN)[0],
             lambda: hip.ld_all(mode, T.Filters(minR2=0.01), window=T.OPT_R2_SCREEN)[0],
             lambda: np.concatenate([hip.ld_all(mode, T.Filters(minR2=0.1), part=k, n_parts=3, window=T.OPT_R2_SCREEN)[0] for k in range(3)])]
    dense = _run(hip, opt, "0", data, None, variants, calls)
    lists = _run(hip, opt, "2" if forced else "1", data, None, variants, calls)
    for k, ((a, ta), (b, tb)) in enumerate(zip(dense, lists)):
        assert ta["list_launches"] == 0 and tb["list_launches"] > 0 and tb["list_pairs"] > 10_000, (k, tb)
        assert len(a) == len(b) > 20, k
        assert np.sort(a, order=ORDER).tobytes() == np.sort(b, order=ORDER).tobytes(), k
    ac = np.minimum(variants["ac"], 2 * N - variants["ac"])
    rare = np.argsort(ac, kind="stable")[:300]
    sub = np.sort(np.concatenate([rare, np.random.default_rng(2).choice(np.setdiff1d(np.arange(M), rare), size=60, replace=False)]))
    opt.set("lists", 2)
    hip.set_problem(N, len(sub))
    hip.upload(data[sub], util.to_hip_meta(variants[sub]), None)
    hip.timing_reset()
    got, _, _ = hip.ld_all(mode, T.Filters(minR2=0.05), window=T.OPT_R2_SCREEN)
    assert hip.timing()["list_launches"] + hip.timing()["probe_launches"] > 0      # (probes where the row's list is short, merges else)
    opt.unset("lists")
    want = O.all_pairs(data[sub], None, variants[sub], N, O.settings(minR2=0.05, unphased=True), vector_only=False)
    assert len(want) > 20
    util.assert_records_match(got, want, variants[sub], n_samples=N, double_root=util.double_root_vetter(data[sub], None, variants[sub], N))


@pytest.mark.parametrize("mode,N,forced", [(T.MODE_PHASED, 66_000, False), (T.MODE_UNPHASED, 66_000, False), (T.MODE_PHASED, 1500, True), (T.MODE_UNPHASED, 1500, True)])
def test_probe_pass_equals_dense_and_merges_only(hip, opt, mode, N, forced):
    """K1's asymmetric path (lib/ld/ld_engine.cpp:230-242: walk the shorter carrier list, test the partner's bitvector): the
    pairs of a zone variant with a variant that keeps no list are probes of its carriers into the partner's plane row(s)
    (k_probe_screen / k_probe_screen_unphased) instead of dense contractions.  Three ways - no lists at all, merges inside the
    zone only (option probe = 0), merges + probes (with the probing rows taking the zone's own columns too, the default, or
    only the columns beyond the zone: option probe_zone) - the same records bit for bit, for several cut-offs (band widths), shards,
    a Fisher cut-off, variants turned into their complements (REF-minor lists), and with a survivor buffer that overflows; with
    probes no tile row inside the zone is contracted."""
    M = 1500 if not forced else 2200
    al = _with_flips(_cohort_alleles(M, N, 1300 + N + mode), 7)
    data, mask = O.bitvectors_from_alleles(al)
    variants = O.variants_from_alleles(al)
    calls = [lambda: hip.ld_all(mode, T.Filters(minR2=0.1), window=T.OPT_R2_SCREEN)[0],
             lambda: hip.ld_all(mode, T.Filters(minR2=0.004), window=T.OPT_R2_SCREEN)[0],
             lambda: np.concatenate([hip.ld_all(mode, T.Filters(minR2=0.02), part=k, n_parts=3, window=T.OPT_R2_SCREEN)[0] for k in range(3)]),
             lambda: hip.ld_all(mode, T.Filters(minR2=0.05, minP=1e-6), window=T.OPT_R2_SCREEN)[0],
             # below the cut-off the r2 screen needs to be "worth the name" (1e-3): without lists no sorted set at all, with
             # them the zone's pairs are still merged and probed (calc -r 0.0009 at N = 1 M: DESIGN 3.5)
             lambda: hip.ld_all(mode, T.Filters(minR2=0.0007), window=T.OPT_R2_SCREEN)[0]]

    def run(lists, probe, cap=0, probe_zone=1, probe_lds=1):
        opt.set("lists", lists); opt.set("probe", probe); opt.set("record_cap", cap); opt.set("probe_zone", probe_zone); opt.set("probe_lds", probe_lds)
        hip.set_problem(N, M)
        hip.upload(data, util.to_hip_meta(variants), None)
        out = []
        for call in calls:
            hip.timing_reset()
            recs = call()
            out.append((np.sort(recs, order=ORDER).tobytes(), len(recs), hip.timing()))
        return out

    dense = run(0, 0)
    merges = run(2 if forced else 1, 0)
    probes = run(2 if forced else 1, 1)
    tiny = run(2 if forced else 1, 1, cap=60)
    # probes for the columns beyond the zone only: the zone's own pairs are merges of two lists (round 4's first form)
    outer = run(2 if forced else 1, 1, probe_zone=0)
    # the probes as gathers from L2 instead of through LDS (k_probe_lds_t, the default): one column per block (the round-4 kernels)
    single = run(2 if forced else 1, 1, probe_lds=0)
    single_outer = run(2 if forced else 1, 1, probe_zone=0, probe_lds=0)
    for k, (d, m, p, t, o, s1, s2) in enumerate(zip(dense, merges, probes, tiny, outer, single, single_outer)):
        assert d[1] > 20 and d[0] == m[0] == p[0] == t[0] == o[0] == s1[0] == s2[0], (k, d[1], m[1], p[1], t[1], o[1], s1[1], s2[1])
        assert s1[2]["probe_pairs"] == p[2]["probe_pairs"] and s1[2]["candidates"] == p[2]["candidates"], (k, s1[2], p[2])
        assert o[2]["list_pairs"] > p[2]["list_pairs"] and o[2]["probe_pairs"] < p[2]["probe_pairs"], (k, o[2], p[2])
        assert d[2]["list_launches"] == 0 and d[2]["probe_launches"] == 0
        assert m[2]["list_launches"] > 0 and m[2]["probe_launches"] == 0
        assert p[2]["probe_launches"] > 0 and p[2]["probe_pairs"] > 1000, (k, p[2])
        assert p[2]["row_pairs"] <= m[2]["row_pairs"] < d[2]["row_pairs"], (k, p[2]["row_pairs"], m[2]["row_pairs"], d[2]["row_pairs"])
        if k == 1:       # a wide band: some tile row of the zone had tiles beyond the zone, now probed
            assert p[2]["row_pairs"] < m[2]["row_pairs"], (p[2]["row_pairs"], m[2]["row_pairs"])
        assert t[2]["probe_launches"] >= p[2]["probe_launches"]             # (an overflowing buffer makes it take fewer rows per launch)
