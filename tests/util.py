"""Shared helpers for the parity tests."""
import numpy as np

from oracle import oracle as O
import tomahawk_amd as T
from tomahawk_amd.hip import META_DTYPE


def random_alleles(M, N, seed, maf_lo=0.05, maf_hi=0.5, miss_rate=0.0, miss_variants=0.0, low_ac=0):
    """int8 [M, N, 2] genotypes; a fraction `miss_variants` of variants gets `miss_rate` missing samples;
    the first `low_ac` variants get a handful of ALT alleles only (singletons / doubletons)."""
    rng = np.random.default_rng(seed)
    p = rng.uniform(maf_lo, maf_hi, size=M)
    al = (rng.random((M, N, 2)) < p[:, None, None]).astype(np.int8)
    for v in range(min(low_ac, M)):
        al[v] = 0
        k = 1 + v % 4
        idx = rng.choice(N * 2, size=k, replace=False)
        al[v].reshape(-1)[idx] = 1
    if miss_variants > 0:
        which = rng.random(M) < miss_variants
        for v in np.nonzero(which)[0]:
            ms = rng.random(N) < miss_rate
            if not ms.any():
                ms[rng.integers(N)] = True
            al[v, ms, :] = 2
    return al


def mosaic_alleles(M, N, seed, n_founders=6, switch=0.03, mut=0.004, miss_rate=0.0, miss_variants=0.0):
    """Genotypes with real LD structure: every haplotype is a mosaic of a few founder haplotypes (switching
    founder with probability `switch` per variant) plus rare mutations - long runs of perfectly or almost
    perfectly correlated variants, identical variants, D' = 1 pairs, the cases iid data never produces."""
    rng = np.random.default_rng(seed)
    founders = (rng.random((n_founders, M)) < rng.uniform(0.1, 0.6, size=M)[None, :]).astype(np.int8)
    H = 2 * N
    src = rng.integers(0, n_founders, size=H)
    hap = np.empty((H, M), dtype=np.int8)
    cur = src.copy()
    for v in range(M):
        sw = rng.random(H) < switch
        cur = np.where(sw, rng.integers(0, n_founders, size=H), cur)
        hap[:, v] = founders[cur, v]
    hap ^= (rng.random((H, M)) < mut).astype(np.int8)
    for v in range(M):                       # keep every site polymorphic (the importer drops invariant sites)
        ac = int(hap[:, v].sum())
        if ac < 2 or ac > H - 2:
            hap[rng.choice(H, size=3, replace=False), v] = 1 if ac < 2 else 0
    al = hap.T.reshape(M, N, 2).copy()
    if miss_variants > 0:
        which = rng.random(M) < miss_variants
        for v in np.nonzero(which)[0]:
            ms = rng.random(N) < miss_rate
            if not ms.any():
                ms[rng.integers(N)] = True
            al[v, ms, :] = 2
    return al


def to_hip_meta(variants):
    m = np.zeros(len(variants), dtype=META_DTYPE)
    for k in ("ac", "an", "pos", "rid", "hwe"):
        m[k] = variants[k]
    m["missing"] = variants["gt_missing"]
    return m


def upload(hip, alleles, variants=None, **kw):
    M, N, _ = alleles.shape
    data, mask = O.bitvectors_from_alleles(alleles)
    if variants is None:
        variants = O.variants_from_alleles(alleles, **kw)
    hip.set_problem(N, M)
    hip.upload(data, to_hip_meta(variants), mask)
    return data, mask, variants


def records_by_pair(recs, key_a, key_b):
    return {(int(r[key_a]), int(r[key_b])): r for r in recs}


def assert_records_match(gpu_recs, orc_recs, variants, n_samples=None, rtol=1e-6, p_floor=1e-290, exact_counts=True):
    """gpu_recs: tomahawk_amd.RECORD_DTYPE (variant indices); orc_recs: oracle RECORD_DTYPE (rid/pos).

    Bar (BASELINE.json north_star): counts bit-exact, statistics within 1e-6 relative.
      * Records produced by PhasedMath (flag bit 0: -p, or -u pairs without double hets) are held
        to exactly that: integer counts identical, every statistic within `rtol`, no floors.
      * Records produced by the unphased cubic (ld_engine.cpp:1363-1558) carry *expected* haplotype
        counts f*2n.  The cubic is ill-conditioned where D ~ 0 and next to a double root (acos near
        +-1): there the reference's own answer is only good to ~1e-8 in haplotype frequency, and the
        last digits depend on libm.  Those records get absolute floors at that scale on top of
        `rtol`; a floor never exceeds 1e-6 of the quantity's natural range.
    """
    pos2idx = {(int(v["rid"]), int(v["pos"])): i for i, v in enumerate(variants)}
    want = {}
    for r in orc_recs:
        want[(pos2idx[(int(r["ridA"]), int(r["Apos"]))], pos2idx[(int(r["ridB"]), int(r["Bpos"]))])] = r
    got = records_by_pair(gpu_recs, "idxA", "idxB")
    assert len(got) == len(gpu_recs), "duplicate pairs in GPU output"
    missing = set(want) - set(got)
    extra = set(got) - set(want)
    assert not missing and not extra, f"pair sets differ: missing {sorted(missing)[:5]} extra {sorted(extra)[:5]}"
    ties, bad = [], []
    for k, w in want.items():
        g = got[k]
        phased_math = bool(int(w["controller"]) & 1)
        if (int(g["flags"]) ^ int(w["controller"])) & ~(1 << 5):   # bit 5 (multiple roots) checked below
            bad.append((k, "flags", int(g["flags"]), int(w["controller"])))
            continue
        total = float(np.sum(w["cnt"]))
        if phased_math:
            floors = dict(D=0.0, Dprime=0.0, R=0.0, R2=0.0, ChiSqFisher=0.0, ChiSqModel=0.0)
            if not np.array_equal(g["cnt"], w["cnt"]):          # integer counts, slot for slot (`exact_counts` is historical)
                bad.append((k, "cnt", g["cnt"].tolist(), w["cnt"].tolist()))
        else:
            floors = dict(D=1e-8, Dprime=1e-6, R=1e-6, R2=1e-8, ChiSqFisher=1e-8 * total, ChiSqModel=0.0)
            if not np.allclose(g["cnt"], w["cnt"], rtol=rtol, atol=1e-8 * total):
                bad.append((k, "cnt", g["cnt"].tolist(), w["cnt"].tolist()))
            if (int(g["flags"]) ^ int(w["controller"])) & (1 << 5):
                # root multiplicity may flip when a second root sits on the admissibility boundary
                ties.append((k, "roots"))
        for f, atol in floors.items():
            if not np.isclose(g[f], w[f], rtol=rtol, atol=atol):
                bad.append((k, f, float(g[f]), float(w[f])))
        # Fisher P underflows to exactly 0 for strong associations: absolute floor.
        # UnphasedMath runs Fisher on round(expected counts) (ld_engine.cpp:1656): when an expected
        # count is a half-integer up to rounding error, round() is decided by the last ulp of the
        # cubic root and either neighbouring table is a faithful answer.
        frac = np.abs(np.asarray(w["cnt"]) - np.floor(w["cnt"]) - 0.5)
        if not phased_math and (frac < 1e-6).any():
            ties.append((k, "round"))
        elif not np.isclose(g["P"], w["P"], rtol=rtol, atol=p_floor):
            bad.append((k, "P", float(g["P"]), float(w["P"])))
    assert not bad, f"{len(bad)} field mismatches of {len(want)} records, first: {bad[:8]}"
    assert len(ties) <= max(2, len(want) // 50), f"too many rounding ties: {len(ties)} of {len(want)}: {ties[:5]}"
