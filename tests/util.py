"""Shared helpers for the parity tests."""
import numpy as np

from oracle import oracle as O
import tomahawk_amd as T
from tomahawk_amd.hip import META_DTYPE

import collections as _collections
import os as _os
_STATS_PATH = _os.environ.get("TWK_PARITY_STATS", "")     # tests/sweeps: record the observed deviations

# ---- bookkeeping of every exemption assert_records_match grants (reported at session end, tests/conftest.py) -------
# kinds: "floor:<field>"   a cubic-path record passed <field> only through its own absolute floor (cubic_floors), not the 1e-6 bar
#        "p-denormal"      Fisher's P below DBL_MIN on both sides, equal to a few hundred steps of the denormal grid
#        "p-floor"         Fisher's P compared through an absolute floor of 1e-320 (the denormal grid) otherwise; 1e-290 until the walks started on the reference's re-synchronisation cells
#        "tie:roots"       root-multiplicity flag (bit 5) differs
#        "tie:round"       round()ed expected counts differ by one: the device's P is Fisher's P of its own table
#        "tie:fisher-stop" P differs by exactly the observed table's own probability (n >= 1e6, or P on the denormal grid)
#        "double-root"     pair reported by one side only, proved to sit on a double root of the cubic
EXEMPTIONS = _collections.Counter()
COMPARED = _collections.Counter()          # "records", "cubic" (records out of the unphased cubic), "calls"
# Caps on the session totals, as a fraction of the cubic-path records compared (floors, roots, round, double-root) or of
# all records compared (p-floor, fisher-stop): about twice the rates of the first full accounting run
# (profiles/r03_parity_exemptions.json).  A drift beyond them fails the session.
# First accounting run (round 3, 171 -m gpu tests): 1,245,531 records, 420,616 of them from the cubic: floor:D / Dprime /
# R / R2 / ChiSqFisher 54 each (1.3e-4), floor:cnt 1,027 (2.4e-3: expected counts next to zero, where a relative bar means
# nothing), tie:round 4 (1e-5), tie:roots 0, double-root 2 (5e-6), tie:fisher-stop 26 (2.1e-5 of all records), p-floor 0.
# (Fisher's test evaluated term by term from the log-factorial table - tried in round 3 - would add ~3e-4 of all records
# as ties where the reference's recurrence starts on denormal terms, P below ~1e-280, and a few hundred on the denormal
# grid itself; the device runs the reference's recurrence from the cells where the reference re-synchronises it, and
# agrees with it down to P = 0.)
EXEMPTION_CAPS = {"floor:D": 3e-4, "floor:Dprime": 3e-4, "floor:R": 3e-4, "floor:R2": 3e-4, "floor:ChiSqFisher": 3e-4,
                  "floor:cnt": 5e-3, "tie:roots": 1e-5, "tie:round": 3e-5, "tie:fisher-stop": 6e-5, "double-root": 1.5e-5,
                  "p-floor": 1e-5, "p-denormal": 2e-3}


def exemption_summary():
    """Session totals + whether every kind stays under its cap -> (dict, list of violations)."""
    out = {"compared": dict(COMPARED), "exemptions": dict(EXEMPTIONS), "rates": {}, "caps": dict(EXEMPTION_CAPS),
           "largest_floor": {"dx_f11_scale": LARGEST_FLOOR["dx"], "ceiling": DX_CEILING},
           "largest_floor_used_fraction": max(FLOOR_USED.values(), default=0.0), "floor_used_fraction_by_field": dict(FLOOR_USED),
           "floor_use_cap": FLOOR_USE_CAP}
    bad = []
    if LARGEST_FLOOR["dx"] > DX_CEILING:
        bad.append(f"largest cubic floor granted {LARGEST_FLOOR['dx']:.3g} > ceiling {DX_CEILING:g}")
    for field, used in FLOOR_USED.items():
        if used >= FLOOR_USE_CAP:
            bad.append(f"a record used {used:.3g} of the floor its conditioning grants on {field} (cap {FLOOR_USE_CAP:g})")
    for kind, cap in EXEMPTION_CAPS.items():
        denom = COMPARED["records"] if kind in ("p-floor", "p-denormal", "tie:fisher-stop") else COMPARED["cubic"]
        rate = EXEMPTIONS[kind] / denom if denom else 0.0
        out["rates"][kind] = rate
        # (sessions that compare a handful of records cannot be held to a rate: allow two of a kind)
        if EXEMPTIONS[kind] > max(2, cap * denom):
            bad.append(f"{kind}: {EXEMPTIONS[kind]} of {denom} ({rate:.3g} > cap {cap:g})")
    return out, bad


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


def extreme_alleles(M, N, seed, miss=False):
    """Hostile genotypes: allele frequencies at the extremes (0.05 % .. 2 % and 98 % .. 99.95 %), an all-het
    variant, complementary and identical variants, and (miss) variants with 30 % / 70 % / 95 % missing samples.
    Every site stays polymorphic with >= 4 called alleles (the reference asserts on monomorphic sites)."""
    rng = np.random.default_rng(seed)
    p = np.concatenate([rng.uniform(0.0005, 0.02, M // 3), rng.uniform(0.98, 0.9995, M // 3), rng.uniform(0.3, 0.7, M - 2 * (M // 3))])
    rng.shuffle(p)
    al = (rng.random((M, N, 2)) < p[:, None, None]).astype(np.int8)
    al[1, :, 0] = 0; al[1, :, 1] = 1
    al[2] = 1 - al[3]
    al[4] = al[5]
    if miss:
        for v in range(0, M, 3):
            ms = rng.random(N) < rng.choice([0.3, 0.7, 0.95])
            al[v, ms, :] = 2
    for v in range(M):
        nz = al[v][al[v] != 2]
        if (nz == 1).sum() == 0 or (nz == 0).sum() == 0 or len(nz) < 4:
            al[v, :2, :] = [[0, 1], [1, 0]]
    return al


def to_hip_meta(variants):
    m = np.zeros(len(variants), dtype=META_DTYPE)
    for k in ("ac", "an", "pos", "rid", "hwe"):
        m[k] = variants[k]
    m["missing"] = variants["gt_missing"]
    return m


_LAST_UPLOAD = {}          # what upload() last sent to the engine: assert_records_match takes the cubic's conditioning from it


def remember_data(alleles, variants):
    """For comparisons of records that did not come through upload() (a CLI run on a .twk written from `alleles`): the data
    set the cubic's conditioning is to be taken from."""
    data, mask = O.bitvectors_from_alleles(alleles)
    _LAST_UPLOAD.update(data=data, mask=mask, variants=variants, n_samples=alleles.shape[1], vet=None)


def upload(hip, alleles, variants=None, **kw):
    M, N, _ = alleles.shape
    data, mask = O.bitvectors_from_alleles(alleles)
    if variants is None:
        variants = O.variants_from_alleles(alleles, **kw)
    hip.set_problem(N, M)
    hip.upload(data, to_hip_meta(variants), mask)
    _LAST_UPLOAD.update(data=data, mask=mask, variants=variants, n_samples=N, vet=None)
    return data, mask, variants


def records_by_pair(recs, key_a, key_b):
    return {(int(r[key_a]), int(r[key_b])): r for r in recs}


def double_root_vetter(data, mask, variants, n_samples):
    """-> f(A, B): True when the unphased cubic of pair (A, B) sits on a double root, i.e. yN^2 and h2 agree to
    a few ulps.  There the reference's branch (three real roots / one) is decided by the last bit of its
    libm's pow(d2, 3.0) - glibc's is not always correctly rounded - and whether the pair is reported at all
    depends on it (ld_engine.cpp:1429-1558).  Such pairs may legitimately differ; nothing else may."""
    def vet(A, B):
        mA = mask[A] if mask is not None and variants["gt_missing"][A] else None
        mB = mask[B] if mask is not None and variants["gt_missing"][B] else None
        c = [float(x) for x in O.count_unphased(data[A], mA, data[B], mB, n_samples)]
        a0, a14, a5, a1664, hets, a2169, a80, a8184, a85 = c
        total = sum(c)
        if total == 0 or hets == 0:
            return False
        P = ((a0 + a14 + a5) * 2.0 + (a1664 + hets + a2169)) / (2.0 * total)
        Q = ((a0 + a1664 + a80) * 2.0 + (a14 + hets + a8184)) / (2.0 * total)
        n11 = 2 * a0 + a14 + a1664
        dee = -n11 * P * Q
        cc = -n11 * (1 - 2 * P - 2 * Q) - hets * (1 - P - Q) + 2 * total * P * Q
        b = 2 * total * (1 - 2 * P - 2 * Q) - 2 * n11 - hets
        a = 4 * total
        xN = -b / (3 * a)
        d2 = (b * b - 3 * a * cc) / (9 * a * a)
        yN = a * xN ** 3 + b * xN * xN + cc * xN + dee
        yN2, h2 = yN * yN, 4 * a * a * d2 ** 3
        return abs(yN2 - h2) <= 16 * np.spacing(max(yN2, h2))

    def root_error(A, B, f11):
        """How far the root the record carries can move under rounding: the larger of (1) the conditioning of the cubic at
        that root - one unit of rounding in every term, |dx| = u * (|a x^3| + |b x^2| + |c x| + |d|) / |g'(x)|, u = 2^-53 - and
        (2) a first-order model of the reference's trigonometric solution where the cubic has three real roots (below).
        -> (dx, the terms' sum, g'(x)); dx = inf where g' vanishes."""
        mA = mask[A] if mask is not None and variants["gt_missing"][A] else None
        mB = mask[B] if mask is not None and variants["gt_missing"][B] else None
        c = [float(x) for x in O.count_unphased(data[A], mA, data[B], mB, n_samples)]
        a0, a14, a5, a1664, hets, a2169, a80, a8184, a85 = c
        total = sum(c)
        if total == 0:
            return float("inf"), 0.0, 0.0
        P = ((a0 + a14 + a5) * 2.0 + (a1664 + hets + a2169)) / (2.0 * total)
        Q = ((a0 + a1664 + a80) * 2.0 + (a14 + hets + a8184)) / (2.0 * total)
        n11 = 2 * a0 + a14 + a1664
        dee = -n11 * P * Q
        cc = -n11 * (1 - 2 * P - 2 * Q) - hets * (1 - P - Q) + 2 * total * P * Q
        b = 2 * total * (1 - 2 * P - 2 * Q) - 2 * n11 - hets
        a = 4 * total
        x = float(f11)
        u = 2.0 ** -53
        terms = abs(a * x ** 3) + abs(b * x * x) + abs(cc * x) + abs(dee)
        slope = abs(3 * a * x * x + 2 * b * x + cc)
        dx = u * terms / slope if slope > 0 else float("inf")
        # ... and what the reference's own method adds where the cubic has three real roots (ld_engine.cpp:1429-1558: x = xN +
        # 2 delta cos((acos(-yN / h) + 2 pi k) / 3)): next to a double root -yN / h -> +-1, acos has slope 1 / sqrt(1 - arg^2)
        # there, and yN itself is what is left of cancelling O(n) terms - the trigonometric form is not backward stable, its
        # error exceeds u * terms / slope by orders of magnitude (hostile sweep, N = 2,504, complete LD: 5.8e-15 against an
        # observed 1.6e-13; this model gives 6.6e-13)
        xN = -b / (3 * a)
        d2 = (b * b - 3 * a * cc) / (9 * a * a)
        if d2 > 0:
            delta = d2 ** 0.5
            h = 2 * a * delta ** 3
            yN = a * xN ** 3 + b * xN * xN + cc * xN + dee
            if h > 0 and abs(yN) <= h * (1 + 1e-9):
                arg = min(1.0, abs(yN / h))
                e_arg = u * (abs(a * xN ** 3) + abs(b * xN * xN) + abs(cc * xN) + abs(dee)) / h + 4 * u * arg
                gap = 1 - arg * arg
                dtheta = e_arg / gap ** 0.5 if gap > e_arg else (2 * e_arg) ** 0.5
                dx = max(dx, (2 * delta / 3) * dtheta + u * (abs(xN) + 2 * delta))
        return dx, terms, slope
    vet.root_error = root_error
    return vet


# Records that come out of the unphased cubic (ld_engine.cpp:1363-1558) carry the root f11 of a cubic whose coefficients
# are O(n) while its slope at the root is O(1) or, next to a double root, far less: one unit of rounding in every term
# moves the root by dx = 2^-53 (|a x^3| + |b x^2| + |c x| + |d|) / |g'(x)| (double_root_vetter(...).root_error).  Device
# (ocml) and reference (glibc) differ by a fraction of that in f11, and everything else in the record follows from it:
#     D = f11 - pA pB                        |dD|  <= dx
#     D' = D / dmax                          |dD'| <= dx / dmax
#     r = D / sqrt(pA qA pB qB)              |dr|  <= dx / sqrt(pA qA pB qB),  |d r2| <= 2 |r| |dr| + dr^2
#     expected counts = f * total            |dcnt| <= dx * total;   ChiSqFisher = total * r2
# so a record's floors are *its own*: ROOT_ERROR_FACTOR x its dx (largest deviation / dx seen: 0.42 - the pair of
# test_cubic_record_with_cancelling_d: dx 8.6e-12, observed 3.6e-12), propagated through its own dmax and allele frequencies
# (taken from the oracle's record).  A well-conditioned record thereby never gets more than D_FLOOR = 1e-14 on f11's scale
# - what plain rounding leaves of a D that is zero: 50 records of the 432 k cubic records of the -m gpu suite differ
# beyond the relative bar, by at most 1e-15 in D (profiles/r04_parity_stats_beyond.json) - and the one absolute floor of
# round 3 that a record could hide behind (D' 1e-6, sized by a single N = 20,000 pair) is gone.  Callers that hold the
# genotypes pass double_root=double_root_vetter(...); without it only D_FLOOR applies.
D_FLOOR = 1e-14
ROOT_ERROR_FACTOR = 4.0
# ... and never more than DX_CEILING, however badly conditioned the cubic: root_error() grows without bound as the slope at
# the root goes to zero (a double root: dx = inf), and an unbounded floor would accept any device value exactly where the
# trigonometric solver is most likely to be wrong.  2e-10 on f11's scale is forty times the largest deviation of any cubic
# record seen so far (4.3e-12, profiles/r04_parity_sweeps.txt) and below the fixed floor of round 3 (D 5e-11 x 4); a
# pair that needs more is a double-root pair and has to go through the explicit vetter (counted, capped).  The largest floor
# a session actually granted is written to parity_exemptions.json ("largest_floor") and held to the same ceiling there.
DX_CEILING = 2e-10
LARGEST_FLOOR = {"dx": 0.0}          # largest dx (f11 scale) behind a floor that a comparison actually needed
# ... and how much of its floor a record that needed one actually USED: (|got - want| - rtol |want|) / floor, per field, the largest over
# the session.  The figure above reports the allowance (and sits at the ceiling whenever one badly conditioned record was compared); this one
# reports the need.  Held below FLOOR_USE_CAP: a record that uses half of what its conditioning grants is a reason to look.
FLOOR_USED = {}                      # field -> largest used fraction
FLOOR_USE_CAP = 0.5


def cubic_floors(cnt, r, dx):
    """Absolute floors of one cubic-path record from the uncertainty dx of its root -> dict for D, Dprime, R, R2, ChiSqFisher, cnt."""
    tot = float(cnt[0] + cnt[1] + cnt[2] + cnt[3])
    pA, pB = float(cnt[0] + cnt[1]) / tot, float(cnt[0] + cnt[2]) / tot
    dmax = min(pA * pB, pA * (1 - pB), (1 - pA) * pB, (1 - pA) * (1 - pB))         # the smaller of the two dmax the sign of D selects between
    den = (pA * (1 - pA) * pB * (1 - pB)) ** 0.5
    inf = float("inf")
    fR = dx / den if den > 0 else inf
    fR2 = 2 * abs(float(r)) * fR + fR * fR
    return dict(D=dx, Dprime=dx / dmax if dmax > 0 else inf, R=fR, R2=fR2, ChiSqFisher=tot * fR2, cnt=dx * tot)


def _one_term_apart(p_a, p_b, table):
    """|p_a - p_b| == hypergeometric probability of `table` = (n11, n21-slot, n12-slot, n22), to 1e-4."""
    from math import lgamma, exp
    n11, n21, n12, n22 = table
    n1_, n_1, n = n11 + n12, n11 + n21, n11 + n12 + n21 + n22
    lb = lambda a, b: 0.0 if b == 0 or a == b else lgamma(a + 1) - lgamma(b + 1) - lgamma(a - b + 1)
    q = exp(lb(n1_, n11) + lb(n - n1_, n_1 - n11) - lb(n, n_1))
    return abs(abs(p_a - p_b) - q) <= 1e-4 * q


def assert_records_match(gpu_recs, orc_recs, variants, n_samples=None, rtol=1e-6, p_floor=1e-320,
                         double_root=None, count=True):
    """gpu_recs: tomahawk_amd.RECORD_DTYPE (variant indices); orc_recs: oracle RECORD_DTYPE (rid/pos).

    Bar (BASELINE.json north_star): counts bit-exact, statistics within 1e-6 relative.
      * Records produced by PhasedMath (flag bit 0: -p, or -u pairs without double hets) are held
        to exactly that: integer counts identical, every statistic within `rtol`, no floors.
      * Records produced by the unphased cubic (ld_engine.cpp:1363-1558) carry *expected* haplotype
        counts f*2n.  The last digits of the root depend on libm, by an amount the cubic's conditioning at that
        root bounds: those records get, on top of `rtol`, absolute floors of their own (cubic_floors: a few times the
        root's uncertainty, propagated through the record's own dmax and allele frequencies).
    """
    pos2idx = {(int(v["rid"]), int(v["pos"])): i for i, v in enumerate(variants)}
    want = {}
    for r in orc_recs:
        want[(pos2idx[(int(r["ridA"]), int(r["Apos"]))], pos2idx[(int(r["ridB"]), int(r["Bpos"]))])] = r
    got = records_by_pair(gpu_recs, "idxA", "idxB")
    assert len(got) == len(gpu_recs), "duplicate pairs in GPU output"
    missing = set(want) - set(got)
    extra = set(got) - set(want)
    n_double_root = 0
    used = _collections.Counter()               # exemptions this call grants
    if double_root is not None and (missing or extra):
        # pairs on a double root of the cubic may be reported by one side only (see double_root_vetter)
        vetted = {k for k in missing | extra if double_root(*k)}
        assert len(vetted) <= max(1, len(want) // 2000), f"too many double-root differences: {sorted(vetted)[:8]}"
        missing -= vetted; extra -= vetted
        n_double_root = len(vetted)
        for k in vetted:
            want.pop(k, None); got.pop(k, None)
    assert not missing and not extra, f"pair sets differ: missing {sorted(missing)[:5]} extra {sorted(extra)[:5]}"
    ties, bad = [], []
    # the conditioning of each record's cubic: from the caller's vetter, else - when the records are those of the data set
    # upload() last sent - from that data set (only the conditioning: pairs on a double root are excused by an explicit vetter only)
    root_error = getattr(double_root, "root_error", None)
    if root_error is None and _LAST_UPLOAD.get("variants") is variants:
        if _LAST_UPLOAD["vet"] is None:
            _LAST_UPLOAD["vet"] = double_root_vetter(_LAST_UPLOAD["data"], _LAST_UPLOAD["mask"], variants, _LAST_UPLOAD["n_samples"])
        root_error = _LAST_UPLOAD["vet"].root_error
    cnt_floor = 0.0
    dev = {}                                     # largest deviations seen on cubic-path records (TWK_PARITY_STATS)
    for k, w in want.items():
        g = got[k]
        phased_math = bool(int(w["controller"]) & 1)
        if not phased_math and _STATS_PATH:
            tot = float(np.sum(w["cnt"]))
            for f in ("D", "Dprime", "R", "R2"):
                dev[f] = max(dev.get(f, 0.0), abs(float(g[f]) - float(w[f])))
                # what an absolute floor has to cover: the part of the deviation the relative bar does not
                dev[f + "_beyond_rtol"] = max(dev.get(f + "_beyond_rtol", 0.0), abs(float(g[f]) - float(w[f])) - rtol * abs(float(w[f])))
            dev["cnt/total"] = max(dev.get("cnt/total", 0.0), float(np.max(np.abs(g["cnt"] - w["cnt"]))) / tot)
            dev["ChiSqFisher/total"] = max(dev.get("ChiSqFisher/total", 0.0), abs(float(g["ChiSqFisher"]) - float(w["ChiSqFisher"])) / tot)
            dev["n"] = dev.get("n", 0) + 1
            dD = abs(float(g["D"]) - float(w["D"]))
            if dD > rtol * abs(float(w["D"])) or abs(float(g["Dprime"]) - float(w["Dprime"])) > rtol * abs(float(w["Dprime"])) or abs(float(g["R"]) - float(w["R"])) > rtol * abs(float(w["R"])):
                item = {"k": list(k), "dD": dD, "dDp": abs(float(g["Dprime"]) - float(w["Dprime"])), "dR": abs(float(g["R"]) - float(w["R"])),
                        "D": float(w["D"]), "Dp": float(w["Dprime"]), "R": float(w["R"]), "tot": tot, "cnt": [float(x) for x in w["cnt"]]}
                re_fn = getattr(double_root, "root_error", None)
                if re_fn is not None:
                    item["root_error"] = list(re_fn(k[0], k[1], float(w["cnt"][0]) / tot))
                dev.setdefault("beyond", []).append(item)
        if (int(g["flags"]) ^ int(w["controller"])) & ~(1 << 5):   # bit 5 (multiple roots) checked below
            bad.append((k, "flags", int(g["flags"]), int(w["controller"])))
            continue
        total = float(np.sum(w["cnt"]))
        if phased_math:
            floors = dict(D=0.0, Dprime=0.0, R=0.0, R2=0.0, ChiSqFisher=0.0, ChiSqModel=0.0)
            if not np.array_equal(g["cnt"], w["cnt"]):          # integer counts, slot for slot
                bad.append((k, "cnt", g["cnt"].tolist(), w["cnt"].tolist()))
        else:
            dx = D_FLOOR
            if root_error is not None and total > 0:
                dx = min(max(dx, ROOT_ERROR_FACTOR * root_error(k[0], k[1], float(w["cnt"][0]) / total)[0]), DX_CEILING)
            cf = cubic_floors([float(x) for x in w["cnt"]], w["R"], dx)
            cnt_floor = cf["cnt"]
            floors = dict(D=cf["D"], Dprime=cf["Dprime"], R=cf["R"], R2=cf["R2"], ChiSqFisher=cf["ChiSqFisher"], ChiSqModel=0.0)
            if not np.allclose(g["cnt"], w["cnt"], rtol=rtol, atol=cnt_floor):          # each cell within the relative bar or the record's floor
                bad.append((k, "cnt", g["cnt"].tolist(), w["cnt"].tolist(), cnt_floor))
            elif not np.allclose(g["cnt"], w["cnt"], rtol=rtol, atol=0.0):
                used["floor:cnt"] += 1
                if count and cnt_floor > 0 and np.isfinite(cnt_floor):
                    need = float(np.max(np.abs(g["cnt"] - w["cnt"]) - rtol * np.abs(w["cnt"])))
                    FLOOR_USED["cnt"] = max(FLOOR_USED.get("cnt", 0.0), need / cnt_floor)
            if (int(g["flags"]) ^ int(w["controller"])) & (1 << 5):
                # root multiplicity may flip when a second root sits on the admissibility boundary
                ties.append((k, "roots"))
        for f, atol in floors.items():
            if not np.isclose(g[f], w[f], rtol=rtol, atol=atol):
                bad.append((k, f, float(g[f]), float(w[f])))
            elif atol and not np.isclose(g[f], w[f], rtol=rtol, atol=0.0):
                used["floor:" + f] += 1
                if count and f == "D":
                    LARGEST_FLOOR["dx"] = max(LARGEST_FLOOR["dx"], float(atol))
                if count and np.isfinite(atol):
                    need = abs(float(g[f]) - float(w[f])) - rtol * abs(float(w[f]))
                    FLOOR_USED[f] = max(FLOOR_USED.get(f, 0.0), need / float(atol))
        # Fisher P underflows to exactly 0 for strong associations: absolute floor.
        # UnphasedMath runs Fisher on round(expected counts) (ld_engine.cpp:1656).  The expected counts
        # are only as good as the cubic root (above), so when one of them lies within that error of a
        # half-integer the two sides may round to neighbouring tables - at N = 1e7 an error of 1e-8 in
        # haplotype frequency is 0.2 counts.  Such a record is not skipped: the device's P must then be
        # Fisher's P of the device's *own* rounded table (computed by the oracle), and that table must
        # be the oracle's up to one count per cell.
        if not np.isclose(g["P"], w["P"], rtol=rtol, atol=0.0):
            gP, wP = float(g["P"]), float(w["P"])
            gt = [int(np.floor(float(x) + 0.5)) for x in g["cnt"]]       # C round(): halves away from zero
            wt = [int(np.floor(float(x) + 0.5)) for x in w["cnt"]]
            neighbour = (not phased_math) and gt != wt and max(abs(a - b) for a, b in zip(gt, wt)) <= 1
            # ... and every cell that rounds differently must sit on a half-integer within the cubic's tolerance
            for a, b, x in zip(gt, wt, w["cnt"]):
                if a != b and abs(float(x) - np.floor(float(x)) - 0.5) > cnt_floor:
                    neighbour = False
            own = O.fisher(gt[0], gt[2], gt[1], gt[3])[2] if neighbour else None
            if max(abs(gP), abs(wP)) < 2.2250738585072014e-308 and abs(gP - wP) <= 2e-321:
                # below DBL_MIN a double has no relative precision left (spacing 4.9e-324): the reference's ratio recurrence and
                # a term-by-term exp() round differently on that grid - a few hundred grid steps is all that can be asked
                used["p-denormal"] += 1
            elif neighbour and np.isclose(g["P"], own, rtol=rtol, atol=p_floor):
                ties.append((k, "round"))
            elif gt == wt and (sum(gt) >= 1_000_000 or max(gP, wP) < 1e-305) and _one_term_apart(gP, wP, gt):
                # kt_fisher_exact stops its tail walks where a term reaches 0.99999999 q (q = the observed
                # table's probability) and adds that term only if it is below 1.00000001 q
                # (fisher_math.cpp:249-258).  On the observed table's own side that term IS q, recomputed
                # through lgamma - whose rounding noise at n ~ 1e7 (~1e-7 relative in q) exceeds the 1e-8
                # band, so whether the observed table's own probability is counted in P is decided by the
                # last bits of libm's lgamma, in the reference itself as on the device (a high-precision
                # evaluation agrees with the reference to 2e-8 when it does count it).  The two P then
                # differ by exactly q: checked here, nothing else is allowed.  The same at any n once q lies on the
                # denormal grid (P < 1e-305): the 1e-8 band is then a fraction of a grid step to a few hundred steps wide,
                # and one step of difference between the host's and the device's exp() decides (first seen when the
                # absolute floor on P went from 1e-290 to 1e-320: N = 100,000, P = 3.2e-316 against 4.7e-316).  (Also for q below
                # ~1e-290 at any n, where the reference's recurrence starts on denormal terms; there the
                # device runs the reference's own recurrence - k_ld_fisher_t - from the same cells and agrees.)
                ties.append((k, "fisher-stop", sum(gt), max(gP, wP)))
            elif np.isclose(gP, wP, rtol=rtol, atol=p_floor):
                # Fisher P underflows to exactly 0 for strong associations: absolute floor for what is left
                used["p-floor"] += 1
            else:
                bad.append((k, "P", gP, wP, own))
    if _STATS_PATH and dev:
        import json, os
        dev.update(test=os.environ.get("PYTEST_CURRENT_TEST", ""), n_samples=n_samples, records=len(want), ties=len(ties))
        os.makedirs(os.path.dirname(_STATS_PATH) or ".", exist_ok=True)
        with open(_STATS_PATH, "a") as fh:
            fh.write(json.dumps(dev) + "\n")
    assert not bad, f"{len(bad)} field mismatches of {len(want)} records, first: {bad[:8]}"
    # every tie kind is argued record by record above AND capped per call, so that a systematic device error (always
    # rounding halves the other way, never counting the observed table in Fisher's sum) cannot hide behind them
    for kind, frac in (("roots", 0.02), ("round", 0.02), ("fisher-stop", 0.15)):
        mine = [t for t in ties if t[1] == kind]
        used["tie:" + kind] += len(mine)
        assert len(mine) <= max(2, int(len(want) * frac)), f"too many {kind} ties: {len(mine)} of {len(want)}: {mine[:5]}"
        if kind == "fisher-stop":       # only where the reference's own stopping rule is undecided: n >= 1e6, or P on the denormal grid
            assert all(t[2] >= 1_000_000 or t[3] < 1e-305 for t in mine), mine[:5]
    used["double-root"] += n_double_root
    n_cubic = sum(1 for w in want.values() if not (int(w["controller"]) & 1))
    if count:                                    # (count=False: a call that is expected to fail, kept out of the session's accounting)
        EXEMPTIONS.update(used)
        COMPARED.update({"records": len(want), "cubic": n_cubic, "calls": 1})
    return {"records": len(want), "cubic": n_cubic, **{k: v for k, v in used.items() if v}}
