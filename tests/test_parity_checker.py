"""The parity checker itself (tests/util.py assert_records_match) on the CPU: what it excuses must stay bounded."""
import numpy as np
import pytest

import tomahawk_amd as T
from oracle import oracle as O
from tests import util


def _as_device_records(orc, variants):
    """Oracle records (rid/pos) rewritten as device records (variant indices), field for field."""
    pos2idx = {(int(v["rid"]), int(v["pos"])): i for i, v in enumerate(variants)}
    out = np.zeros(len(orc), dtype=T.RECORD_DTYPE)
    for i, r in enumerate(orc):
        out[i]["idxA"] = pos2idx[(int(r["ridA"]), int(r["Apos"]))]
        out[i]["idxB"] = pos2idx[(int(r["ridB"]), int(r["Bpos"]))]
        out[i]["flags"] = r["controller"]
        for f in ("cnt", "D", "Dprime", "R", "R2", "P", "ChiSqFisher", "ChiSqModel"):
            out[i][f] = r[f]
    return out


def test_cubic_floor_is_bounded_even_when_the_root_error_model_says_infinity():
    """root_error() returns inf where the cubic's slope at the root vanishes (a double root).  The floor such a record gets
    must not follow it there: a device value that is off by 1e-6 in D has to fail however the record is conditioned."""
    N, M = 200, 24
    al = util.random_alleles(M, N, seed=11)
    data, mask = O.bitvectors_from_alleles(al)
    variants = O.variants_from_alleles(al)
    want = O.all_pairs(data, mask, variants, N, O.settings(minR2=0.0, unphased=True), vector_only=True)
    cubic = [i for i, r in enumerate(want) if not (int(r["controller"]) & 1)]
    assert cubic, "the data set must reach the unphased cubic"
    got = _as_device_records(want, variants)
    util.assert_records_match(got, want, variants, count=False)                 # identical records pass

    def hopeless(*_):            # a vetter that excuses nothing but claims every root is infinitely ill-conditioned
        return False
    hopeless.root_error = lambda A, B, f11: (float("inf"), 1.0, 0.0)
    bad = got.copy()
    bad[cubic[0]]["D"] += 1e-6
    with pytest.raises(AssertionError):
        util.assert_records_match(bad, want, variants, double_root=hopeless, count=False)
    # ... while a deviation inside the ceiling is what the floor is for
    ok = got.copy()
    ok[cubic[0]]["D"] += 0.5 * util.DX_CEILING
    util.assert_records_match(ok, want, variants, double_root=hopeless, count=False)
    assert util.ROOT_ERROR_FACTOR * 8.6e-12 < util.DX_CEILING < 5e-11 * util.ROOT_ERROR_FACTOR * 2


def test_ledger_records_how_much_of_a_floor_was_used_not_how_much_was_granted():
    """parity_exemptions.json's largest_floor_used_fraction: (|got - want| - rtol |want|) / floor of the record that came closest to its
    floor - the need, where largest_floor reports the allowance (and sits at the ceiling whenever one badly conditioned record was seen).
    A record that uses half of its floor or more fails the session."""
    import collections
    N, M = 200, 24
    al = util.random_alleles(M, N, seed=11)
    data, mask = O.bitvectors_from_alleles(al)
    variants = O.variants_from_alleles(al)
    want = O.all_pairs(data, mask, variants, N, O.settings(minR2=0.0, unphased=True), vector_only=True)
    cubic = [i for i, r in enumerate(want) if not (int(r["controller"]) & 1) and abs(float(r["D"])) < 1e-4]
    assert cubic
    got = _as_device_records(want, variants)

    def vet(*_):
        return False
    vet.root_error = lambda A, B, f11: (1e-11, 1.0, 1.0)              # floor on D: ROOT_ERROR_FACTOR x 1e-11 = 4e-11
    keep = (dict(util.FLOOR_USED), dict(util.LARGEST_FLOOR), collections.Counter(util.EXEMPTIONS), collections.Counter(util.COMPARED))
    try:
        util.FLOOR_USED.clear()
        off = got.copy()
        off[cubic[0]]["D"] += 1e-11                                    # a quarter of the floor (the relative bar covers < 1e-10 of a D this small)
        util.assert_records_match(off, want, variants, double_root=vet)
        used = util.FLOOR_USED["D"]
        assert 0.2 < used < 0.26, used
        summary, bad = util.exemption_summary()
        assert summary["largest_floor_used_fraction"] >= used and not [b for b in bad if "used" in b]
        off[cubic[0]]["D"] += 1.5e-11                                  # now 2.5e-11 of 4e-11: passes the comparison, fails the session
        util.assert_records_match(off, want, variants, double_root=vet)
        assert util.FLOOR_USED["D"] > util.FLOOR_USE_CAP
        assert [b for b in util.exemption_summary()[1] if "used" in b]
    finally:
        util.FLOOR_USED.clear(); util.FLOOR_USED.update(keep[0]); util.LARGEST_FLOOR.update(keep[1])
        util.EXEMPTIONS.clear(); util.EXEMPTIONS.update(keep[2]); util.COMPARED.clear(); util.COMPARED.update(keep[3])
