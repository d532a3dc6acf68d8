import sys, numpy as np
sys.path.insert(0, '.')
import tomahawk_amd as T
from oracle import oracle as O
from tests import util
hip = T.HipLd(0)
bad = 0; total = 0
for it in range(36):
    N = [64, 128, 192, 7, 320, 1024, 33, 100, 2504][it % 9]
    miss = it % 2 == 1
    al = util.extreme_alleles(70, N, 900 + it, miss)
    data, mask, variants = util.upload(hip, al)
    for mode, ph in ((T.MODE_UNPHASED, False), (T.MODE_PHASED, True), (T.MODE_AUTO, None)):
        st = O.settings(minR2=0.0, phased=bool(ph), unphased=(ph is False))
        want = O.all_pairs(data, mask, variants, N, st, vector_only=False)
        got, _, _ = hip.ld_all(mode, T.Filters(minR2=0.0))
        total += len(want)
        try:
            util.assert_records_match(got, want, variants, double_root=util.double_root_vetter(data, mask, variants, N))
        except AssertionError as e:
            bad += 1
            print("MISMATCH it", it, "N", N, "mode", mode, "miss", miss, str(e)[:500], flush=True)
print("records compared", total, "failing datasets", bad)
