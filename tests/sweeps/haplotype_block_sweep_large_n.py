import sys, numpy as np
sys.path.insert(0, '.')
import tomahawk_amd as T
from oracle import oracle as O
from tests import util
hip = T.HipLd(0)
bad = 0; total = 0
for it, N in enumerate([5000, 20000, 20000, 100000, 100000, 300000]):
    M = 90 if N <= 100000 else 50
    miss = (it % 3 == 2)
    al = util.mosaic_alleles(M, N, 7000 + it, n_founders=4 + it % 4, switch=[0.005, 0.02][it % 2], mut=[0.0, 0.001][it % 2],
                             miss_rate=0.02 if miss else 0.0, miss_variants=0.3 if miss else 0.0)
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
            print("MISMATCH it", it, "N", N, "mode", mode, "miss", miss, str(e)[:400], flush=True)
    print("done N", N, flush=True)
print("records compared", total, "failing datasets", bad)
