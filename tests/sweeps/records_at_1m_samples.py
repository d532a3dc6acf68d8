import sys, numpy as np, time
sys.path.insert(0, '.')
import tomahawk_amd as T
from oracle import oracle as O
from tests import util
hip = T.HipLd(0)
for N, M, seed, miss in ((1_000_000, 20, 1, False), (1_000_003, 16, 2, True), (500_000, 24, 3, False)):
    al = util.mosaic_alleles(M, N, seed, n_founders=5, switch=0.05, mut=0.01, miss_rate=0.01 if miss else 0.0, miss_variants=0.3 if miss else 0.0)
    data, mask, variants = util.upload(hip, al)
    vet = util.double_root_vetter(data, mask, variants, N)
    for mode, ph in ((T.MODE_UNPHASED, False), (T.MODE_PHASED, True), (T.MODE_AUTO, None)):
        st = O.settings(minR2=0.0, phased=bool(ph), unphased=(ph is False))
        t = time.time(); want = O.all_pairs(data, mask, variants, N, st, vector_only=False); to = time.time() - t
        got, _, _ = hip.ld_all(mode, T.Filters(minR2=0.0))
        util.assert_records_match(got, want, variants, double_root=vet)
        print("N", N, "mode", mode, "records", len(want), "match (oracle %.1fs)" % to, flush=True)
