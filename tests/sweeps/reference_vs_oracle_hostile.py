import sys, os, tempfile, numpy as np
sys.path.insert(0, '.')
from oracle import oracle as O
from tests import util
from tomahawk_amd import hostlib
sys.path.insert(0, 'tests/golden')
from make_golden import parse_dump, forward_only
from tests.test_oracle_golden import oracle_matrix

def nasty(M, N, seed, miss):
    rng = np.random.default_rng(seed)
    p = np.concatenate([rng.uniform(0.0005, 0.02, M // 3), rng.uniform(0.98, 0.9995, M // 3), rng.uniform(0.3, 0.7, M - 2 * (M // 3))])
    rng.shuffle(p)
    al = (rng.random((M, N, 2)) < p[:, None, None]).astype(np.int8)
    # some all-het and complementary variants
    al[1, :, 0] = 0; al[1, :, 1] = 1
    al[2] = 1 - al[3]
    al[4] = al[5]
    if miss:
        for v in range(0, M, 3):
            ms = rng.random(N) < rng.choice([0.3, 0.7, 0.95])
            al[v, ms, :] = 2
    for v in range(M):                      # the reference asserts on monomorphic sites
        nz = al[v][al[v] != 2]
        if (nz == 1).sum() == 0 or (nz == 0).sum() == 0 or len(nz) < 4:
            al[v, :2, :] = [[0, 1], [1, 0]]
    return al

tmp = tempfile.mkdtemp()
tot = 0; bad = 0
for it in range(24):
    N = [64, 128, 192, 64, 320, 1024][it % 6]
    miss = it % 2 == 1
    M = 70
    al = nasty(M, N, 900 + it, miss)
    pos = (1000 + 100 * np.arange(M)).astype(np.uint32); rid = np.zeros(M, np.uint32)
    twk = os.path.join(tmp, "n.twk")
    hostlib.write_twk(twk, al, pos, rid, phased=np.ones(M, np.uint8), n_contigs=1, block_size=30)
    data, mask = O.bitvectors_from_alleles(al)
    variants = O.variants_from_alleles(al, pos=pos, rid=rid, phase=1)
    for tag, flag, kw in (("p", ["-p"], dict(phased=True)), ("u", ["-u"], dict(unphased=True)), ("d", [], {})):
        two = os.path.join(tmp, "o.two")
        r = O.run_ref(["calc", "-i", twk, "-o", two, "-r", "0", "-t", "1"] + flag)
        ref = forward_only(parse_dump(O.run_ref(["dump", two]).stdout))
        got = oracle_matrix(O.all_pairs(data, mask, variants, N, O.settings(minR2=0.0, **kw), vector_only=False))
        tot += len(ref)
        same = ref.shape == got.shape and np.array_equal(ref, got, equal_nan=True)
        if not same:
            bad += 1
            print("DIFF it", it, "N", N, "miss", miss, tag, ref.shape, got.shape)
            if ref.shape == got.shape:
                d = np.argwhere(~((ref == got) | (np.isnan(ref) & np.isnan(got))))
                print(d[:5], ref[d[0][0]], got[d[0][0]])
print("records", tot, "differing sets", bad)
