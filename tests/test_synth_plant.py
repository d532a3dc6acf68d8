"""The synthetic benchmark input with LD planted in it (twk_hip_plant, include/twk_hip.h): host twin on the CPU, the device
generator against it on the GPU.  iid genotypes hold no pair near the default r2 cut-off; a plant turns every odd variant
2k + 1, k < n_planted, into a noisy copy of an even one - exactly n_planted pairs in LD, whose partners and flip
probabilities any rank (and any test) can derive from the seed alone."""
import numpy as np
import pytest

import tomahawk_amd as T
from oracle import oracle as O


def _popcount(words):
    return int(np.unpackbits(np.ascontiguousarray(words).view(np.uint8)).sum())


def test_plant_is_a_bijection_of_copies_onto_sources_and_leaves_the_rest_alone():
    N, M, seed = 3000, 1000, 42
    pl = T.Plant.spread(M, max_eps=0.4)
    assert pl.n_planted == pl.half == M // 2
    srcs = {}
    for v in range(M):
        ps = T.plant_source(seed, pl, v)
        if v % 2 == 0:
            assert ps is None
        else:
            assert ps is not None and ps[0] % 2 == 0 and ps[0] < M and 0.0 <= ps[1] < 0.4
            srcs[v] = ps
    assert len({s for s, _ in srcs.values()}) == M // 2                 # no two copies share a source
    eps = np.array([e for _, e in srcs.values()])
    assert eps.min() < 0.02 and eps.max() > 0.38 and 0.17 < eps.mean() < 0.23
    dist = np.array([abs(v - s) for v, (s, _) in srcs.items()])
    assert dist.max() > M // 2 and np.median(dist) > M // 8             # spread over the whole triangle
    for v in (0, 10, 998):                                              # sources are the plain generator's rows
        a, ac_a = T.synth_bitvector(seed, N, v)
        b, ac_b = T.synth_bitvector(seed, N, v, pl)
        assert np.array_equal(a, b) and ac_a == ac_b == _popcount(a)
    for v in (1, 77, 999):                                              # a copy = its source with about eps of the alleles flipped
        s, e = srcs[v]
        d, ac = T.synth_bitvector(seed, N, v, pl)
        assert ac == _popcount(d)
        flips = _popcount(d ^ T.synth_bitvector(seed, N, s, pl)[0])
        assert abs(flips / (2 * N) - e) < 4 * (e * (1 - e) / (2 * N)) ** 0.5 + 1e-9, (v, e, flips)
    # fewer copies than sources; a fixed distance (window runs); no plant at all
    few = T.Plant.spread(M, n_planted=10)
    assert T.plant_source(seed, few, 19) is not None and T.plant_source(seed, few, 21) is None
    near = T.Plant.near(M, 7)
    assert T.plant_source(seed, near, 101)[0] == 108
    assert np.array_equal(T.synth_bitvector(seed, N, 21, few)[0], T.synth_bitvector(seed, N, 21)[0])


def test_planted_pairs_span_the_default_cutoff_under_the_oracle():
    """r2 of a copy against its source, by the oracle's UnphasedMath: from ~1 (eps ~ 0) to below 0.1 (eps ~ 0.4), falling with eps."""
    N, M, seed = 20_000, 600, 7
    pl = T.Plant.spread(M, max_eps=0.4)
    st = O.settings(minR2=0.0, unphased=True)
    got = []
    for v in range(1, M, 2):
        s, e = T.plant_source(seed, pl, v)
        d, ac_d = T.synth_bitvector(seed, N, v, pl)
        a, ac_a = T.synth_bitvector(seed, N, s, pl)
        variants = np.zeros(2, dtype=O.VARIANT_DTYPE)
        variants["ac"] = [ac_a, ac_d]; variants["pos"] = [1000, 1100]; variants["hwe"] = 1.0; variants["gt_phase"] = 1
        r = O.all_pairs(np.stack([a, d]), None, variants, N, st, vector_only=False)
        got.append((e, float(r["R2"][0])))
    got.sort()
    eps, r2 = np.array(got).T
    assert r2[eps < 0.01].min() > 0.9 and r2[eps > 0.38].max() < 0.1
    assert (r2 >= 0.1).sum() > 0.6 * len(r2) and (r2 < 0.1).sum() > 0.05 * len(r2)
    assert np.corrcoef(eps, r2)[0, 1] < -0.9


def test_bad_plants_are_refused_without_a_device():
    lib = T.load_library()
    import ctypes as C
    out = np.zeros(T.hip.words64(64), dtype=np.uint64)
    for bad in (T.Plant(5, 4, 1, 0, 0.1), T.Plant(4, 4, 0, 0, 0.1), T.Plant(4, 4, 1, 0, 0.6), T.Plant(4, 4, 1, 0, -0.1)):
        assert lib.twk_synth_plant_source(1, C.byref(bad), 1, None, None) == 0
        assert lib.twk_synth_planted_bitvector(1, 64, 1, C.byref(bad), out.ctypes.data) == 0 and not out.any()


@pytest.mark.gpu
@pytest.mark.parametrize("N", [64, 1000, 70_001])
def test_device_generator_equals_the_host_twin(hip, N):
    M, seed = 300, 99
    pl = T.Plant.spread(M, max_eps=0.35)
    hip.set_problem(N, M)
    hip.generate_synthetic(seed, plant=pl)
    data, mask = hip.download()
    assert not mask.any()
    ac_dev = hip.marginals()[0]
    for v in (0, 1, 2, 57, 58, 299):
        row, ac = T.synth_bitvector(seed, N, v, pl)
        assert np.array_equal(data[v], row) and ac == ac_dev[v], v
    # a slab of the same data set (first_variant: global ids) agrees with the whole on the variants they share
    hip.set_problem(N, 100)
    hip.generate_synthetic(seed, first_variant=150, plant=pl)
    slab, _ = hip.download()
    assert np.array_equal(slab, data[150:250])
    # and a bad plant is refused
    with pytest.raises(T.HipError):
        hip.generate_synthetic(seed, plant=T.Plant(200, 150, 1, 0, 0.1))
