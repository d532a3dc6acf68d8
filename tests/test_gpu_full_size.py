"""BASELINE.json's configurations at their FULL sizes (the sizes bench.py measures), checked where a check does not need
the whole answer: the synthetic input generated in HBM has a bit-identical host twin (twk_synth_bitvector), so any row of a
50,000-variant x 1,000,000-sample problem can be rebuilt on the host and any pair handed to the scalar oracle.  Tiles are
taken from the far corners of the resident matrix - row offsets beyond 4 GiB, the last (partial) tile, rows against columns
49,000 variants apart - so that the addressing of the full-size problem is what is exercised; plus two size-independent
identities over whole launches: every pair is decided exactly once (pair counts of shards and regions add up), and the records
of a region computed inside the full problem equal those of the same variants uploaded as a small problem of their own."""
import numpy as np
import pytest

import tomahawk_amd as T
from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu
SEED = 42


def _rows(N, ids):
    from concurrent.futures import ThreadPoolExecutor          # (0.1 s a row at N = 1 M; the generator is a C call: no GIL)
    with ThreadPoolExecutor(8) as pool:
        return np.stack(list(pool.map(lambda v: T.synth_bitvector(SEED, N, int(v))[0], ids)))


def _variants(N, ids, data):
    v = np.zeros(len(ids), dtype=O.VARIANT_DTYPE)
    for k, (i, row) in enumerate(zip(ids, data)):
        v["ac"][k] = int(sum(bin(int(x)).count("1") for x in row)) if N <= 4096 else int(np.unpackbits(row.view(np.uint8)).sum())
    v["pos"] = 1000 + 100 * np.asarray(ids, dtype=np.uint32)
    v["hwe"] = 1.0
    v["gt_phase"] = 1
    return v


@pytest.mark.parametrize("name,N,M,mode", [("configs[2]", 1_000_000, 50_000, T.MODE_UNPHASED), ("configs[1]", 100_000, 10_000, T.MODE_PHASED)])
def test_full_size_problem_sampled_against_the_oracle(hip, name, N, M, mode):
    hip.set_problem(N, M)
    hip.generate_synthetic(SEED)
    phased = mode == T.MODE_PHASED
    counter = O.count_phased if phased else O.count_unphased
    rng = np.random.default_rng(7)
    # tiles from the corners of the pair space (a0, nA, b0, nB)
    for a0, nA, b0, nB in ((M - 200, 200, M - 200, 200), (0, 130, M - 140, 140), (M // 2 - 64, 128, M // 2 + 17, 150)):
        cells = hip.count_tile(mode, a0, nA, b0, nB)
        ii, jj = rng.integers(0, nA, 10), rng.integers(0, nB, 10)
        rows_a, rows_b = _rows(N, a0 + ii), _rows(N, b0 + jj)
        for k in range(10):
            assert np.array_equal(cells[ii[k], jj[k]], counter(rows_a[k], None, rows_b[k], None, N)), (name, a0 + ii[k], b0 + jj[k])
    # records of a far-corner region inside the full problem == the same variants as a problem of their own == the oracle
    ids = np.concatenate([np.arange(M - 96, M), np.arange(M - 4000, M - 4000 + 32)])          # 128 variants, two clusters
    ids.sort()
    f = T.Filters(minR2=0.0)
    a, npa, _ = hip.ld_region(mode, f, M - 96, 96, M - 96, 96, True)
    b, npb, _ = hip.ld_region(mode, f, M - 4000, 32, M - 96, 96, False)
    assert npa == 96 * 95 // 2 and npb == 32 * 96
    data = _rows(N, ids)
    variants = _variants(N, ids, data)
    st = O.settings(minR2=0.0, phased=phased, unphased=not phased)
    want = O.all_pairs(data, None, variants, N, st, vector_only=True)
    remap = {int(v): k for k, v in enumerate(ids)}
    got = np.concatenate([a, b])
    got["idxA"] = [remap[int(x)] for x in got["idxA"]]; got["idxB"] = [remap[int(x)] for x in got["idxB"]]
    keep = {(int(x), int(y)) for x, y in zip(got["idxA"], got["idxB"])}
    pos = {(int(v["rid"]), int(v["pos"])): k for k, v in enumerate(variants)}
    want = np.array([w for w in want if (pos[(int(w["ridA"]), int(w["Apos"]))], pos[(int(w["ridB"]), int(w["Bpos"]))]) in keep], dtype=want.dtype)
    assert len(want) == len(got) > 4000
    vet = util.double_root_vetter(data, None, variants, N)
    conditioning = lambda A, B: False                  # (only the cubic's conditioning: no pair is excused as a double root)
    conditioning.root_error = vet.root_error
    util.assert_records_match(got, want, variants, n_samples=N, double_root=conditioning)
    # every pair of the triangle is decided exactly once, however it is cut: shards, and a region's band
    parts = [T.shard_rows(M, k, 8)[2] for k in range(8)]
    assert sum(parts) == M * (M - 1) // 2
