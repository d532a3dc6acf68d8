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


# ---- the three-product form (k_count3_list_t + k_screen3_pairs + k_recount_unphased) at the sizes bench.py runs it --------------
def _plain_rows(N, ids, plant):
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(16) as pool:
        return np.stack(list(pool.map(lambda v: T.synth_bitvector(SEED, N, int(v), plant)[0], ids)))


def _oracle_pairs(N, pairs, index, data, variants, st):
    """The oracle's records of the given variant pairs (global ids, each a two-variant problem of its own: the scalar oracle at
    N = 1 M takes milliseconds a pair) -> oracle records that carry the variants' own positions."""
    out = []
    for a, b in pairs:
        ia, ib = index[a], index[b]
        r = O.all_pairs(np.stack([data[ia], data[ib]]), None, variants[[ia, ib]], N, st, vector_only=False)
        out.append(r)
    return np.concatenate(out) if out else np.zeros(0, dtype=O.RECORD_DTYPE)


def _localise(recs, index):
    got = recs.copy()
    got["idxA"] = [index[int(x)] for x in recs["idxA"]]; got["idxB"] = [index[int(x)] for x in recs["idxB"]]
    return got


def _bytes_equal(a, b):
    order = ["idxA", "idxB"]
    return np.sort(a, order=order).tobytes() == np.sort(b, order=order).tobytes()


@pytest.mark.parametrize("min_chunks", [8, 1])
def test_three_product_form_at_the_headline_size_against_the_oracle(hip, opt, min_chunks):
    """BENCH's headline (configs[2]: 1,000,000 samples x 50,000 variants, calc -u, r2 >= 0.1) runs the three-product form at a row
    length - 31,250 words, 977 K chunks - where it used to be compared with the oracle only up to N = 50,000.  Here, inside the
    resident full-size problem with LD planted in it (the generator of bench.py's extra.cfg3_planted):
      (a) a triangle over the last 6,400 variants - 5,050 tiles: the first 954 stored whole, the rest cut along K and added with
          atomics - at the default cut-off: three = 2 gives the bytes of three = 0, and exactly the planted pairs the ORACLE keeps
          (pairs from r2 ~ 1 down to below the cut-off: one wrongly screened out would be missing), statistics under the bar;
      (b) far-corner regions with rows overwritten by noisy copies of other rows (twk_hip_upload_bitvectors into the resident
          problem), at the default cut-off and at r2 >= 2e-6 - which one iid pair in twenty passes at this N, so the candidate
          list, the recount and the cubic all run on ordinary pairs - against the oracle on every pair of the regions."""
    N, M = 1_000_000, 50_000
    plant = T.Plant.spread(M, max_eps=0.4)
    hip.set_problem(N, M)
    hip.generate_synthetic(SEED, plant=plant)
    opt.set("count_min_chunks", min_chunks)
    mode = T.MODE_UNPHASED

    def both(call):
        opt.set("three", 0); hip.timing_reset()
        four = call()
        assert hip.timing()["three_launches"] == 0
        opt.set("three", 2); hip.timing_reset()
        three = call()
        tm = hip.timing()
        assert tm["three_launches"] > 0 and tm["fused_launches"] == 0 and tm["three_launches"] == tm["count_launches"], tm
        return four, three, tm

    # (a)
    a0, n = M - 6400, 6400
    f = T.Filters(minR2=0.1)
    (p, np0, _), (q, np1, nr), tm = both(lambda: hip.ld_region(mode, f, a0, n, a0, n, True))
    assert np0 == np1 == n * (n - 1) // 2 and nr == len(q) and _bytes_equal(p, q)
    planted = []
    for v in range(a0 | 1, M, 2):
        s, eps = T.plant_source(SEED, plant, v)
        if s >= a0:
            planted.append((min(s, v), max(s, v), eps))
    assert len(planted) > 300 and tm["recount_candidates"] >= len(q) > 0.5 * len(planted)
    ids = sorted({x for a, b, _ in planted for x in (a, b)})
    index = {v: k for k, v in enumerate(ids)}
    data = _plain_rows(N, ids, plant)
    variants = _variants(N, ids, data)
    want = _oracle_pairs(N, [(a, b) for a, b, _ in planted], index, data, variants, O.settings(minR2=0.1, unphased=True))
    assert 0.5 * len(planted) < len(want) < len(planted)           # the plant spans the cut-off: some copies are too noisy
    assert {(int(x), int(y)) for x, y in zip(q["idxA"], q["idxB"])} <= {(a, b) for a, b, _ in planted}      # nothing but planted pairs survives
    vet = util.double_root_vetter(data, None, variants, N)
    cond = lambda A, B: False
    cond.root_error = vet.root_error
    util.assert_records_match(_localise(q, index), want, variants, n_samples=N, double_root=cond)
    if min_chunks != 8:
        return
    # (b) rows of the far corners overwritten with noisy copies: triangle [M - 96, M) and rectangle [M - 4000, +32) x [M - 96, M)
    ids = np.concatenate([np.arange(M - 4000, M - 4000 + 32), np.arange(M - 96, M)])
    index = {int(v): k for k, v in enumerate(ids)}
    data = _plain_rows(N, ids, plant)
    rng = np.random.default_rng(11)
    copies = [(M - 96 + 2 * i + 1, M - 96 + 2 * i, e) for i, e in enumerate(np.linspace(0.0, 0.46, 24))]           # inside the triangle
    copies += [(M - 4000 + j, M - 40 + j, e) for j, e in enumerate(np.linspace(0.01, 0.44, 8))]                      # across the rectangle
    for dst, src, e in copies:
        flips = np.packbits((rng.random(2 * N) < e).astype(np.uint8), bitorder="little").view(np.uint64)
        data[index[dst]] = data[index[src]] ^ flips
    variants = _variants(N, ids, data)
    meta = util.to_hip_meta(variants)
    for dst, _, _ in copies:
        k = index[dst]
        hip.upload(data[k:k + 1], meta[k:k + 1], None, first=dst)
    back, _ = hip.download(M - 96, 96)
    assert np.array_equal(back, data[32:])
    vet = util.double_root_vetter(data, None, variants, N)
    cond = lambda A, B: False
    cond.root_error = vet.root_error
    keep_pairs = {(a, b) for a in range(32, 128) for b in range(a + 1, 128)} | {(a, b) for a in range(32) for b in range(32, 128)}
    pos = {(int(v["rid"]), int(v["pos"])): k for k, v in enumerate(variants)}
    for minR2 in (0.1, 2e-6):
        f = T.Filters(minR2=minR2)
        call = lambda: (np.concatenate([hip.ld_region(mode, f, M - 96, 96, M - 96, 96, True)[0], hip.ld_region(mode, f, M - 4000, 32, M - 96, 96, False)[0]]),)
        (p,), (q,), tm = both(call)
        assert _bytes_equal(p, q)
        want = O.all_pairs(data, None, variants, N, O.settings(minR2=minR2, unphased=True), vector_only=False)
        want = np.array([w for w in want if (pos[(int(w["ridA"]), int(w["Apos"]))], pos[(int(w["ridB"]), int(w["Bpos"]))]) in keep_pairs], dtype=want.dtype)
        assert len(want) == len(q) >= (20 if minR2 >= 0.1 else 150), (minR2, len(want), len(q))
        assert tm["recount_candidates"] >= len(q)
        util.assert_records_match(_localise(q, index), want, variants, n_samples=N, double_root=cond)
