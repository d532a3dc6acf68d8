"""The engine's planner (csrc/hip/ld_plan.h through twk_hip_plan_region) on the CPU: which rows a shard owns, which columns every row
reaches, the launches that cover them.  Pure host arithmetic - no GPU, no context.  The reference's counterparts are the square-chunk
partition and the block-pair ticker (lib/ld/ld_balancing.h:23-80, 176-233); what must hold is what they guarantee: every wanted pair
is visited exactly once, whatever the number of shards, the tile edge or the window."""
import numpy as np
import pytest

import tomahawk_amd as T


def _meta(M, pos=None, rid=None):
    m = np.zeros(M, dtype=T.META_DTYPE)
    m["pos"] = np.arange(M, dtype=np.uint32) * 100 + 1000 if pos is None else pos
    m["rid"] = 0 if rid is None else rid
    m["ac"] = 5
    return m


def _coverage(tiles, M):
    """How often the tiles visit each pair (i, j): count matrix [M, M]; a diagonal tile visits col > row only."""
    cov = np.zeros((M, M), dtype=np.int32)
    for t in tiles:
        a0, nA, b0, nB = int(t["rowA0"]), int(t["nA"]), int(t["rowB0"]), int(t["nB"])
        sub = np.ones((nA, nB), dtype=np.int32)
        if t["diag"]:
            assert a0 == b0 and nB >= nA
            sub = np.triu(sub, k=1)
        cov[a0:a0 + nA, b0:b0 + nB] += sub
    return cov


@pytest.mark.parametrize("M,tile,ppv", [(1000, 0, 1), (1000, 128, 2), (777, 256, 1), (3000, 0, 3), (129, 0, 2)])
@pytest.mark.parametrize("n_parts", [1, 3, 8])
def test_triangle_shards_cover_every_pair_exactly_once(M, tile, ppv, n_parts):
    meta = _meta(M)
    total = np.zeros((M, M), dtype=np.int32)
    pairs = 0
    prev_end = 0
    for k in range(n_parts):
        pl = T.plan_region(meta, 5000, 0, M, 0, M, True, part=k, n_parts=n_parts, tile_variants=tile, planes_per_variant=ppv)
        assert pl["row_begin"] == prev_end and pl["row_end"] >= pl["row_begin"]
        assert (pl["row_begin"], pl["row_end"]) == T.shard_rows(M, k, n_parts)[:2]
        prev_end = pl["row_end"]
        cov = _coverage(pl["tiles"], M)
        want = np.zeros((M, M), dtype=np.int32)
        want[pl["row_begin"]:pl["row_end"]] = np.triu(np.ones((M, M), dtype=np.int32), k=1)[pl["row_begin"]:pl["row_end"]]
        assert np.array_equal(cov, want), (k, int(np.abs(cov - want).sum()))
        assert pl["n_pairs"] == int(want.sum()) and pl["n_band_launches"] == 0
        assert all(t["nA"] <= 32768 and t["nB"] <= 32768 for t in pl["tiles"])
        total += cov
        pairs += pl["n_pairs"]
    assert prev_end == M and pairs == M * (M - 1) // 2
    assert np.array_equal(total, np.triu(np.ones((M, M), dtype=np.int32), k=1))


@pytest.mark.parametrize("n_parts", [1, 4])
@pytest.mark.parametrize("fused", [False, True])
def test_rectangle_and_band_launches_cover_their_rows(n_parts, fused):
    """A rectangle (a -c / -C chunk off the diagonal), and the fused form's band launches: rows x all the columns they reach, at most
    band_max_launches of them, at least 2^band_work_log2 tile-chunks each."""
    M = 4000
    meta = _meta(M)
    total = np.zeros((M, M), dtype=np.int32)
    for k in range(n_parts):
        pl = T.plan_region(meta, 2504, 200, 1500, 1800, 2100, False, part=k, n_parts=n_parts, fused=fused, k_chunks=5, band_work_log2=8, band_max_launches=5)
        cov = _coverage(pl["tiles"], M)
        want = np.zeros((M, M), dtype=np.int32)
        want[200 + pl["row_begin"]:200 + pl["row_end"], 1800:3900] = 1
        assert np.array_equal(cov, want)
        assert pl["n_pairs"] == int(want.sum())
        if fused:
            assert 1 <= pl["n_band_launches"] == len(pl["tiles"]) <= 5
            assert all(t["rowB0"] == 1800 and t["nB"] == 2100 for t in pl["tiles"])
        else:
            assert pl["n_band_launches"] == 0
        total += cov
    assert total[200:1700, 1800:3900].min() == 1 == total.max() and total.sum() == 1500 * 2100
    # the triangle as band launches: every launch starts on the diagonal and reaches the last column
    pl = T.plan_region(meta, 2504, 0, M, 0, M, True, fused=True, k_chunks=5, band_work_log2=6)
    assert pl["n_band_launches"] == len(pl["tiles"]) == 8 and all(t["diag"] == 1 and t["rowB0"] + t["nB"] == M for t in pl["tiles"])
    assert np.array_equal(_coverage(pl["tiles"], M), np.triu(np.ones((M, M), dtype=np.int32), k=1))


@pytest.mark.parametrize("n_parts", [1, 3])
@pytest.mark.parametrize("fused", [False, True])
def test_window_reach_matches_brute_force_and_every_in_window_pair_is_visited_once(n_parts, fused):
    rng = np.random.default_rng(3)
    M, w = 2500, 7000
    rid = np.sort(rng.integers(0, 3, size=M)).astype(np.uint32)
    pos = np.zeros(M, dtype=np.uint32)
    for r in range(3):
        k = rid == r
        pos[k] = np.sort(rng.integers(0, 400_000, size=int(k.sum())))
    meta = _meta(M, pos, rid)
    inw = (rid[:, None] == rid[None, :]) & (np.abs(pos[:, None].astype(np.int64) - pos[None, :].astype(np.int64)) <= w)
    inw = np.triu(inw, k=1)
    seen = np.zeros((M, M), dtype=np.int32)
    pairs = 0
    for k in range(n_parts):
        pl = T.plan_region(meta, 2504, 0, M, 0, M, True, part=k, n_parts=n_parts, window=T.OPT_WINDOW, l_window=w, fused=fused, k_chunks=3, band_work_log2=6)
        lo, hi = pl["lo"].astype(np.int64), pl["hi"].astype(np.int64)
        for r in range(pl["row_begin"], pl["row_end"]):          # the reach of a row is exactly its in-window partners behind it
            cols = np.flatnonzero(inw[r])
            if len(cols):
                assert lo[r] <= cols[0] and cols[-1] < hi[r] and hi[r] - lo[r] == len(cols), (r, lo[r], hi[r], cols[0], cols[-1])
            else:
                assert hi[r] == lo[r]
        cov = _coverage(pl["tiles"], M)
        mine = np.zeros_like(inw); mine[pl["row_begin"]:pl["row_end"]] = inw[pl["row_begin"]:pl["row_end"]]
        assert (cov[mine] == 1).all(), int((cov[mine] != 1).sum())          # every in-window pair of the shard once (tiles may hold other pairs too)
        assert pl["n_pairs"] == int(mine.sum())
        if fused:
            assert pl["n_band_launches"] >= 1
        seen += cov * mine
        pairs += pl["n_pairs"]
    assert pairs == int(inw.sum()) and np.array_equal(seen, inw.astype(np.int32))


@pytest.mark.parametrize("screen,minR2", [(1, 0.1), (2, 0.1), (1, 0.5), (2, 0.002)])
def test_r2_band_leaves_out_only_pairs_that_cannot_reach_the_cutoff(screen, minR2):
    """Positions in order of minor allele count; outside [lo, hi) no 2 x 2 table with the two variants' margins reaches the cut-off."""
    rng = np.random.default_rng(screen * 7 + int(minR2 * 1000))
    N, M = 5000, 1800
    T2 = 2 * N
    mac = np.sort(np.concatenate([rng.integers(1, 40, size=M // 2), rng.integers(40, N, size=M - M // 2)]))
    ac = np.where(rng.random(M) < 0.3, T2 - mac, mac).astype(np.uint32)          # some variants have ALT as their major allele
    meta = _meta(M)
    pl = T.plan_region(meta, N, 0, M, 0, M, True, popc=ac, screen=screen, minR2=minR2, planes_per_variant=screen, fused=True, k_chunks=5, band_work_log2=8)
    lo, hi = pl["lo"].astype(np.int64), pl["hi"].astype(np.int64)
    assert (np.diff(hi) >= 0).all() and (lo == np.minimum(np.arange(M) + 1, M)).all()
    a = mac[:, None] / T2; b = mac[None, :] / T2
    a, b = np.minimum(a, b), np.maximum(a, b)
    slack = 1e-5 if screen == 2 else 0.0
    best = (a * (1 - b) + slack) ** 2 / (a * (1 - a) * b * (1 - b))
    j = np.arange(M)[None, :]
    outside = (j > np.arange(M)[:, None]) & (j >= hi[:, None])
    assert (best[outside] < minR2).all()
    assert np.array_equal(_coverage(pl["tiles"], M) > 0, _coverage(pl["tiles"], M) == 1)
    cov = _coverage(pl["tiles"], M)
    inside = (j > np.arange(M)[:, None]) & (j < hi[:, None])
    assert (cov[inside] == 1).all() and pl["n_pairs"] == M * (M - 1) // 2
    # allele-count order: the last band (the commonest variants) comes first
    if pl["n_band_launches"] > 1:
        first_rows = [int(t["rowA0"]) for t in pl["tiles"][:pl["n_band_launches"]]]
        assert first_rows == sorted(first_rows, reverse=True)


def test_planner_rejects_what_the_engine_rejects():
    meta = _meta(100)
    with pytest.raises(T.HipError):
        T.plan_region(meta, 100, 0, 100, 0, 100, True, part=2, n_parts=2)
    with pytest.raises(T.HipError):
        T.plan_region(meta, 100, 0, 100, 10, 90, True)
    with pytest.raises(T.HipError):
        T.plan_region(meta, 100, 0, 101, 0, 101, True)
    with pytest.raises(T.HipError):
        T.plan_region(meta, 100, 0, 100, 0, 100, True, screen=1)          # the r2 band needs the allele counts
