"""`tomahawk calc` on the GPU against the COMPILED REFERENCE run live, side by side, on inputs no fixture holds.

Needs both a GPU and oracle/_ref/tomahawk_ref (built by `make -C oracle ref` in the dev container; the binary
travels to the GPU box with the snapshot, the reference's sources do not).  Where the binary is absent the tests
skip - the committed fixtures (tests/golden/, test_gpu_golden.py, test_gpu_cli_golden.py) cover the same paths.

Every record of both files is compared (pair set, flags, counts, statistics) under the bars of tests/util.py.
"""
import os
import subprocess

import numpy as np
import pytest

import tomahawk_amd as T
from oracle import oracle as O
from tests import util
from tomahawk_amd import hostlib

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/tomahawk_ref is not built")]


def _forward_records(path, index):
    """.two file -> (engine-style records of the forward copies, sorted by pair; n records in the file)."""
    recs, _ = hostlib.read_two(path)
    a = np.array([index[(int(r), int(p))] for r, p in zip(recs["ridA"], recs["packA"] >> 2)], dtype=np.int64)
    b = np.array([index[(int(r), int(p))] for r, p in zip(recs["ridB"], recs["packB"] >> 2)], dtype=np.int64)
    f = a < b
    assert f.sum() * 2 == len(recs), "forward / reverse copies do not pair up"
    o = np.lexsort((b[f], a[f]))
    fr = recs[f][o]
    out = np.zeros(len(fr), dtype=T.RECORD_DTYPE)
    out["idxA"] = a[f][o]; out["idxB"] = b[f][o]; out["flags"] = fr["controller"]; out["cnt"] = fr["cnt"]
    for k in ("D", "Dprime", "R", "R2", "P", "ChiSqFisher", "ChiSqModel"):
        out[k] = fr[k]
    return out, len(recs)


def _as_oracle_records(eng, variants):
    w = np.zeros(len(eng), dtype=O.RECORD_DTYPE)
    w["controller"] = eng["flags"]
    w["ridA"] = variants["rid"][eng["idxA"]]; w["Apos"] = variants["pos"][eng["idxA"]]
    w["ridB"] = variants["rid"][eng["idxB"]]; w["Bpos"] = variants["pos"][eng["idxB"]]
    w["cnt"] = eng["cnt"]
    for k in ("D", "Dprime", "R", "R2", "P", "ChiSqFisher", "ChiSqModel"):
        w[k] = eng[k]
    return w


def _dataset(kind, N, M, seed, n_contigs):
    if kind == "mosaic":
        al = util.mosaic_alleles(M, N, seed, n_founders=6, switch=0.02, mut=0.004, miss_rate=0.02, miss_variants=0.2)
    elif kind == "clean":
        al = util.mosaic_alleles(M, N, seed, n_founders=5, switch=0.03, mut=0.003)
    else:
        al = util.random_alleles(M, N, seed, maf_lo=0.02, maf_hi=0.5, miss_rate=0.05, miss_variants=0.3, low_ac=4)
    rid = (np.arange(M) * n_contigs // M).astype(np.uint32)
    pos = np.zeros(M, np.uint32)
    for c in range(n_contigs):
        m = rid == c
        pos[m] = 500 + 41 * np.arange(int(m.sum()))
    return al, pos, rid


# (kind, N, M, seed, contigs, flags, env for the engine)
COMPAT = {"TWK_REF_COMPAT": "1"}
RUNS = [
    ("clean", 1000, 700, 11, 1, ["-p"], None),
    ("clean", 1000, 700, 11, 1, ["-u", "-r", "0.02"], None),
    ("iid", 333, 450, 12, 2, ["-u", "-r", "0.005"], None),          # (iid genotypes: r2 ~ 1 / N, nothing reaches the default 0.1)
    ("iid", 333, 450, 12, 2, ["-r", "0.005"], None),
    ("iid", 333, 450, 12, 2, ["-u", "-r", "0.01", "-P", "0.01"], None),
    ("iid", 333, 450, 12, 2, ["-c", "3", "-C", "3", "-r", "0.005"], None),              # a diagonal chunk of the -c / -C balancer
    ("iid", 333, 450, 12, 2, ["-I", "2:1000-6000", "-u", "-r", "0.005"], None),
    # -p with missing genotypes at 2N % 128 != 0: the reference's own records carry its tail / padding slips (q6/q7)
    ("iid", 333, 450, 12, 2, ["-p", "-r", "0.005"], COMPAT),
    ("mosaic", 64, 600, 13, 3, ["-p"], None),               # 2N % 128 == 0: no slips, plain comparison
    ("mosaic", 64, 600, 13, 3, [], None),
    ("mosaic", 2504, 900, 14, 2, ["-r", "0.3"], None),
    ("mosaic", 2504, 900, 14, 2, ["-u", "-w", "4000"], COMPAT),   # the reference's window mode as it behaves (q8)
    # BASELINE configs[0] as it stands (1k diploid samples x 1k variants, unphased) and configs[1]'s shape (100k samples, phased)
    ("clean", 1000, 1000, 18, 1, ["-u"], None),
    ("clean", 100_000, 1500, 19, 1, ["-p"], None),
    # the headline's sample count, through the reference itself rather than the oracle
    ("clean", 1_000_000, 120, 15, 1, ["-u", "-r", "0.01"], None),
    ("mosaic", 1_000_000, 100, 16, 1, ["-r", "0.01"], None),
    ("clean", 1_000_000, 120, 15, 1, ["-p", "-r", "0.01"], None),
    # ... and with the three-product form of the unphased contraction forced on (the launch sampler keeps to four products on data
    # this rich in LD): k_count3_list_t + k_screen3_pairs + k_recount_unphased at the headline's row length against the reference itself
    ("clean", 1_000_000, 120, 15, 1, ["-u", "-r", "0.01", "--engine-option", "three=2"], None),
    # configs[4]'s sample count (counts near 2e7: the int narrowing of Fisher's arguments, its stop-band instability)
    ("clean", 10_000_000, 24, 17, 1, ["-u", "-r", "0.001"], None),
]


@pytest.mark.parametrize("kind,N,M,seed,n_contigs,flags,env", RUNS, ids=[f"{r[0]}-N{r[1]}-{'_'.join(r[5]) or 'default'}" for r in RUNS])
def test_cli_equals_the_reference_run_live(tmp_path, kind, N, M, seed, n_contigs, flags, env):
    al, pos, rid = _dataset(kind, N, M, seed, n_contigs)
    twk = str(tmp_path / "in.twk")
    hostlib.write_twk(twk, al, pos, rid, phased=np.ones(M, np.uint8), n_contigs=n_contigs, block_size=100)
    variants = O.variants_from_alleles(al, pos=pos, rid=rid, phase=1)
    index = {(int(r), int(p)): i for i, (r, p) in enumerate(zip(rid, pos))}
    ref_two, my_two = str(tmp_path / "ref.two"), str(tmp_path / "mine.two")
    ref_flags = [x for i, x in enumerate(flags) if x != "--engine-option" and (i == 0 or flags[i - 1] != "--engine-option")]      # (the engine's own switch)
    O.run_ref(["calc", "-i", twk, "-o", ref_two, "-t", "4"] + ref_flags, stdin=subprocess.DEVNULL)
    r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", twk, "-o", my_two] + flags, capture_output=True, text=True,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr
    if "three=2" in flags:
        assert "launches in the three-product form" in r.stderr, r.stderr[-600:]
    want, n_ref = _forward_records(ref_two, index)
    got, n_mine = _forward_records(my_two, index)
    assert len(want) > 50, "the case produces too few records to mean anything"
    data, mask = O.bitvectors_from_alleles(al)
    vet = util.double_root_vetter(data, mask, variants, N)
    util.assert_records_match(got, _as_oracle_records(want, variants), variants, double_root=vet)
    if not (set(map(tuple, np.stack([got["idxA"], got["idxB"]], 1).tolist())) ^ set(map(tuple, np.stack([want["idxA"], want["idxB"]], 1).tolist()))):
        assert n_ref == n_mine
    # and the reference reads what the engine wrote
    lines = [l for l in O.run_ref(["view", "-i", my_two, "-H"]).stdout.splitlines() if l and not l.startswith("FLAG\t")]
    assert len(lines) == n_mine


def test_off_diagonal_chunk_is_its_rectangle_of_the_full_run(tmp_path):
    """`-c 3 -C 2` is the off-diagonal chunk: row blocks [0, 3) against column blocks [3, 6).  The reference's ticker
    hands out the first column block of every row after the first as a *diagonal* block pair (ld_balancing.h:217-218,
    SURVEY A.6 q9), so its own output for such a chunk lacks those block pairs and repeats within-block pairs of the
    diagonal chunks; the engine computes the rectangle.  Checked against the reference's run of the whole triangle:
    the engine's chunk = the reference's records whose variants fall in the chunk's row and column blocks, and the three
    chunks together = the whole run."""
    N, M = 333, 450
    al, pos, rid = _dataset("iid", N, M, 12, 2)
    twk = str(tmp_path / "in.twk")
    hostlib.write_twk(twk, al, pos, rid, phased=np.ones(M, np.uint8), n_contigs=2, block_size=100)      # 3 blocks per contig
    variants = O.variants_from_alleles(al, pos=pos, rid=rid, phase=1)
    index = {(int(r), int(p)): i for i, (r, p) in enumerate(zip(rid, pos))}
    flags = ["-u", "-r", "0.005"]
    ref_two = str(tmp_path / "ref.two")
    O.run_ref(["calc", "-i", twk, "-o", ref_two, "-t", "4"] + flags, stdin=subprocess.DEVNULL)
    whole, _ = _forward_records(ref_two, index)
    data, mask = O.bitvectors_from_alleles(al)
    vet = util.double_root_vetter(data, mask, variants, N)
    n_first = int((rid == 0).sum())                       # blocks [0, 3) are contig 1
    parts = []
    for chunk, keep in (("1", (whole["idxA"] < n_first) & (whole["idxB"] < n_first)),
                        ("2", (whole["idxA"] < n_first) & (whole["idxB"] >= n_first)),
                        ("3", (whole["idxA"] >= n_first) & (whole["idxB"] >= n_first))):
        my_two = str(tmp_path / f"c{chunk}.two")
        r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", twk, "-o", my_two, "-c", "3", "-C", chunk] + flags, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        got, _ = _forward_records(my_two, index)
        assert keep.sum() > 50
        util.assert_records_match(got, _as_oracle_records(whole[keep], variants), variants, double_root=vet)
        parts.append(got)
    assert abs(sum(len(p) for p in parts) - len(whole)) <= 2            # (double-root pairs may be on one side only)


@pytest.mark.parametrize("w,compat", [(3000, True), (1700, True), (3000, False)])
def test_scalc_equals_the_reference_run_live(tmp_path, w, compat):
    """`scalc -I chr:pos -w W` (single site against its neighbourhood, scalc.h:50-194, ld.cpp:170-260).  The reference
    works through the neighbours in groups of 100 and drops the last partial group (q15): with TWK_REF_COMPAT=1 the two
    files hold the same records; without it the engine's file holds the reference's records plus the dropped remainder."""
    N, M = 333, 450
    al, pos, rid = _dataset("iid", N, M, 21, 2)
    twk = str(tmp_path / "in.twk")
    hostlib.write_twk(twk, al, pos, rid, phased=np.ones(M, np.uint8), n_contigs=2, block_size=100)
    target = f"1:{int(pos[120])}"
    ref_two, my_two = str(tmp_path / "ref.two"), str(tmp_path / "mine.two")
    ref = subprocess.run([O.REF_BIN, "scalc", "-i", twk, "-o", ref_two, "-I", target, "-w", str(w), "-t", "2"], capture_output=True, text=True,
                         stdin=subprocess.DEVNULL)
    r = subprocess.run([hostlib.CLI_PATH, "scalc", "-i", twk, "-o", my_two, "-I", target, "-w", str(w)], capture_output=True, text=True,
                       env=dict(os.environ, **(COMPAT if compat else {})))
    if ref.returncode != 0:             # fewer than 100 neighbours: the reference has no full group and gives up; so does the engine when asked to mirror it
        assert "no surrounding variants" in ref.stderr and compat
        assert r.returncode == 1 and "no surrounding variants" in r.stderr
        r = subprocess.run([hostlib.CLI_PATH, "scalc", "-i", twk, "-o", my_two, "-I", target, "-w", str(w)], capture_output=True, text=True)
        assert r.returncode == 0 and 40 < len(hostlib.read_two(my_two)[0]) < 200
        return
    assert r.returncode == 0, r.stderr
    a = hostlib.two_as_matrix(hostlib.read_two(ref_two)[0])
    b = hostlib.two_as_matrix(hostlib.read_two(my_two)[0])
    key = lambda m: [tuple(int(x) for x in row[1:5]) for row in m]
    ka, kb = key(a), key(b)
    assert len(set(ka)) == len(ka) and len(set(kb)) == len(kb) and len(ka) > 40
    if compat:
        assert set(ka) == set(kb)
    else:
        assert set(ka) <= set(kb) and len(kb) - len(ka) < 200            # the remainder of the last group of 100, both copies
    ib = {k: i for i, k in enumerate(kb)}
    bsel = b[[ib[k] for k in ka]]
    assert np.array_equal(a[:, 0], bsel[:, 0])                              # flags
    phased_math = (a[:, 0].astype(np.int64) & 1) == 1
    assert np.array_equal(a[phased_math, 5:9], bsel[phased_math, 5:9])      # integer counts, slot for slot
    # cubic-path records: each its own floors (tests/util.py cubic_floors: the root's rounding noise D_FLOOR through the record's dmax and allele frequencies)
    for row_a, row_b in zip(a, bsel):
        fl = util.cubic_floors(row_a[5:9], row_a[11], util.D_FLOOR)
        assert np.allclose(row_b[5:9], row_a[5:9], rtol=1e-6, atol=fl["cnt"]), (row_a, row_b)
        for col, name in ((9, "D"), (10, "Dprime"), (11, "R"), (12, "R2")):
            assert np.isclose(row_b[col], row_a[col], rtol=1e-6, atol=fl[name]), (name, row_a, row_b)
    np.testing.assert_allclose(bsel[:, 13], a[:, 13], rtol=1e-6, atol=1e-320)      # P
