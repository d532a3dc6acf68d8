"""`view` and `sort` of .two files: host tools, no GPU needed.

Parity bar: byte-identical text output and byte-identical sorted records / index entries against
the reference, (a) through the committed fixtures the compiled reference produced
(tests/golden/make_golden_two_tools.py) and (b) live against oracle/_ref where it exists.
"""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from tomahawk_amd import hostlib as H

GOLD = os.path.join(os.path.dirname(__file__), "golden")
INPUT = os.path.join(GOLD, "two_tools_input.two")
with open(os.path.join(GOLD, "two_tools_expected.json")) as _f:
    EXPECTED = json.load(_f)


def strip_dated(text: bytes) -> bytes:
    return b"".join(l for l in text.splitlines(keepends=True) if not l.startswith((b"##tomahawk_view", b"##tomahawk_sort")))


def view(path, args, binary=None, **kw):
    p = subprocess.run([binary or H.CLI_PATH, "view", "-i", path] + list(args), capture_output=True, **kw)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout


@pytest.fixture(scope="module")
def sorted_file(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("sort") / "sorted.two")
    H.sort_two(INPUT, out, n_threads=3)
    return out


def test_sort_matches_reference_fixture(sorted_file):
    exp = EXPECTED["sort"]
    recs, info = H.read_two(sorted_file)
    assert len(recs) == exp["n"] and info["state"] == exp["state"] == 2
    assert hashlib.md5(recs.tobytes()).hexdigest() == exp["md5"]           # same order, same bytes
    state, ent, ctg = H.two_index(sorted_file)
    assert state == 2 and ent.tolist() == exp["entries"] and ctg.tolist() == exp["contigs"]


@pytest.mark.parametrize("case", EXPECTED["view"], ids=lambda c: c["on"] + " ".join(c["args"]))
def test_view_matches_reference_fixture(case, sorted_file):
    path = INPUT if case["on"] == "input:" else sorted_file
    out = strip_dated(view(path, case["args"]))
    lines = out.splitlines()
    assert len(lines) == case["n_lines"]
    assert [l.decode() for l in lines if not l.startswith(b"#")][:3] == case["head"]
    assert hashlib.md5(out).hexdigest() == case["md5"]


def test_sort_properties_and_external_merge(tmp_path, monkeypatch):
    """Sortedness, permutation, block cuts, per-contig index; the external (multi-run) path gives the same file."""
    rng = np.random.default_rng(7)
    n = 60000
    r = np.zeros(n, dtype=H.TWO_DTYPE)
    r["controller"] = 1
    r["ridA"] = rng.integers(0, 4, n); r["ridB"] = rng.integers(0, 4, n)
    r["packA"] = rng.integers(0, 5000, n).astype(np.uint32) << 2
    r["packB"] = rng.integers(0, 5000, n).astype(np.uint32) << 2
    r["R2"] = rng.random(n)
    src = str(tmp_path / "in.two")
    H.write_two(src, r, n_samples=4, n_contigs=4, block_records=3000)
    a = str(tmp_path / "a.two")
    H.sort_two(src, a, n_threads=4)
    sa, ia = H.read_two(a)
    key = lambda x: np.stack([x["ridA"], x["ridB"], x["packA"] >> 2, x["packB"] >> 2], 1).astype(np.int64)
    ka = key(sa)
    order = np.lexsort((ka[:, 3], ka[:, 2], ka[:, 1], ka[:, 0]))
    assert np.array_equal(ka, ka[order])                                           # sorted by (ridA, ridB, Apos, Bpos)
    rows = lambda x: np.sort(np.frombuffer(x.tobytes(), dtype="S106"))
    assert np.array_equal(rows(sa), rows(r))                                       # a permutation of the input
    state, ent, ctg = H.two_index(a)
    assert state == 2 and ent[:, 2].sum() == n and ent[:, 2].max() <= 10000
    off = 0
    for rid, ridB, cnt, minpos, maxpos, b_unc in ent.tolist():
        blk = sa[off:off + cnt]; off += cnt
        assert (blk["ridA"] == rid).all() and b_unc == 8 + 106 * cnt           # a block never spans two ridA
        assert minpos == blk["packA"][0] >> 2 and maxpos == blk["packA"][-1] >> 2
        assert ridB == (blk["ridB"][0] if (blk["ridB"] == blk["ridB"][0]).all() else -1)
    for rid in range(4):
        assert ctg[rid, 0] == rid and ctg[rid, 1] == (sa["ridA"] == rid).sum() and ctg[rid, 4] == (ent[:, 0] == rid).sum()
    # external path: a memory limit of a few kilobytes makes runs of <= 1000 records (the floor), merged
    b = str(tmp_path / "b.two")
    H.sort_two(src, b, memory_limit_gb=1e-6, n_threads=2)
    sb, _ = H.read_two(b)
    assert sa.tobytes() == sb.tobytes()
    assert np.array_equal(H.two_index(b)[1], ent)
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".tmp")]             # run files are removed


def test_view_binary_roundtrip_and_errors(tmp_path, sorted_file):
    # -O b keeps what the text view shows, as a readable .two with a sorted index
    out = str(tmp_path / "sub.two")
    p = subprocess.run([H.CLI_PATH, "view", "-i", sorted_file, "-O", "b", "-o", out, "-r", "0.5", "-I", "2"], capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    sub, info = H.read_two(out)
    full, _ = H.read_two(sorted_file)
    want = full[(full["ridA"] == 1) & (full["R2"] >= 0.5)]
    assert info["state"] == 2 and sub.tobytes() == want.tobytes()
    a = strip_dated(view(out, ["-H"])); b = strip_dated(view(sorted_file, ["-H", "-r", "0.5", "-I", "2"]))
    assert a == b
    # interval queries need a sorted index, as in the reference (index.cpp:231-240 finds no block otherwise)
    p = subprocess.run([H.CLI_PATH, "view", "-i", INPUT, "-I", "1:1000-2000"], capture_output=True)
    assert p.returncode == 1 and b"Found no blocks overlapping" in p.stderr
    p = subprocess.run([H.CLI_PATH, "view", "-i", sorted_file, "-I", "nope:1-2"], capture_output=True)
    assert p.returncode == 1 and b"Contig does not exist" in p.stderr
    p = subprocess.run([H.CLI_PATH, "view", "-i", sorted_file, "-r", "abc"], capture_output=True)
    assert p.returncode == 1 and b"not a valid float" in p.stderr
    p = subprocess.run([H.CLI_PATH, "view", "-i", str(tmp_path / "missing.two")], capture_output=True)
    assert p.returncode == 1
    # one worker thread and many give the same bytes
    assert view(sorted_file, ["-H", "-t", "1"]) == view(sorted_file, ["-H", "-t", "7"])


@pytest.mark.skipif(not O.have_ref(), reason="compiled reference (oracle/_ref) not available")
def test_view_and_sort_live_against_reference(tmp_path):
    """On a calc-shaped file written by the reference itself (golden fixture ref_n64_small_p.two)."""
    src = os.path.join(GOLD, "ref_n64_small_p.two")
    mine, ref = str(tmp_path / "mine.two"), str(tmp_path / "ref.two")
    subprocess.run([H.CLI_PATH, "sort", "-i", src, "-o", mine, "-t", "2"], check=True, capture_output=True)
    subprocess.run([O.REF_BIN, "sort", "-i", src, "-o", ref, "-t", "2"], check=True, capture_output=True)
    assert H.read_two(mine)[0].tobytes() == H.read_two(ref)[0].tobytes()
    assert all(np.array_equal(x, y) for x, y in zip(H.two_index(mine)[1:], H.two_index(ref)[1:]))
    for path, args in [(src, []), (src, ["-u", "-r", "0.05"]), (src, ["-a", "10", "-A", "40"]), (ref, ["-I", "1:1000-1500,1:3000-4000"]),
                       (mine, ["-I", "1:1200-1300", "-I", "1:1250-2000", "-H"]), (mine, ["-I", "1", "-l"])]:
        assert strip_dated(view(path, args)) == strip_dated(view(path, args, binary=O.REF_BIN)), args


def test_corrupt_files_fail_cleanly(tmp_path):
    """Byte flips, truncations and absurd sizes: exit code 0 or 1, never a crash or a runaway allocation."""
    import random
    src = open(os.path.join(GOLD, "ref_n64_small_p.two"), "rb").read()
    rnd = random.Random(5)
    path = str(tmp_path / "fz.two")
    for it in range(45):
        b = bytearray(src)
        if it % 3 == 0:
            for _ in range(rnd.randint(1, 5)):
                b[rnd.randrange(len(b))] = rnd.randrange(256)
        elif it % 3 == 1:
            b = b[: rnd.randrange(4, len(b))]
        else:
            i = rnd.randrange(len(b) - 8)
            b[i:i + 4] = b"\xff\xff\xff\x7f"
        open(path, "wb").write(b)
        p = subprocess.run([H.CLI_PATH, "view", "-i", path, "-H", "-t", "2"], capture_output=True, timeout=60)
        assert p.returncode in (0, 1), (it, p.returncode, p.stderr[-200:])
        try:
            H.read_two(path)
        except RuntimeError:
            pass


@pytest.mark.skipif(not O.have_ref(), reason="compiled reference (oracle/_ref) not available")
def test_concat_live_against_reference(tmp_path):
    """`concat`: same record stream and index entries as the reference's concat()."""
    parts = []
    for k, n in enumerate((1000, 700, 10)):
        g = np.random.default_rng(k)
        r = np.zeros(n, dtype=H.TWO_DTYPE)
        r["controller"] = g.integers(1, 100, n); r["ridA"] = g.integers(0, 2, n); r["ridB"] = r["ridA"]
        r["packA"] = g.integers(0, 1000, n).astype(np.uint32) << 2; r["packB"] = g.integers(0, 1000, n).astype(np.uint32) << 2
        r["R2"] = g.random(n)
        parts.append(str(tmp_path / f"p{k}.two"))
        H.write_two(parts[-1], r, n_samples=5, n_contigs=2, block_records=300)
    args = [x for p in parts for x in ("-i", p)]
    mine, ref = str(tmp_path / "mine.two"), str(tmp_path / "ref.two")
    subprocess.run([H.CLI_PATH, "concat"] + args + ["-o", mine], check=True, capture_output=True)
    subprocess.run([O.REF_BIN, "concat"] + args + ["-o", ref], check=True, capture_output=True)
    assert H.read_two(mine)[0].tobytes() == H.read_two(ref)[0].tobytes()
    a, b = H.two_index(mine), H.two_index(ref)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
