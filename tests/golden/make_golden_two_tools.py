"""Golden fixtures for `view` / `sort` from the COMPILED REFERENCE.

Run in the dev container (needs oracle/_ref/tomahawk_ref):
    python tests/golden/make_golden_two_tools.py

  two_tools_input.two     unsorted 3-contig .two, 12,000 seeded records in blocks of 2,500, written by OUR writer
  two_tools_expected.json what the reference made of it:
      sort:  md5 of the sorted record bytes, index state, index block entries and per-contig entries
      view:  for each argument list, the number of output lines, the md5 of the text (the
             ##tomahawk_view* and ##tomahawk_sort* header lines, which carry dates, removed) and its first data lines;
             cases prefixed "sorted:" run on the reference-sorted file (interval queries need it)
Only data is stored: no reference source text.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O              # noqa: E402
from tomahawk_amd import hostlib as H       # noqa: E402

VIEW_CASES = [
    [], ["-H"], ["-h"], ["-r", "0.5"], ["-r", "0.2", "-R", "0.6", "-u"], ["-l"], ["-z", "0.3"], ["-Z", "0.2"],
    ["-p", "1e-3", "-P", "0.5"], ["-d", "-0.1", "-D", "0.1"], ["-b", "0.5", "-B", "0.9"], ["-1", "50", "-5", "80"],
    ["-2", "20"], ["-3", "5", "-7", "130"], ["-4", "30"], ["-a", "100", "-A", "300"], ["-x", "3", "-X", "10"],
    ["-m", "0", "-M", "1"], ["-f", "8"], ["-F", "1024"], ["-f", "3", "-F", "4"],
]
SORTED_VIEW_CASES = [
    ["-I", "2:1000-20000"], ["-I", "1:5000"], ["-I", "3"], ["-I", "1:1000-30000,2:50000-90000"],
    ["-I", "1:100-40000", "-I", "1:30000-50000", "-H"], ["-I", "2:1000-50000,2", "-r", "0.5"],
    ["-I", "1:10-20000", "-I", "3:60000-99000", "-u"],
]


def make_input(n=12000, seed=2024):
    rng = np.random.default_rng(seed)
    r = np.zeros(n, dtype=H.TWO_DTYPE)
    r["controller"] = rng.integers(1, 4096, n)
    r["ridA"] = rng.integers(0, 3, n)
    r["ridB"] = np.where(rng.random(n) < 0.8, r["ridA"], rng.integers(0, 3, n))
    r["packA"] = (rng.integers(0, 100000, n).astype(np.uint32) << 2) | rng.integers(0, 4, n).astype(np.uint32)
    r["packB"] = (rng.integers(0, 100000, n).astype(np.uint32) << 2) | rng.integers(0, 4, n).astype(np.uint32)
    r["cnt"] = rng.integers(0, 200, (n, 4))
    q = lambda x: np.round(x, 4)
    r["R"] = q(rng.random(n) * 2 - 1); r["R2"] = q(r["R"] ** 2)
    r["D"] = q(rng.random(n) - 0.5); r["Dprime"] = q(rng.random(n))
    r["P"] = q(rng.random(n) ** 6); r["ChiSqFisher"] = q(rng.random(n) * 30); r["ChiSqModel"] = q(rng.random(n) * (rng.random(n) < 0.3))
    return r


def strip_dated(text: bytes) -> bytes:
    return b"".join(l for l in text.splitlines(keepends=True) if not l.startswith((b"##tomahawk_view", b"##tomahawk_sort")))


def run_view(path, args):
    p = subprocess.run([O.REF_BIN, "view", "-i", path] + args, capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    return strip_dated(p.stdout)


def main():
    assert O.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    inp = os.path.join(HERE, "two_tools_input.two")
    H.write_two(inp, make_input(), n_samples=10, n_contigs=3, block_records=2500)
    exp = {"view": [], "sort": {}}
    with tempfile.TemporaryDirectory() as d:
        srt = os.path.join(d, "sorted.two")
        p = subprocess.run([O.REF_BIN, "sort", "-i", inp, "-o", srt, "-t", "2"], capture_output=True)
        assert p.returncode == 0, p.stderr.decode()
        recs, info = H.read_two(srt)
        state, ent, ctg = H.two_index(srt)
        exp["sort"] = dict(md5=hashlib.md5(recs.tobytes()).hexdigest(), n=int(len(recs)), state=state,
                           entries=ent.tolist(), contigs=ctg.tolist())
        for tag, path, cases in (("", inp, VIEW_CASES), ("sorted:", srt, SORTED_VIEW_CASES)):
            for args in cases:
                out = run_view(path, args)
                lines = out.splitlines()
                data = [l.decode() for l in lines if not l.startswith(b"#")]
                exp["view"].append(dict(on=tag or "input:", args=args, n_lines=len(lines), md5=hashlib.md5(out).hexdigest(),
                                        head=data[:3]))
    with open(os.path.join(HERE, "two_tools_expected.json"), "w") as f:
        json.dump(exp, f, indent=1)
    print("wrote", inp, os.path.getsize(inp), "bytes;", len(exp["view"]), "view cases")


if __name__ == "__main__":
    main()
