"""Generate the golden fixtures from the COMPILED REFERENCE (oracle/_ref/tomahawk_ref).

Run in the dev container (needs /root/reference to have been built by `make -C oracle ref`):
    python tests/golden/make_golden.py

For every case: seeded genotypes -> .twk written by OUR writer (libtomahawk_amd) -> the reference
reads it (`twkinfo`: format parity of the writer), runs `calc -r 0 -t 1` forced phased / unphased /
default, and `dump`s every record with %.17g.  Stored per case in tests/golden/<case>.npz:
    alleles   int8 [M, N, 2]   (0 ref, 1 alt, 2 missing)
    pos, rid  uint32 [M]
    info      what the reference reader saw per variant (ac, an, n_het, n_hom, phase, missing, ptype, n_runs)
    rec_p / rec_u / rec_d   float64 [n, 16] reference records (-p / -u / default), forward copies only,
                            columns: flags ridA Apos ridB Bpos cnt0..3 D Dprime R R2 P ChiSqFisher ChiSqModel
One reference-written .two (case n64_small, -p) is kept verbatim as a reader fixture.
Only data is stored: no reference source text.
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O          # noqa: E402
from tests import util                  # noqa: E402
from tomahawk_amd import hostlib        # noqa: E402

CASES = {
    # name: (N, M, seed, kwargs for util.random_alleles, n_contigs, block_size)
    "n64_small":    (64, 60, 101, dict(low_ac=5), 1, 25),
    "n100_pad":     (100, 90, 102, dict(low_ac=5), 1, 40),      # N % 64 != 0: padding corrections
    "n1000":        (1000, 110, 103, dict(low_ac=6), 2, 50),    # two contigs
    "n64_missing":  (64, 70, 104, dict(miss_rate=0.1, miss_variants=0.35, low_ac=4), 1, 30),
    "n128_missing": (128, 80, 105, dict(miss_rate=0.06, miss_variants=0.3, low_ac=4), 1, 30),
    "n64_scalc":    (64, 300, 106, dict(miss_rate=0.1, miss_variants=0.1), 1, 64),
    # haplotype mosaics: strong LD, identical and complementary variants, D' = 1 pairs, double roots of the cubic
    "n64_ld":       (64, 90, 107, dict(mosaic=True), 1, 30),
    "n500_ld":      (500, 100, 108, dict(mosaic=True, switch=0.01), 1, 40),
    "n128_ld_miss": (128, 80, 109, dict(mosaic=True, miss_rate=0.05, miss_variants=0.3), 1, 30),
    # missing genotypes with 2N NOT a multiple of 128: -p goes through PhasedVectorized's scalar tail and padding
    # correction, whose slips the reference's records carry; -u / default are unaffected
    "n100_miss":    (100, 80, 110, dict(miss_rate=0.08, miss_variants=0.35, low_ac=4), 1, 30),       # 2N = 200: two tail words
    "n70_ld_miss":  (70, 70, 111, dict(mosaic=True, miss_rate=0.06, miss_variants=0.3), 1, 30),      # 2N = 140: one tail word
    "n161_miss":    (161, 60, 112, dict(miss_rate=0.05, miss_variants=0.4), 1, 30),                  # 2N = 322 = 2*128 + 66
}


def parse_dump(text):
    rows = [l.split("\t") for l in text.splitlines() if l and not l.startswith("#")]
    if not rows:
        return np.zeros((0, 16))
    return np.array(rows, dtype=np.float64)


def forward_only(rec):
    """calc writes every pair twice (A,B) and (B,A) (ld_engine.cpp:1290-1298); keep (rid,pos)A < (rid,pos)B."""
    keyA = rec[:, 1] * 2**32 + rec[:, 2]
    keyB = rec[:, 3] * 2**32 + rec[:, 4]
    f = rec[keyA < keyB]
    order = np.lexsort((f[:, 4], f[:, 3], f[:, 2], f[:, 1]))
    return f[order]


def main():
    assert O.have_ref(), "build the reference first: make -C oracle ref"
    tmp = tempfile.mkdtemp(prefix="twk_golden_")
    only = set(sys.argv[1:])            # optional: names of the cases to (re)generate
    for name, (N, M, seed, kw, n_contigs, bsize) in CASES.items():
        if only and name not in only:
            continue
        kw = dict(kw)
        al = util.mosaic_alleles(M, N, seed, **kw) if kw.pop("mosaic", False) else util.random_alleles(M, N, seed, **kw)
        rid = np.sort(np.arange(M) * n_contigs // M).astype(np.uint32)
        pos = np.zeros(M, dtype=np.uint32)
        for r in range(n_contigs):
            k = np.nonzero(rid == r)[0]
            pos[k] = 1000 + 100 * np.arange(len(k))
        twk = os.path.join(tmp, name + ".twk")
        hostlib.write_twk(twk, al, pos, rid, phased=np.ones(M, np.uint8), n_contigs=n_contigs, block_size=bsize)
        info = O.run_ref(["twkinfo", twk]).stdout
        rows = [l.split("\t") for l in info.splitlines() if not l.startswith("#")]
        info_arr = np.array([[float(x) for x in r] for r in rows])
        # rid pos ac an n_het n_hom phase missing ptype hwe n_runs
        assert len(info_arr) == M and (info_arr[:, 1] == pos).all() and (info_arr[:, 0] == rid).all()
        assert (info_arr[:, 2] == (al == 1).sum(axis=(1, 2))).all() and (info_arr[:, 3] == (al == 2).sum(axis=(1, 2))).all()
        out = {}
        for tag, flag in ((("p", ["-p"]), ("u", ["-u"]), ("d", [])) if name != "n64_scalc" else ()):
            two = os.path.join(tmp, f"{name}_{tag}.two")
            O.run_ref(["calc", "-i", twk, "-o", two, "-r", "0", "-t", "1"] + flag)
            rec = parse_dump(O.run_ref(["dump", two]).stdout)
            out["rec_" + tag] = forward_only(rec)
            assert len(rec) == 2 * len(out["rec_" + tag])
            if name == "n64_small" and tag == "p":
                shutil.copy(two, os.path.join(HERE, "ref_n64_small_p.two"))
        if name == "n1000":
            # interval slicing and single-site mode (calc -I, scalc): two contigs, 55 variants each, 50/block
            for tag, args in (("I_contig2", ["calc", "-I", "2", "-p"]), ("I_range", ["calc", "-I", "1:2500-4300", "-u"])):
                two = os.path.join(tmp, f"{name}_{tag}.two")
                O.run_ref(args[:1] + ["-i", twk, "-o", two, "-r", "0", "-t", "1"] + args[1:])
                rec = parse_dump(O.run_ref(["dump", two]).stdout)
                out["rec_" + tag] = forward_only(rec)
        if name == "n64_scalc":
            # scalc: target = variant 150 (0-based pos 16000 -> 1-based 16001), +-5000 bp = exactly 100 neighbours.
            # (The reference only keeps neighbours in full groups of 100, ld.cpp:203-205,239-244.)
            two = os.path.join(tmp, f"{name}_scalc.two")
            O.run_ref(["scalc", "-i", twk, "-o", two, "-t", "1", "-I", "1:16001", "-w", "5000"])
            out["rec_scalc"] = parse_dump(O.run_ref(["dump", two]).stdout)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), alleles=al, pos=pos, rid=rid,
                            info=info_arr[:, [2, 3, 4, 5, 6, 7, 8, 10]].astype(np.uint32), **out)
        print(name, {k: v.shape for k, v in out.items()})
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
