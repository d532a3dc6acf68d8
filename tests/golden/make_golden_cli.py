"""Golden fixtures for the `tomahawk calc` command line, from the COMPILED REFERENCE's own CLI
(oracle/_ref/tomahawk_ref), on haplotype-block data: 2,504 samples x 3,000 variants on two contigs,
15 % of the variants with missing samples (the case DESIGN.md quotes for its end-to-end comparison).

Run in the dev container (needs `make -C oracle ref`):
    python tests/golden/make_golden_cli.py

tests/golden/cli_mosaic_input.npz     alleles (bit-packed), pos, rid: the .twk is rebuilt from it by OUR writer
tests/golden/cli_mosaic_<case>.npz    what `tomahawk_ref calc <flags>` wrote, forward copies only, sorted by pair:
    idxA, idxB   uint16 [n]   the whole pair set (variant indices in file order)
    flags        uint16 [n]   twk1_two_t.controller of every record
    sample       float64 [n/16, 16]  every 16th record in full (%.17g): flags ridA Apos ridB Bpos cnt0..3 D Dprime R R2 P ChiSqFisher ChiSqModel
    sums         float64 [11] column sums of cnt0..3 D Dprime R R2 P ChiSqFisher ChiSqModel over ALL records
    n_file       number of records in the reference's file (forward + reverse)
tests/golden/cli_mosaic_chain.json.gz  text the reference's `sort` + `view -H` print for one calc output, and its
                                       `concat` of two `-c 3` chunks (sorted lines)
Only data is stored: no reference source text.
"""
import gzip
import json
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
from tests.golden.make_golden import parse_dump, forward_only   # noqa: E402

N, M = 2504, 3000
CASES = {       # name: reference CLI flags
    "u":        ["-u"],
    "default":  [],
    "u_P":      ["-u", "-P", "1e-10"],
    "u_r03":    ["-u", "-r", "0.3"],
    "I_range":  ["-I", "1:5000-40000"],
    "I_contig": ["-I", "2"],
    "c3C1":     ["-c", "3", "-C", "1"],
    "c3C3":     ["-c", "3", "-C", "3"],
    "c6C6":     ["-c", "6", "-C", "6"],
    "w20000":   ["-w", "20000"],
    "w5000":    ["-w", "5000"],
    "u_w5000":  ["-u", "-w", "5000"],        # forced mode: the slave's per-pair window test and its goto end_cycle (q8)
    "u_w1000":  ["-u", "-w", "1000"],
    "w1000":    ["-w", "1000"],              # default mode: the window only prunes whole block pairs
}
CHAIN_CALC = ["-p", "-r", "0.75", "-I", "2"]        # phased, no missing-data pairs involved in the comparison below


def build_input():
    al = util.mosaic_alleles(M, N, 4242, n_founders=8, switch=0.01, mut=0.002, miss_rate=0.01, miss_variants=0.15)
    pos = (1000 + 37 * np.arange(M)).astype(np.uint32)
    rid = (np.arange(M) >= 2000).astype(np.uint32)
    pos[2000:] -= pos[2000] - 500
    return al, pos, rid


def write_input_twk(path, al, pos, rid):
    hostlib.write_twk(path, al, pos, rid, phased=np.ones(len(pos), np.uint8), n_contigs=2, block_size=200)


def main():
    assert O.have_ref(), "build the reference first: make -C oracle ref"
    tmp = tempfile.mkdtemp(prefix="twk_golden_cli_")
    al, pos, rid = build_input()
    # 2 bits per allele would do; packbits of the three one-hot planes compresses better
    np.savez_compressed(os.path.join(HERE, "cli_mosaic_input.npz"), alt=np.packbits(al == 1), miss=np.packbits(al == 2),
                        shape=np.array(al.shape), pos=pos, rid=rid)
    twk = os.path.join(tmp, "in.twk")
    write_input_twk(twk, al, pos, rid)
    index = {(int(r), int(p)): i for i, (r, p) in enumerate(zip(rid, pos))}
    for name, flags in CASES.items():
        two = os.path.join(tmp, name + ".two")
        O.run_ref(["calc", "-i", twk, "-o", two, "-t", "8"] + flags)
        rec = parse_dump(O.run_ref(["dump", two]).stdout)
        fwd = forward_only(rec)
        assert len(rec) == 2 * len(fwd)
        idxA = np.array([index[(int(a), int(b))] for a, b in zip(fwd[:, 1], fwd[:, 2])], dtype=np.uint16)
        idxB = np.array([index[(int(a), int(b))] for a, b in zip(fwd[:, 3], fwd[:, 4])], dtype=np.uint16)
        np.savez_compressed(os.path.join(HERE, f"cli_mosaic_{name}.npz"), idxA=idxA, idxB=idxB, flags=fwd[:, 0].astype(np.uint16),
                            sample=fwd[::16], sums=fwd[:, 5:16].sum(axis=0), n_file=np.array(len(rec)),
                            cli_flags=np.array(flags, dtype="U16"))
        print(name, flags, len(fwd), "forward records")
    # the tools behind calc: sort + view, concat
    chain = {}
    two = os.path.join(tmp, "chain.two")
    O.run_ref(["calc", "-i", twk, "-o", two, "-t", "1"] + CHAIN_CALC)
    srt = os.path.join(tmp, "chain_sorted.two")
    O.run_ref(["sort", "-i", two, "-o", srt])
    chain["calc_flags"] = CHAIN_CALC
    chain["sorted_view"] = O.run_ref(["view", "-i", srt, "-H"]).stdout.splitlines()
    chain["sorted_view_interval"] = O.run_ref(["view", "-i", srt, "-H", "-I", "2:1000-9000", "-r", "0.9"]).stdout.splitlines()
    parts = []
    for k in ("1", "3"):
        part = os.path.join(tmp, f"chunk{k}.two")
        O.run_ref(["calc", "-i", twk, "-o", part, "-t", "1", "-p", "-r", "0.8", "-c", "3", "-C", k])
        parts.append(part)
    cat = os.path.join(tmp, "cat.two")
    O.run_ref(["concat", "-i", parts[0], "-i", parts[1], "-o", cat])
    chain["concat_calc_flags"] = ["-p", "-r", "0.8", "-c", "3"]
    chain["concat_view_sorted_lines"] = sorted(O.run_ref(["view", "-i", cat, "-H"]).stdout.splitlines())
    with gzip.open(os.path.join(HERE, "cli_mosaic_chain.json.gz"), "wt") as f:
        json.dump(chain, f)
    print("chain:", len(chain["sorted_view"]), "sorted view lines,", len(chain["sorted_view_interval"]), "interval lines,",
          len(chain["concat_view_sorted_lines"]), "concat lines")
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
