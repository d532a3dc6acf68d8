"""`tomahawk import`: VCF text -> .twk without htslib.  Host code, no GPU needed.

The reference importer needs htslib and cannot be built here, so the end-to-end output has no live
reference twin ("parity unpinned" for the VCF front end); what is pinned:
  * the Hardy-Weinberg exact test against the compiled reference (twk1_t::calculateHardyWeinberg)
    and against committed known answers it produced;
  * the produced .twk against the compiled reference's *reader* (`twkinfo`: per-variant fields);
  * the genotypes, counts and filters against what the test wrote into the VCF.
"""
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from tests import util
from tomahawk_amd import hostlib as H

GOLD = os.path.join(os.path.dirname(__file__), "golden")
BASES = "ATGC"


def write_vcf(path, alleles, pos, chrom, phased=True, contigs=("20", "21"), ref_alt=None, extra_format=False, gz=False,
              raw_lines=None):
    """alleles int8 [M, N, 2] in {0, 1, 2=missing}; pos 1-based; chrom: index into contigs per site."""
    M, N, _ = alleles.shape
    out = ["##fileformat=VCFv4.2"]
    out += [f"##contig=<ID={c},length={63025520 + i},assembly=b37>" for i, c in enumerate(contigs)]
    out += ['##INFO=<ID=AC,Number=A,Type=Integer,Description="Allele count">',
            '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
            '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Depth, with a comma">']
    out.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"S{i}" for i in range(N)))
    sym = np.array(["0", "1", "."])
    ph = np.broadcast_to(np.asarray(phased, dtype=bool), (M,)) if np.ndim(phased) <= 1 else None
    for v in range(M):
        sep = ("|" if ph[v] else "/") if ph is not None else None
        a, b = sym[alleles[v, :, 0]], sym[alleles[v, :, 1]]
        if sep is not None:
            gts = np.char.add(np.char.add(a, sep), b)
        else:
            gts = np.char.add(np.char.add(a, np.where(np.asarray(phased)[v], "|", "/")), b)
        if extra_format:
            gts = np.char.add(gts, ":17")
        ra = ref_alt[v] if ref_alt is not None else (BASES[v % 4], BASES[(v + 1) % 4])
        out.append(f"{contigs[chrom[v]]}\t{pos[v]}\trs{v}\t{ra[0]}\t{ra[1]}\t.\tPASS\tAC=1\t{'GT:DP' if extra_format else 'GT'}\t"
                   + "\t".join(gts.tolist()))
    for at, line in (raw_lines or []):
        out.insert(len(out) - M + at, line)
    text = ("\n".join(out) + "\n").encode()
    with (gzip.open(path, "wb") if gz else open(path, "wb")) as f:
        f.write(text)


def read_back(path):
    N, data, mask, meta, extra = H.read_twk(path)
    return N, data, mask, meta, extra


@pytest.mark.parametrize("gz,extra_format", [(False, False), (True, True)])
def test_import_roundtrip_matches_direct_writer(tmp_path, gz, extra_format):
    """VCF -> import -> .twk decodes to the genotypes written, with the fields the reference encoder fills."""
    N, M = 150, 400
    al = util.random_alleles(M, N, 11, miss_rate=0.03, miss_variants=0.25)
    pos = np.arange(M) * 37 + 101
    chrom = (np.arange(M) >= 250).astype(int)
    pos[250:] -= pos[250] - 55
    phased = np.random.default_rng(5).random(M) < 0.7
    vcf = str(tmp_path / ("in.vcf.gz" if gz else "in.vcf"))
    write_vcf(vcf, al, pos, chrom, phased=phased, extra_format=extra_format, gz=gz)
    out = str(tmp_path / "out.twk")
    cnt = H.import_vcf(vcf, out, threshold_miss=0.5, remove_univariate=False, block_size=64, n_threads=3)
    assert cnt["sites"] == M and cnt["written"] == M and sum(cnt[k] for k in H.IMPORT_COUNTERS[:10]) == 0
    n, data, mask, meta, extra = read_back(out)
    want_data, want_mask = O.bitvectors_from_alleles(al)
    assert n == N and np.array_equal(data, want_data) and np.array_equal(mask, want_mask)
    assert np.array_equal(meta["pos"], pos - 1) and np.array_equal(meta["rid"], chrom)          # 0-based like bcf1_t::pos
    assert np.array_equal(meta["ac"], (al == 1).sum(axis=(1, 2))) and np.array_equal(meta["an"], (al == 2).sum(axis=(1, 2)))
    complete = (al != 2).all(axis=2)
    het = ((al[:, :, 0] != al[:, :, 1]) & complete).sum(1); hom = ((al[:, :, 0] == 1) & (al[:, :, 1] == 1)).sum(1)
    assert np.array_equal(extra[:, 0], het) and np.array_equal(extra[:, 1], hom)
    assert np.array_equal(extra[:, 2], phased.astype(np.uint32))                                  # uniform phase per site
    hom1 = ((al[:, :, 0] == 0) & (al[:, :, 1] == 0)).sum(1)
    assert np.array_equal(meta["hwe"], [H.hwe_exact(int(a), int(b), int(c)) for a, b, c in zip(hom1, het, hom)])
    # the same genotypes through the direct writer decode identically
    direct = str(tmp_path / "direct.twk")
    H.write_twk(direct, al, pos - 1, chrom, phased, n_contigs=2, block_size=64)
    n2, d2, m2, meta2, extra2 = read_back(direct)
    assert np.array_equal(d2, data) and np.array_equal(m2, mask) and np.array_equal(extra2, extra)
    lit = H.header_literals(out, is_two=False)
    assert lit.startswith("##fileformat=VCFv4.2\n##FILTER=<ID=PASS,") and "##contig=<ID=21,length=63025521,assembly=b37>" in lit
    assert "##tomahawk_importVersion=" in lit and "#CHROM" not in lit


def test_import_site_filters(tmp_path):
    """Every drop reason of the reference importer (importer.cpp:121-196, genotype_encoder.h:197-275), in file order."""
    N = 40
    rng = np.random.default_rng(3)
    base = (rng.random((1, N, 2)) < 0.3).astype(np.int8)
    def site(): return (rng.random((N, 2)) < 0.3).astype(np.int8)
    rows, pos, ra, expect = [], [], [], []
    def add(al, p, r=("A", "G"), why=None):
        rows.append(al); pos.append(p); ra.append(r); expect.append(why)
    add(site(), 100)                                           # kept
    add(site(), 100, why="duplicates")                         # same position as a kept site
    add(site(), 100)                                           # third record at the position: kept again (prev was dropped)
    add(np.zeros((N, 2), np.int8), 200, why="invariant")
    add(np.ones((N, 2), np.int8), 210, why="invariant")
    a = np.zeros((N, 2), np.int8); a[:, 1] = 1
    add(a, 220, why="invariant")                               # every sample 0|1: one haplotype class
    m = site(); m[: N // 2] = 2
    add(m, 300, why="missing_threshold")                       # 50 % complete < 0.9
    add(site(), 400, r=("A", "G,T"), why="not_biallelic")
    add(site(), 410, r=("A", "."), why="not_biallelic")
    add(site(), 420, r=("AT", "G"), why="not_snp")
    add(site(), 430, r=("A", "<DEL>"), why="not_snp")
    add(site(), 440, r=("N", "G"), why="not_snp")
    h = site(); h[0] = (0, 0); h[1] = (1, 1); h[2:] = (0, 1)   # far too many heterozygotes
    add(h, 500, why="hwe")
    add(site(), 600)                                           # kept
    al = np.stack(rows); M = len(rows)
    vcf = str(tmp_path / "f.vcf")
    hap = "20\t700\t.\tA\tG\t.\t.\t.\tGT\t" + "\t".join(["0"] + ["0|1"] * (N - 1))          # one haploid call
    dot = "20\t710\t.\tA\tG\t.\t.\t.\tGT\t" + "\t".join(["."] + ["0|1", "0|0"] * ((N - 1) // 2) + ["1|1"])   # "." is haploid-missing
    nogt = "20\t720\t.\tA\tG\t.\t.\t.\tDP\t" + "\t".join(["7"] * N)
    nofmt = "20\t730\t.\tA\tG\t.\t.\t."
    write_vcf(vcf, al, pos, np.zeros(M, int), ref_alt=ra,
              raw_lines=[(M, hap), (M + 1, dot), (M + 2, nogt), (M + 3, nofmt)])
    out = str(tmp_path / "f.twk")
    cnt = H.import_vcf(vcf, out, hwe=1e-4, n_threads=2)
    want = {k: 0 for k in H.IMPORT_COUNTERS}
    for w in expect:
        if w: want[w] += 1
    want.update(mixed_ploidy=2, no_genotypes=1, no_format=1, sites=M + 4, written=sum(w is None for w in expect))
    assert cnt == want
    _, data, mask, meta, _ = read_back(out)
    assert meta["pos"].tolist() == [p - 1 for p, w in zip(pos, expect) if w is None]


def test_import_phase_rules_and_small_inputs(tmp_path):
    N = 12
    al = np.zeros((3, N, 2), np.int8); al[:, :6, 0] = 1; al[1, 7] = 2
    vcf = str(tmp_path / "p.vcf")
    # site 0: all "/" -> unphased; site 1: all "|" with one "./." -> phased; site 2: mixed separators -> unphased
    ph = np.ones((3, N), bool); ph[0] = False; ph[2, 5] = False
    write_vcf(vcf, al, [10, 20, 30], [0, 0, 0], phased=ph)
    out = str(tmp_path / "p")                                    # extension is forced (importer.cpp:57-66)
    cnt = H.import_vcf(vcf, out, threshold_miss=0.5)
    assert cnt["written"] == 3 and os.path.exists(out + ".twk")
    _, _, _, meta, extra = read_back(out + ".twk")
    assert extra[:, 2].tolist() == [0, 1, 0] and meta["missing"].tolist() == [0, 1, 0]
    # fewer than 5 complete genotypes
    al4 = np.zeros((1, 4, 2), np.int8); al4[0, 0, 0] = 1
    write_vcf(vcf, al4, [10], [0])
    assert H.import_vcf(vcf, out)["insufficient_samples"] == 1
    # not a VCF / no genotypes declared
    bad = str(tmp_path / "bad.vcf")
    open(bad, "w").write("hello\n")
    with pytest.raises(RuntimeError):
        H.import_vcf(bad, out)
    p = subprocess.run([H.CLI_PATH, "import", "-i", bad, "-o", out], capture_output=True)
    assert p.returncode == 1 and b"failed import" in p.stderr
    p = subprocess.run([H.CLI_PATH, "import", "-i", vcf, "-o", out, "-n", "1.5"], capture_output=True)
    assert p.returncode == 1 and b"Cannot set missingness filter to > 1" in p.stderr


HWE_KAT = json.load(open(os.path.join(GOLD, "hwe_kat.json")))


def test_hwe_known_answers_from_reference():
    for hom1, het, hom2, want in HWE_KAT["cases"]:
        assert H.hwe_exact(hom1, het, hom2) == want, (hom1, het, hom2)


@pytest.mark.skipif(not O.have_ref(), reason="compiled reference (oracle/_ref) not available")
def test_reference_reader_accepts_imported_file(tmp_path):
    N, M = 64, 130
    al = util.random_alleles(M, N, 21, miss_rate=0.05, miss_variants=0.3)
    pos = np.arange(M) * 11 + 7
    vcf = str(tmp_path / "r.vcf.gz")
    write_vcf(vcf, al, pos, np.zeros(M, int), gz=True)
    out = str(tmp_path / "r.twk")
    cnt = H.import_vcf(vcf, out, threshold_miss=0.0, remove_univariate=False, block_size=50)
    assert cnt["written"] == M
    txt = subprocess.run([O.REF_BIN, "twkinfo", out], capture_output=True, text=True, check=True).stdout.splitlines()
    assert txt[0].startswith(f"#n_samples={N} n_contigs=2 n_blocks=3")
    blocks = [l.split("\t") for l in txt if l.startswith("#block")]
    # first block: 0-based minpos, later blocks 1-based (importer.cpp:262-264 vs core.cpp:221)
    assert [int(b[3]) for b in blocks] == [pos[0] - 1, pos[50], pos[100]] and [int(b[2]) for b in blocks] == [50, 50, 30]
    rows = np.array([[float(x) for x in l.split("\t")] for l in txt if not l.startswith("#")])
    assert np.array_equal(rows[:, 1], pos - 1) and np.array_equal(rows[:, 2], (al == 1).sum(axis=(1, 2)))
    assert np.array_equal(rows[:, 3], (al == 2).sum(axis=(1, 2)))
    _, _, _, meta, extra = read_back(out)
    assert np.array_equal(rows[:, 4], extra[:, 0]) and np.array_equal(rows[:, 5], extra[:, 1]) and np.array_equal(rows[:, 9], meta["hwe"])
    # the exact test itself, live
    for hom1, het, hom2 in [(17, 30, 9), (400, 90, 20), (3, 0, 2), (0, 25, 0)]:
        ref = float(subprocess.run([O.REF_BIN, "hwe", str(hom1), str(het), str(hom2)], capture_output=True, text=True).stdout)
        assert H.hwe_exact(hom1, het, hom2) == ref


def test_import_input_variants(tmp_path):
    """CRLF line ends, a contig the header does not declare, stdin input, a header-only file."""
    N = 10
    rng = np.random.default_rng(8)
    rows = []
    for v in range(12):
        g = rng.integers(0, 2, (N, 2))
        rows.append(f"{'chrA' if v < 6 else 'chrB'}\t{100 + 10 * v}\t.\tA\tC\t.\t.\t.\tGT\t" + "\t".join(f"{a}|{b}" for a, b in g))
    head = ["##fileformat=VCFv4.2", "##contig=<ID=chrA,length=1000>", '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
            "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"S{i}" for i in range(N))]
    vcf = str(tmp_path / "crlf.vcf")
    open(vcf, "wb").write(("\r\n".join(head + rows) + "\r\n").encode())
    out = str(tmp_path / "o.twk")
    cnt = H.import_vcf(vcf, out, remove_univariate=False, threshold_miss=0.0)
    assert cnt["written"] == 12 and cnt["sites"] == 12
    _, _, _, meta, _ = read_back(out)
    assert meta["rid"].tolist() == [0] * 6 + [1] * 6                         # chrB appended behind the declared contigs
    # the same through stdin
    p = subprocess.run([H.CLI_PATH, "import", "-i", "-", "-o", str(tmp_path / "stdin.twk"), "-r", "-n", "0"],
                       input=open(vcf, "rb").read(), capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    assert read_back(str(tmp_path / "stdin.twk"))[3]["pos"].tolist() == meta["pos"].tolist()
    # header only: a valid, empty .twk
    open(vcf, "w").write("\n".join(head) + "\n")
    cnt = H.import_vcf(vcf, out)
    assert cnt["sites"] == 0 and cnt["written"] == 0
    n, data, _, _, _ = read_back(out)
    assert n == N and len(data) == 0


# ---- BCF2 input ------------------------------------------------------------------------------------------------------
def _typed_int(v):
    import struct
    if -120 <= v <= 127:
        return bytes([0x11]) + struct.pack("<b", v)
    if -32000 <= v <= 32767:
        return bytes([0x12]) + struct.pack("<h", v)
    return bytes([0x13]) + struct.pack("<i", v)


def _typed_str(s):
    b = s.encode()
    return (bytes([len(b) << 4 | 7]) if len(b) < 15 else bytes([0xF7]) + _typed_int(len(b))) + b


def write_bcf(path, header_lines, sample_names, records, gz=True):
    """A BCF2.2 file (BCF2 spec, section 6) from scratch.  records: dicts with chrom (index into the ##contig lines), pos
    (1-based), ref, alts (list), fmt (list of (FORMAT id, per-sample lists of ints)) where GT values are given as
    (allele or None, phased) tuples per ploidy position, None for a shorter ploidy."""
    import struct
    text = "\n".join(header_lines) + "\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(sample_names) + "\n"
    ids = ["PASS"]
    for l in header_lines:
        if l.startswith(("##FILTER=<", "##INFO=<", "##FORMAT=<")):
            i = l[l.index("ID=") + 3:].split(",")[0].rstrip(">")
            if i not in ids:
                ids.append(i)
    out = b"BCF\x02\x02" + struct.pack("<I", len(text) + 1) + text.encode() + b"\0"
    n_s = len(sample_names)
    for r in records:
        shared = struct.pack("<iiifII", r["chrom"], r["pos"] - 1, len(r["ref"]), float("nan"),
                             (1 + len(r["alts"])) << 16 | 0, len(r.get("fmt", [])) << 24 | n_s)
        shared += _typed_str("rs1") + _typed_str(r["ref"]) + b"".join(_typed_str(a) for a in r["alts"]) + bytes([0x00])   # no FILTER
        indiv = b""
        for name, vals in r.get("fmt", []):
            indiv += _typed_int(ids.index(name))
            if name == "GT":
                ploidy = max(len(v) for v in vals)
                wide = r.get("gt_int16", False)
                indiv += bytes([ploidy << 4 | (2 if wide else 1)])
                for v in vals:
                    for k in range(ploidy):
                        if k >= len(v):
                            x = -32767 if wide else -127                 # vector end: this sample has fewer alleles
                        else:
                            a, ph = v[k]
                            x = ((a + 1) if a is not None else 0) << 1 | int(bool(ph))
                        indiv += struct.pack("<h" if wide else "<b", x)
            else:
                indiv += bytes([0x11]) + bytes(int(x) & 0xFF for x in vals)
        out += struct.pack("<II", len(shared), len(indiv)) + shared + indiv
    with (gzip.open(path, "wb") if gz else open(path, "wb")) as f:
        f.write(out)


@pytest.mark.parametrize("gz", [True, False])
def test_import_bcf_equals_import_vcf(tmp_path, gz):
    """`tomahawk import` reads BCF2 (the reference reads it through htslib): the same sites given as VCF text and as
    BCF produce the same .twk - genotypes, phase, counts, filters, site order."""
    N, M = 37, 60
    rng = np.random.default_rng(8)
    al = util.random_alleles(M, N, 8, miss_rate=0.1, miss_variants=0.3)
    pos = np.sort(rng.choice(np.arange(100, 5000), size=M, replace=False))
    chrom = (np.arange(M) >= 40).astype(int)
    pos[40:] = np.sort(pos[40:])
    phased = rng.random(M) < 0.7
    contigs = ("20", "21")
    vcf, bcf = str(tmp_path / "in.vcf"), str(tmp_path / ("in.bcf" if gz else "in.raw.bcf"))
    ref_alt = [(BASES[v % 4], BASES[(v + 1) % 4]) for v in range(M)]
    ref_alt[5] = ("A", "ACGTACGTACGTACGTACGT")           # not a SNP (and a string longer than 14: the long-length form)
    raw = [(10, "20\t%d\trsX\tA\tC,G\t.\tPASS\tAC=1\tGT\t" % (pos[10] - 1 if pos[10] - 1 not in pos else pos[10]) + "\t".join(["0|2"] * N)),    # multi-allelic
           (20, "20\t%d\trsY\tA\tC\t.\tPASS\tAC=1" % (pos[20] - 1 if pos[20] - 1 not in pos else pos[20]))]                                    # no FORMAT
    write_vcf(vcf, al, pos, chrom, phased=phased, contigs=contigs, ref_alt=ref_alt, extra_format=True, raw_lines=raw)
    header = [l for l in open(vcf).read().splitlines() if l.startswith("##")]
    # the same records, in file order, as BCF
    lines = [l for l in open(vcf).read().splitlines() if not l.startswith("#")]
    recs = []
    for l in lines:
        f = l.split("\t")
        r = dict(chrom=contigs.index(f[0]), pos=int(f[1]), ref=f[3], alts=f[4].split(","))
        if len(f) > 8:
            gts, dps = [], []
            for smp in f[9:]:
                g = smp.split(":")[0]
                sep = "|" if "|" in g else "/"
                parts = g.split(sep)
                gts.append([(None if a == "." else int(a), k > 0 and sep == "|") for k, a in enumerate(parts)])
                dps.append(17)
            r["fmt"] = [("GT", gts)] + ([("DP", dps)] if f[8] == "GT:DP" else [])
            r["gt_int16"] = (len(recs) % 3 == 0)
        recs.append(r)
    write_bcf(bcf, header, [f"S{i}" for i in range(N)], recs, gz=gz)
    out_v, out_b = str(tmp_path / "v.twk"), str(tmp_path / "b.twk")
    cv = H.import_vcf(vcf, out_v, threshold_miss=0.5, remove_univariate=False, block_size=16)
    cb = H.import_vcf(bcf, out_b, threshold_miss=0.5, remove_univariate=False, block_size=16)
    assert cv == cb and cv["written"] >= M - 3 and cv["not_biallelic"] == 1 and cv["not_snp"] == 1 and cv["no_format"] == 1
    a, b = read_back(out_v), read_back(out_b)
    assert a[0] == b[0] == N
    for x, y in zip(a[1:], b[1:]):
        assert np.array_equal(x, y)


def test_import_bcf_ploidy_and_gt_placement(tmp_path):
    """Haploid samples are padded with vector-end markers in BCF (mixed ploidy: dropped like in VCF); a FORMAT whose
    first key is not GT has no genotypes for the importer."""
    N = 8
    header = ["##fileformat=VCFv4.2", "##contig=<ID=1,length=1000>", '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="d">',
              '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">']
    dip = [[(i % 2, False), ((i // 2) % 2, True)] for i in range(N)]
    hap = [list(g) for g in dip]; hap[3] = hap[3][:1]
    recs = [dict(chrom=0, pos=10, ref="A", alts=["C"], fmt=[("GT", dip)]),
            dict(chrom=0, pos=20, ref="A", alts=["C"], fmt=[("GT", hap)]),                       # one haploid sample
            dict(chrom=0, pos=30, ref="A", alts=["C"], fmt=[("DP", [5] * N), ("GT", dip)]),     # GT not first
            dict(chrom=0, pos=40, ref="G", alts=["T"], fmt=[("GT", dip)])]
    bcf, out = str(tmp_path / "p.bcf"), str(tmp_path / "p.twk")
    write_bcf(bcf, header, [f"S{i}" for i in range(N)], recs)
    c = H.import_vcf(bcf, out, threshold_miss=0.0, remove_univariate=False)
    assert c["sites"] == 4 and c["written"] == 2 and c["mixed_ploidy"] == 1 and c["no_genotypes"] == 1
    n, data, mask, meta, extra = read_back(out)
    assert n == N and meta["pos"].tolist() == [9, 39]
    want = np.array([[g[0][0], g[1][0]] for g in dip], dtype=np.int8)[None]
    d, _ = O.bitvectors_from_alleles(np.repeat(want, 2, axis=0))
    assert np.array_equal(data, d) and extra[:, 2].tolist() == [1, 1]                           # phased: second allele's bit
