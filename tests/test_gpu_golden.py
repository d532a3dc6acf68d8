"""GPU vs the reference's own output (golden fixtures), through the C ABI and through the CLI."""
import glob
import os
import subprocess

import numpy as np
import pytest

import tomahawk_amd as T
from oracle import oracle as O
from tests import util
from tomahawk_amd import hostlib

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "n*.npz")))      # cli_* belong to test_gpu_cli_golden.py
MODES = {"p": T.MODE_PHASED, "u": T.MODE_UNPHASED, "d": T.MODE_AUTO}


def golden_as_oracle_records(mat):
    r = np.zeros(len(mat), dtype=O.RECORD_DTYPE)
    r["controller"] = mat[:, 0]; r["ridA"] = mat[:, 1]; r["Apos"] = mat[:, 2]; r["ridB"] = mat[:, 3]; r["Bpos"] = mat[:, 4]
    r["cnt"] = mat[:, 5:9]
    for i, f in enumerate(("D", "Dprime", "R", "R2", "P", "ChiSqFisher", "ChiSqModel")):
        r[f] = mat[:, 9 + i]
    return r


def _cases_with_records():
    return [(name, tag) for tag in ("p", "u", "d") for name in CASES if "rec_" + tag in np.load(os.path.join(GOLDEN, name + ".npz")).files]


@pytest.mark.parametrize("name,tag", _cases_with_records())
def test_hip_equals_reference_records(hip, name, tag):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    al = z["alleles"]
    variants = O.variants_from_alleles(al, pos=z["pos"], rid=z["rid"], phase=1)
    util.upload(hip, al, variants)
    # -p, missing genotypes, 2N not a multiple of 128: the reference's records carry PhasedVectorized's tail / padding
    # slips; the engine reproduces them on request (TWK_HIP_OPT_REF_COMPAT) and only then
    N = al.shape[1]
    compat = tag == "p" and bool((al == 2).any()) and (2 * N) % 128 != 0
    got, npairs, _ = hip.ld_all(MODES[tag], T.Filters(minR2=0.0), window=T.OPT_REF_COMPAT if compat else 0)
    M = al.shape[0]
    assert npairs == M * (M - 1) // 2
    want = golden_as_oracle_records(z["rec_" + tag])
    if compat:       # by default the engine returns the correct tables: equal to the oracle's, different from the reference's
        fixed, _, _ = hip.ld_all(MODES[tag], T.Filters(minR2=0.0))
        data, mask = O.bitvectors_from_alleles(al)
        util.assert_records_match(fixed, O.all_pairs(data, mask, variants, N, O.settings(minR2=0.0, phased=True), vector_only=False), variants)
        assert len(fixed) != len(got) or not np.array_equal(np.sort(fixed, order=["idxA", "idxB"])["cnt"], np.sort(got, order=["idxA", "idxB"])["cnt"])
    # includes which off-diagonal count sits in cnt[1] for pairs the reference ran through its run-length
    # kernel (missing data + low allele counts): the device mirrors that choice
    util.assert_records_match(got, want, variants)


@pytest.mark.parametrize("name,flag,tag", [("n64_small", "-p", "p"), ("n100_pad", "-u", "u"), ("n1000", "-p", "p"),
                                           ("n128_missing", None, "d")])
def test_cli_calc_end_to_end(tmp_path, name, flag, tag):
    """`tomahawk calc -i in.twk -o out.two -r 0`: .twk in, .two out, every record twice."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    al = z["alleles"]
    M = al.shape[0]
    n_contigs = int(z["rid"].max()) + 1
    twk = str(tmp_path / "in.twk")
    hostlib.write_twk(twk, al, z["pos"], z["rid"], phased=np.ones(M, np.uint8), n_contigs=n_contigs, block_size=23)
    out = str(tmp_path / "res")            # extension is forced to .two (ld.cpp:589-598)
    cmd = [hostlib.CLI_PATH, "calc", "-i", twk, "-o", out, "-r", "0", "-t", "2"] + ([flag] if flag else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "variants/s" in r.stderr and not r.stdout
    recs, info = hostlib.read_two(out + ".two")
    assert info["n_samples"] == al.shape[1] and info["n_contigs"] == n_contigs and info["state"] == 0
    want = z["rec_" + tag]
    assert len(recs) == 2 * len(want)
    m = hostlib.two_as_matrix(recs)
    keyA, keyB = m[:, 1] * 2**32 + m[:, 2], m[:, 3] * 2**32 + m[:, 4]
    fwd = m[keyA < keyB]
    rev = m[keyA > keyB]
    fwd = fwd[np.lexsort((fwd[:, 4], fwd[:, 3], fwd[:, 2], fwd[:, 1]))]
    rev = rev[np.lexsort((rev[:, 2], rev[:, 1], rev[:, 4], rev[:, 3]))]
    assert np.array_equal(fwd[:, 1:5], want[:, 1:5])
    assert np.array_equal(rev[:, [3, 4, 1, 2]], want[:, 1:5]) and np.array_equal(rev[:, 5:], fwd[:, 5:])
    variants = O.variants_from_alleles(al, pos=z["pos"], rid=z["rid"], phase=1)
    pos2idx = {(int(v["rid"]), int(v["pos"])): i for i, v in enumerate(variants)}
    got = np.zeros(len(fwd), dtype=T.RECORD_DTYPE)
    got["idxA"] = [pos2idx[(int(a), int(b))] for a, b in fwd[:, 1:3]]
    got["idxB"] = [pos2idx[(int(a), int(b))] for a, b in fwd[:, 3:5]]
    got["flags"] = fwd[:, 0]; got["cnt"] = fwd[:, 5:9]
    for i, f in enumerate(("D", "Dprime", "R", "R2", "P", "ChiSqFisher", "ChiSqModel")):
        got[f] = fwd[:, 9 + i]
    w = golden_as_oracle_records(want)
    util.remember_data(al, variants)
    util.assert_records_match(got, w, variants)
    lit = hostlib.header_literals(out + ".two")
    assert "##tomahawk_calcCommand=tomahawk calc -i" in lit and "##tomahawk_calcVersion=" in lit


def _fwd_records(mat, variants):
    """[n,16] .two matrix (forward copies) -> tomahawk_amd.RECORD_DTYPE keyed by variant index."""
    pos2idx = {(int(v["rid"]), int(v["pos"])): i for i, v in enumerate(variants)}
    got = np.zeros(len(mat), dtype=T.RECORD_DTYPE)
    got["idxA"] = [pos2idx[(int(a), int(b))] for a, b in mat[:, 1:3]]
    got["idxB"] = [pos2idx[(int(a), int(b))] for a, b in mat[:, 3:5]]
    got["flags"] = mat[:, 0]; got["cnt"] = mat[:, 5:9]
    for i, f in enumerate(("D", "Dprime", "R", "R2", "P", "ChiSqFisher", "ChiSqModel")):
        got[f] = mat[:, 9 + i]
    return got


@pytest.mark.parametrize("ival,flag,tag", [("2", "-p", "I_contig2"), ("1:2500-4300", "-u", "I_range")])
def test_cli_interval_slicing(tmp_path, ival, flag, tag):
    """calc -I: whole blocks overlapping the interval, like the reference (ld.cpp:257-368)."""
    z = np.load(os.path.join(GOLDEN, "n1000.npz"))
    al = z["alleles"]; M = al.shape[0]
    twk = str(tmp_path / "in.twk")
    hostlib.write_twk(twk, al, z["pos"], z["rid"], phased=np.ones(M, np.uint8), n_contigs=2, block_size=50)
    out = str(tmp_path / "o.two")
    r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", twk, "-o", out, "-r", "0", flag, "-I", ival], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    m = hostlib.two_as_matrix(hostlib.read_two(out)[0])
    keyA, keyB = m[:, 1] * 2**32 + m[:, 2], m[:, 3] * 2**32 + m[:, 4]
    fwd = m[keyA < keyB]
    variants = O.variants_from_alleles(al, pos=z["pos"], rid=z["rid"], phase=1)
    util.remember_data(al, variants)
    util.assert_records_match(_fwd_records(fwd, variants), golden_as_oracle_records(z["rec_" + tag]), variants)
    r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", twk, "-o", out, "-I", "7"], capture_output=True, text=True)
    assert r.returncode == 1 and "Contig does not exist" in r.stderr


def test_cli_scalc_single_site(tmp_path):
    """`tomahawk scalc -I chr:pos -w W`: target x neighbours, both copies of every record."""
    z = np.load(os.path.join(GOLDEN, "n64_scalc.npz"))
    al = z["alleles"]; M = al.shape[0]
    twk = str(tmp_path / "in.twk")
    hostlib.write_twk(twk, al, z["pos"], z["rid"], phased=np.ones(M, np.uint8), n_contigs=1, block_size=64)
    out = str(tmp_path / "s.two")
    r = subprocess.run([hostlib.CLI_PATH, "scalc", "-i", twk, "-o", out, "-I", "1:16001", "-w", "5000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    m = hostlib.two_as_matrix(hostlib.read_two(out)[0])
    want = z["rec_scalc"]
    assert len(m) == len(want)
    tgt_pos = z["pos"][150]
    variants = O.variants_from_alleles(al, pos=z["pos"], rid=z["rid"], phase=1)
    g = _fwd_records(m[m[:, 2] == tgt_pos], variants)            # forward copies: the target is A, also for neighbours before it
    w = golden_as_oracle_records(want[want[:, 2] == tgt_pos])
    util.remember_data(al, variants)
    util.assert_records_match(g, w, variants)
    # more neighbours than a multiple of 100: the reference drops the remainder (ld.cpp:203-205,239-244), we keep it
    r = subprocess.run([hostlib.CLI_PATH, "scalc", "-i", twk, "-o", out, "-I", "1:16001", "-w", "6000"], capture_output=True, text=True)
    assert r.returncode == 0 and len(hostlib.read_two(out)[0]) > len(want)
    r = subprocess.run([hostlib.CLI_PATH, "scalc", "-i", twk, "-o", out, "-I", "1:999999"], capture_output=True, text=True)
    assert r.returncode == 1


def test_cli_chunked_equals_whole(tmp_path):
    """-c 3 -C k (reference farm mode): the three parts partition the whole output."""
    N, M = 96, 240
    al = util.random_alleles(M, N, 5)
    pos = (1000 + 10 * np.arange(M)).astype(np.uint32)
    twk = str(tmp_path / "in.twk")
    hostlib.write_twk(twk, al, pos, np.zeros(M, np.uint32), np.ones(M, np.uint8), block_size=20)   # 12 blocks
    def run(extra, out):
        r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", twk, "-o", out, "-r", "0.05", "-p"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return hostlib.two_as_matrix(hostlib.read_two(out)[0])
    whole = run([], str(tmp_path / "w.two"))
    parts = [run(["-c", "3", "-C", str(k)], str(tmp_path / f"p{k}.two")) for k in (1, 2, 3)]
    cat = np.concatenate(parts)
    key = lambda m: m[np.lexsort((m[:, 4], m[:, 2]))]
    assert len(cat) == len(whole) > 0 and np.array_equal(key(cat), key(whole))


def test_cli_window_mode(tmp_path):
    N, M = 64, 200
    al = util.random_alleles(M, N, 9)
    pos = (1000 + 1000 * np.arange(M)).astype(np.uint32)
    twk = str(tmp_path / "in.twk")
    hostlib.write_twk(twk, al, pos, np.zeros(M, np.uint32), np.ones(M, np.uint8), block_size=16)
    def run(extra, out):
        r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", twk, "-o", out, "-r", "0", "-u"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return hostlib.two_as_matrix(hostlib.read_two(out)[0])
    whole = run([], str(tmp_path / "w.two"))
    win = run(["-w", "20000"], str(tmp_path / "win.two"))
    sel = whole[np.abs(whole[:, 2] - whole[:, 4]) <= 20000]
    key = lambda m: m[np.lexsort((m[:, 4], m[:, 2]))]
    assert 0 < len(win) < len(whole) and np.array_equal(key(win), key(sel))


def _env_and_engine_options(extra):
    """Keys "opt:<key>" are engine switches and go on the command line (--engine-option key=value: the libraries read
    no switch from the environment); the rest is environment (TWK_HIP_GPUS / TWK_HIP_PART / TWK_REF_COMPAT)."""
    opts = [a for k, v in extra.items() if k.startswith("opt:") for a in ("--engine-option", f"{k[4:]}={v}")]
    return dict(os.environ, **{k: v for k, v in extra.items() if not k.startswith("opt:")}), opts


def test_cli_multi_gpu_driver_threads_equal_single(tmp_path):
    """TWK_HIP_GPUS=n: one driver thread and one engine context per GPU (here all on GPU 0: --engine-option force_device=0), one shared
    writer; TWK_HIP_PART=k/n: this process's share of a multi-node run, merged with concat."""
    N, M = 80, 500
    al = util.random_alleles(M, N, 77, miss_rate=0.05, miss_variants=0.2)
    pos = (1000 + 10 * np.arange(M)).astype(np.uint32)
    twk = str(tmp_path / "in.twk")
    hostlib.write_twk(twk, al, pos, np.zeros(M, np.uint32), np.ones(M, np.uint8), block_size=50)
    def run(env_extra, out, extra=()):
        env, opts = _env_and_engine_options(env_extra)
        r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", twk, "-o", out, "-r", "0.02"] + opts + list(extra), capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        return hostlib.two_as_matrix(hostlib.read_two(out)[0]), r.stderr
    whole, log0 = run({"opt:progress_ms": "0"}, str(tmp_path / "w.two"))
    prog = [l for l in log0.splitlines() if "[PROGRESS]" in l]
    assert "Time elapsed" in prog[0] and "Est. Time left" in prog[0] and "%" in prog[1]          # ticker lines (ld_progress.h:48-75)
    key = lambda m: m[np.lexsort((m[:, 4], m[:, 2]))]
    assert len(whole) > 0
    for mode in ((), ("-u",), ("-p",)):
        one, _ = run({}, str(tmp_path / "one.two"), mode)
        # three contexts; every one is pinned to device 0 because the box has one GPU
        multi, log = run({"TWK_HIP_GPUS": "3", "opt:force_device": "0", "opt:progress_ms": "0"}, str(tmp_path / "m.two"), mode)
        assert len(one) > 0 and np.array_equal(key(one), key(multi))
        assert "Using 3 GPUs" in log and "GPU 2: count kernel" in log
    # farm mode: two processes x two GPUs each = four shards, outputs merged with concat
    parts = []
    for k in range(2):
        out = str(tmp_path / f"farm{k}.two")
        run({"TWK_HIP_GPUS": "2", "opt:force_device": "0", "TWK_HIP_PART": f"{k}/2"}, out)
        parts.append(out)
    cat = str(tmp_path / "cat.two")
    r = subprocess.run([hostlib.CLI_PATH, "concat", "-i", parts[0], "-i", parts[1], "-o", cat], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(key(whole), key(hostlib.two_as_matrix(hostlib.read_two(cat)[0])))
    # more GPUs than the box has is an error, not a silent fallback
    r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", twk, "-o", str(tmp_path / "x.two")], capture_output=True, text=True,
                       env=dict(os.environ, TWK_HIP_GPUS="64"))
    assert r.returncode != 0 and "device(s) are visible" in r.stderr


def test_cli_window_mode_slabs_per_gpu(tmp_path):
    """`calc -w` on several GPUs: every GPU loads only the blocks of its band of rows plus the halo its window reaches
    (what makes configs[4] - 500 GB of bitvectors - fit), and the union of the bands is the single-GPU output.  Several
    contigs with positions that start over, missing genotypes, every mode; also as two processes (TWK_HIP_PART)."""
    N = 90
    sizes = [700, 500, 40, 360]
    M = sum(sizes)
    al = util.mosaic_alleles(M, N, 91, n_founders=5, switch=0.02, mut=0.01, miss_rate=0.05, miss_variants=0.25)
    rid = np.repeat(np.arange(len(sizes)), sizes).astype(np.uint32)
    pos = np.concatenate([1000 + 50 * np.arange(n) for n in sizes]).astype(np.uint32)
    twk = str(tmp_path / "in.twk")
    hostlib.write_twk(twk, al, pos, rid, np.ones(M, np.uint8), n_contigs=len(sizes), block_size=37)
    key = lambda m: m[np.lexsort((m[:, 4], m[:, 3], m[:, 2], m[:, 1]))]
    def run(env_extra, out, extra):
        env, opts = _env_and_engine_options(env_extra)
        r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", twk, "-o", out, "-r", "0.05", "-w", "2500"] + opts + list(extra), capture_output=True,
                           text=True, env=env)
        assert r.returncode == 0, r.stderr
        return key(hostlib.two_as_matrix(hostlib.read_two(out)[0])), r.stderr
    for mode in ((), ("-u",), ("-p",)):
        one, _ = run({}, str(tmp_path / "one.two"), mode)
        multi, log = run({"TWK_HIP_GPUS": "4", "opt:force_device": "0"}, str(tmp_path / "m.two"), mode)
        assert len(one) > 2000 and np.array_equal(one, multi)
        halos = [int(l.split("+ ")[1].split(" halo")[0]) for l in log.splitlines() if "halo variants" in l]
        assert len(halos) == 4 and max(halos) <= 3 * 37 + 50 and "GPU 3: rows = variants" in log        # a window of 50 variants: <= 2-3 blocks of halo
    parts = []
    for k in range(2):
        out = str(tmp_path / f"farm{k}.two")
        run({"TWK_HIP_GPUS": "2", "opt:force_device": "0", "TWK_HIP_PART": f"{k}/2"}, out, ())
        parts.append(out)
    cat = str(tmp_path / "cat.two")
    r = subprocess.run([hostlib.CLI_PATH, "concat", "-i", parts[0], "-i", parts[1], "-o", cat], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    whole, _ = run({}, str(tmp_path / "w.two"), ())
    assert np.array_equal(whole, key(hostlib.two_as_matrix(hostlib.read_two(cat)[0])))
    # TWK_REF_COMPAT window filter composes with the slabs (forced mode) and switches them off where it must (default mode)
    for mode in (("-u",), ()):
        one, _ = run({"TWK_REF_COMPAT": "1"}, str(tmp_path / "c1.two"), mode)
        multi, _ = run({"TWK_REF_COMPAT": "1", "TWK_HIP_GPUS": "3", "opt:force_device": "0"}, str(tmp_path / "c3.two"), mode)
        assert len(one) > 0 and np.array_equal(one, multi)


def test_cli_stdout_output_is_the_same_file(tmp_path):
    """`-o -` (ld.cpp:585-587): the .two container goes to stdout, logging stays on stderr; piped into a file it is the
    file `-o path` writes (one GPU, survivors in pair order: deterministic), and `view` reads it."""
    z = np.load(os.path.join(GOLDEN, "n1000.npz"))
    al = z["alleles"]
    M = al.shape[0]
    twk = str(tmp_path / "in.twk")
    hostlib.write_twk(twk, al, z["pos"], z["rid"], phased=np.ones(M, np.uint8), n_contigs=int(z["rid"].max()) + 1, block_size=23)
    a, b = str(tmp_path / "a.two"), str(tmp_path / "b.two")
    r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", twk, "-o", a, "-r", "0", "-p"], capture_output=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", twk, "-o", "-", "-r", "0", "-p"], capture_output=True)
    assert r.returncode == 0 and b"variants/s" in r.stderr
    open(b, "wb").write(r.stdout)
    ra, ia = hostlib.read_two(a)
    rb, ib = hostlib.read_two(b)
    assert ia == ib and ra.tobytes() == rb.tobytes() and len(ra) == 2 * len(z["rec_p"])
    state, ent, _ = hostlib.two_index(b)
    assert state == 0 and ent[:, 2].sum() == len(rb)


def test_cli_corrupt_input_fails_cleanly(tmp_path):
    """Damaged .twk blocks - bad zstd frames, record headers that lie, run lengths that do not add up to the sample count
    (found by the device inflate kernel) - end the run with an error message and exit code 1; nothing crashes or hangs."""
    N, M = 120, 400
    al = util.random_alleles(M, N, 3, miss_rate=0.05, miss_variants=0.2)
    twk = str(tmp_path / "ok.twk")
    hostlib.write_twk(twk, al, (1000 + 10 * np.arange(M)).astype(np.uint32), np.zeros(M, np.uint32), np.ones(M, np.uint8), block_size=64)
    good = open(twk, "rb").read()
    r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", twk, "-o", str(tmp_path / "ok.two")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    rng = np.random.default_rng(1)
    outcomes = set()
    for trial in range(12):
        bad = bytearray(good)
        lo = len(good) // 8 + int(rng.integers(0, len(good) // 2))
        if trial % 3 == 0:
            for k in range(lo, lo + 40): bad[k] = int(rng.integers(0, 256))          # scribble inside a compressed block
        elif trial % 3 == 1:
            bad[lo] ^= 1 << int(rng.integers(0, 8))                                     # one flipped bit
        else:
            del bad[lo:lo + int(rng.integers(1, 200))]                                  # bytes missing: every offset after it is off
        p = str(tmp_path / f"bad{trial}.twk")
        open(p, "wb").write(bytes(bad))
        r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", p, "-o", str(tmp_path / "bad.two")], capture_output=True, text=True, timeout=120)
        assert r.returncode in (0, 1), (trial, r.returncode, r.stderr[-300:])
        outcomes.add(r.returncode)
        if r.returncode == 1:
            assert "ERROR" in r.stderr or "failed" in r.stderr.lower()
    assert 1 in outcomes


def test_cli_full_chain_import_calc_sort_view(tmp_path):
    """VCF -> import -> calc (default mode, missing genotypes) -> sort -> view: the chain a user of the reference runs."""
    from tests.test_import import write_vcf
    N, M = 96, 300
    al = util.random_alleles(M, N, 5, miss_rate=0.04, miss_variants=0.2)
    pos = 500 + 25 * np.arange(M)
    vcf, twk, direct = str(tmp_path / "in.vcf.gz"), str(tmp_path / "in.twk"), str(tmp_path / "direct.twk")
    write_vcf(vcf, al, pos, np.zeros(M, int), gz=True)
    cnt = hostlib.import_vcf(vcf, twk, threshold_miss=0.5, remove_univariate=False, block_size=64)
    assert cnt["written"] == M
    hostlib.write_twk(direct, al, pos - 1, np.zeros(M, np.uint32), np.ones(M, np.uint8), n_contigs=2, block_size=64)
    outs = []
    for src in (twk, direct):
        out = str(tmp_path / (os.path.basename(src) + ".two"))
        r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", src, "-o", out, "-r", "0.05"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs.append(out)
    a, b = (hostlib.read_two(o)[0] for o in outs)
    assert len(a) > 0 and a.tobytes() == b.tobytes()           # imported file == directly written genotypes, record for record
    srt = str(tmp_path / "sorted.two")
    r = subprocess.run([hostlib.CLI_PATH, "sort", "-i", outs[0], "-o", srt], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([hostlib.CLI_PATH, "view", "-i", srt, "-H", "-I", "20:1000-3000", "-u"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [l.split("\t") for l in r.stdout.splitlines()[1:]]
    sel = a[(a["packA"] >> 2 >= 1000) & (a["packA"] >> 2 <= 3000) & ((a["packA"] >> 2) < (a["packB"] >> 2))]
    assert len(rows) == len(sel) > 0 and all(row[1] == "20" and 1001 <= int(row[2]) <= 3001 for row in rows)


@pytest.mark.gpu
def test_cli_hand_off_queue_writes_the_same_file(tmp_path):
    """Between the engine's thread and the record emitter `tomahawk calc` keeps a queue of copied pieces (engine option
    emit_queue_pieces, 8 buffers of 2^20 survivors by default; 0: the engine's thread feeds the emitter itself).  2.4 M surviving
    pairs - three pieces - with no queue, a queue of one buffer (the sink waits for it every time) and the default: one
    GPU's output is deterministic, so the three files hold the same records in the same blocks."""
    import hashlib
    N, M = 64, 2200
    al = util.random_alleles(M, N, 5)
    twk = str(tmp_path / "in.twk")
    hostlib.write_twk(twk, al, (1000 + 10 * np.arange(M)).astype(np.uint32), np.zeros(M, np.uint32), np.ones(M, np.uint8), block_size=100)
    digests = {}
    for pieces in (0, 1, 8):
        out = str(tmp_path / f"q{pieces}.two")
        r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", twk, "-o", out, "-p", "-r", "0", "-P", "1", "--engine-option", f"emit_queue_pieces={pieces}"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "output: " in r.stderr
        recs, info = hostlib.read_two(out)           # (the files themselves differ in the command line their headers quote)
        digests[pieces] = (hashlib.sha256(recs.tobytes()).hexdigest(), info["n_blocks"])
        assert len(recs) > 2 * (1 << 20)             # (both copies of) more than one full piece
    assert digests[0] == digests[1] == digests[8]


@pytest.mark.gpu
def test_cli_record_codec_writes_the_same_records_and_the_reference_reads_them(tmp_path):
    """The default at `-k 1` (round 6; `--engine-option record_codec=0` for libzstd): the output blocks' zstd frames come from the records' own encoder
    (csrc/host/twk_repcodec.h) or, where its 32 KiB sample comes out a quarter larger than libzstd's, from libzstd as before.
    Two inputs: 1,200 samples (noisy statistics: the encoder's kind of block - the file must differ from the default's and stay
    within 15 % of it) and 64 samples (few distinct values: left to libzstd, same size).  Either way the records, the blocks and
    the index are the default run's, and the compiled reference's `view` prints the same lines from both files."""
    import hashlib
    for N, M, want_codec in ((1200, 900, True), (64, 1500, False)):
        al = util.mosaic_alleles(M, N, 8, n_founders=10, switch=0.05, mut=0.01) if want_codec else util.random_alleles(M, N, 5)
        twk = str(tmp_path / f"in{N}.twk")
        hostlib.write_twk(twk, al, (1000 + 10 * np.arange(M)).astype(np.uint32), np.zeros(M, np.uint32), np.ones(M, np.uint8), block_size=100)
        got = {}
        for codec in (0, 1, "default", "k3"):
            out = str(tmp_path / f"c{N}_{codec}.two")
            extra = ["--engine-option", f"record_codec={codec}"] if codec in (0, 1) else ["-k", "3"] if codec == "k3" else []
            r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", twk, "-o", out, "-p", "-r", "0", "-P", "1"] + extra, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            recs, info = hostlib.read_two(out)
            state, ent, _ = hostlib.two_index(out)
            got[codec] = (hashlib.sha256(recs.tobytes()).hexdigest(), info["n_blocks"], ent[:, :5].tolist(), os.path.getsize(out), out)
            assert len(recs) > 300_000
        assert got[0][:3] == got[1][:3] == got["default"][:3] == got["k3"][:3]
        # the default (-k 1, the reference's) takes the records' encoder; -k 3 asks for ratio and gets libzstd at that level
        assert abs(got["default"][3] - got[1][3]) < 200
        if want_codec:
            assert got["k3"][3] < got[1][3] and got[0][3] < got[1][3]
        if want_codec:
            assert got[0][3] != got[1][3] and got[1][3] < 1.15 * got[0][3]
        else:
            assert abs(got[0][3] - got[1][3]) < 200          # (the headers quote different command lines)
        if O.have_ref():
            lines = [[l for l in O.run_ref(["view", "-i", g[4]]).stdout.splitlines() if not l.startswith("#")] for g in (got[0], got[1])]
            assert lines[0] == lines[1] and len(lines[0]) > 300_000
