"""Host-side formats and the C-ABI surface (CPU only, no compute calls)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import tomahawk_amd as T
from oracle import oracle as O
from tests import util
from tomahawk_amd import hostlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "twk_hip.h")).read()
    names = sorted(set(re.findall(r"\b(twk_hip_[a-z_]+|twk_synth_[a-z_]+)\s*\(", hdr)))
    names = [n for n in names if n != "twk_hip_record_sink"]
    assert len(names) >= 17
    lib = T.load_library()
    for n in names:
        assert hasattr(lib, n), f"libtwk_hip.so does not export {n}"
    assert lib.twk_hip_abi_version() == 5
    assert lib.twk_hip_strerror(-4).decode() == "record buffer too small"


def test_struct_layouts_match_header():
    assert T.RECORD_DTYPE.itemsize == 104 and hostlib.TWO_DTYPE.itemsize == 106
    assert T.hip.META_DTYPE.itemsize == 32 and ctypes.sizeof(T.hip._Tile) == 32 and ctypes.sizeof(T.hip._Filters) == 40


def test_no_device_fails_loudly_without_fallback():
    if T.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(T.HipError) as e:
        T.HipLd(0)
    assert e.value.code == -3


def test_read_reference_written_two():
    """Our .two reader on a file written by the reference itself."""
    recs, info = hostlib.read_two(os.path.join(GOLDEN, "ref_n64_small_p.two"))
    z = np.load(os.path.join(GOLDEN, "n64_small.npz"))
    want = z["rec_p"]
    assert info["n_samples"] == 64 and info["n_contigs"] == 1 and info["state"] == 0
    assert len(recs) == 2 * len(want)
    m = hostlib.two_as_matrix(recs)
    fwd = m[m[:, 2] < m[:, 4]]
    fwd = fwd[np.lexsort((fwd[:, 4], fwd[:, 2]))]
    assert np.array_equal(fwd, want)
    rev = m[m[:, 2] > m[:, 4]]
    rev = rev[np.lexsort((rev[:, 2], rev[:, 4]))]
    # reverse copy: (rid,pos) swapped, everything else (including cnt) untouched
    assert np.array_equal(rev[:, [4, 2]], want[:, [2, 4]]) and np.array_equal(rev[:, 5:], want[:, 5:])
    assert "##tomahawk_calcCommand=" in hostlib.header_literals(os.path.join(GOLDEN, "ref_n64_small_p.two"))


@pytest.mark.parametrize("N,M,kw", [(64, 30, {}), (100, 45, dict(low_ac=5)), (257, 20, dict(miss_rate=0.2, miss_variants=0.5)),
                                    (5000, 6, dict(maf_lo=0.0005, maf_hi=0.002))])
def test_twk_roundtrip(tmp_path, N, M, kw):
    al = util.random_alleles(M, N, 77, **kw)
    pos = (1000 + 50 * np.arange(M)).astype(np.uint32)
    rid = (np.arange(M) * 2 // M).astype(np.uint32)
    p = str(tmp_path / "x.twk")
    hostlib.write_twk(p, al, pos, rid, phased=np.arange(M) % 2, n_contigs=2, block_size=7,
                      hwe=np.linspace(0, 1, M))
    n, data, mask, meta, extra = hostlib.read_twk(p)
    want_data, want_mask = O.bitvectors_from_alleles(al)
    assert n == N and np.array_equal(data, want_data)
    assert np.array_equal(mask, want_mask if want_mask is not None else np.zeros_like(data))
    assert np.array_equal(meta["pos"], pos) and np.array_equal(meta["rid"], rid)
    assert np.array_equal(meta["ac"], (al == 1).sum(axis=(1, 2))) and np.array_equal(meta["an"], (al == 2).sum(axis=(1, 2)))
    assert np.array_equal(meta["missing"], (al == 2).any(axis=(1, 2)))
    np.testing.assert_array_equal(meta["hwe"], np.linspace(0, 1, M))
    assert np.array_equal(extra[:, 2], np.arange(M) % 2)


def test_truncated_and_foreign_files_are_rejected(tmp_path):
    al = util.random_alleles(10, 64, 1)
    p = str(tmp_path / "x.twk")
    hostlib.write_twk(p, al, np.arange(10, dtype=np.uint32), np.zeros(10, np.uint32), np.ones(10, np.uint8))
    raw = open(p, "rb").read()
    open(p, "wb").write(raw[: len(raw) - 20])
    with pytest.raises(RuntimeError):
        hostlib.read_twk(p)
    open(p, "wb").write(b"NOTATWKFILE" + raw[11:])
    with pytest.raises(RuntimeError):
        hostlib.read_twk(p)
    with pytest.raises(RuntimeError):
        hostlib.read_two(os.path.join(GOLDEN, "n64_small.npz"))


def test_synthetic_twk_matches_host_generator(tmp_path):
    N, M = 300, 12
    p = str(tmp_path / "s.twk")
    hostlib.write_synthetic_twk(p, N, M, seed=42, phased=False, block_size=5, n_threads=3)
    n, data, mask, meta, extra = hostlib.read_twk(p)
    assert n == N and not mask.any()
    for v in range(M):
        bv, ac = T.synth_bitvector(42, N, v)
        assert np.array_equal(data[v], bv) and meta["ac"][v] == ac and meta["pos"][v] == 1000 + 100 * v
    # determinism + seed sensitivity
    assert np.array_equal(T.synth_bitvector(42, N, 3)[0], T.synth_bitvector(42, N, 3)[0])
    assert not np.array_equal(T.synth_bitvector(42, N, 3)[0], T.synth_bitvector(43, N, 3)[0])


def test_cli_usage_and_errors():
    import subprocess
    r = subprocess.run([hostlib.CLI_PATH], capture_output=True, text=True)
    assert r.returncode == 1 and "tomahawk calc" in r.stderr
    r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", "/nonexistent.twk", "-o", "/tmp/x", "-r", "2"], capture_output=True, text=True)
    assert r.returncode == 1 and "Cannot have minimum R-squared value > 1" in r.stderr
    r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", "/nonexistent.twk", "-o", "/tmp/x"], capture_output=True, text=True)
    assert r.returncode == 1 and "Failed to open file" in r.stderr
    r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", "a", "-o", "b", "-a", "3"], capture_output=True, text=True)
    assert r.returncode == 1 and "Unrecognized option" in r.stderr          # reference: calc.h:216-218
    r = subprocess.run([hostlib.CLI_PATH, "scalc", "-i", "/nonexistent.twk", "-o", "/tmp/x", "-w", "0"], capture_output=True, text=True)
    assert r.returncode == 1 and "non-positive window" in r.stderr
    r = subprocess.run([hostlib.CLI_PATH, "view"], capture_output=True, text=True)
    assert r.returncode == 0 and "Usage:  tomahawk view" in r.stderr            # usage, like view.h:63-66
    r = subprocess.run([hostlib.CLI_PATH, "decay"], capture_output=True, text=True)
    assert r.returncode == 1 and "Illegal command" in r.stderr


def test_concat_copies_blocks_and_rebases_index(tmp_path):
    """`tomahawk concat` (lib/concat.h): two copies of a reference-written .two -> every record twice, valid index."""
    import subprocess
    src = os.path.join(GOLDEN, "ref_n64_small_p.two")
    out = str(tmp_path / "cat")
    r = subprocess.run([hostlib.CLI_PATH, "concat", "-i", src, "-i", src, "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    one, info1 = hostlib.read_two(src)
    two, info2 = hostlib.read_two(out + ".two")           # read_two also checks index vs block stream
    assert len(two) == 2 * len(one) and info2["n_blocks"] == 2 * info1["n_blocks"]
    assert two[: len(one)].tobytes() == one.tobytes() and two[len(one):].tobytes() == one.tobytes()
    assert "##tomahawk_concatCommand=tomahawk concat" in hostlib.header_literals(out + ".two")
    if O.have_ref():                                       # the reference reads what we wrote
        d = O.run_ref(["dump", out + ".two"]).stdout.splitlines()
        assert len([l for l in d if not l.startswith("#")]) == len(two)
    lst = tmp_path / "files.txt"
    lst.write_text(src + "\n" + src + "\n" + src + "\n")
    r = subprocess.run([hostlib.CLI_PATH, "concat", "-I", str(lst), "-o", str(tmp_path / "c3.two")], capture_output=True, text=True)
    assert r.returncode == 0 and len(hostlib.read_two(str(tmp_path / "c3.two"))[0]) == 3 * len(one)
    r = subprocess.run([hostlib.CLI_PATH, "concat", "-i", src, "-o", out], capture_output=True, text=True)
    assert r.returncode == 1 and "Only one input file" in r.stderr


def test_reference_client_source_builds_against_the_shim(tmp_path):
    """A libtomahawk client written against the reference (`#include "ld.h"`, tomahawk::twk_ld, -ltomahawk; lib/calc.h:96,
    237-238) compiles and links unchanged: include/ld.h is a shim of the reference header's name, lib/libtomahawk.so the
    engine's library under the reference library's name.  (Running it needs a GPU: without one Compute() fails loudly.)"""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "tomahawk_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libtomahawk.so")):
        os.symlink("libtomahawk_amd.so", os.path.join(lib, "libtomahawk.so"))       # (make host creates it)
    src = tmp_path / "client.cpp"
    src.write_text('#include "ld.h"\n'
                   'namespace tomahawk { std::string LITERAL_COMMAND_LINE = "client"; }\n'
                   'int main(int argc, char** argv) {\n'
                   '    tomahawk::twk_ld_settings settings;\n'
                   '    settings.in = argc > 1 ? argv[1] : ""; settings.out = "out.two"; settings.force_phased = true;\n'
                   '    tomahawk::twk_ld ld;\n'
                   '    if (ld.Compute(settings) == false) return 1;\n'
                   '    return 0;\n}\n')
    exe = str(tmp_path / "client")
    r = subprocess.run([shutil.which("g++") or "g++", "-std=c++11", "-I" + os.path.join(root, "include"), str(src), "-o", exe,
                        "-L" + lib, "-ltomahawk", "-ltwk_hip", "-Wl,-rpath," + lib], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "No file-name provided" in r.stderr


def test_reference_client_that_touches_record_types_builds_against_the_shim(tmp_path):
    """`#include "ld.h"` also declares the output record and its block (reference include/core.h:756-834, 851-902, reached
    through ld.h:30-31): a client that fills twk1_two_t records, flags them, collects them in a twk1_two_block_t, sorts it
    and serialises it compiles against include/ alone; the 106-byte records it packs are the .two record of
    hostlib.TWO_DTYPE (lib/core.cpp:470-490), in the order of `tomahawk sort` (core.cpp:458-468), and read back equal."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "client.cpp"
    src.write_text(r'''#include "ld.h"
#include <cstdio>
#include <vector>
static_assert(tomahawk::twk1_two_t::packed_size == 106, "packed size");
int main() {
    tomahawk::twk1_two_block_t blk;
    const unsigned pos[5][4] = {{1, 0, 900, 50}, {0, 1, 7, 8}, {0, 0, 500, 20}, {0, 0, 500, 10}, {0, 0, 3, 1073741823u}};
    for (int i = 0; i < 5; ++i) {
        tomahawk::twk1_two_t r;
        r.ridA = pos[i][0]; r.ridB = pos[i][1]; r.Apos = pos[i][2]; r.Bpos = pos[i][3];
        r.Aphased = 1; r.Bmiss = i & 1;
        r[0] = 10 + i; r[3] = 0.5 * i; r.R = -0.25 * i; r.R2 = r.R * r.R; r.D = 1e-3 * i; r.Dprime = 1; r.P = 1e-300;
        r.ChiSqModel = 3; r.ChiSqFisher = 4;
        r.SetUsedPhasedMath(); r.SetSameContig(r.ridA == r.ridB); r.SetInvalidHWEB(i == 2);
        blk += r;
    }
    if (blk.size() != 5 || blk.m != 500) return 2;
    blk.Sort();
    std::vector<unsigned char> bytes(blk.packed_bytes());
    blk.pack(bytes.data());
    tomahawk::twk1_two_block_t back;
    if (!back.unpack(bytes.data(), bytes.size()) || back.size() != 5) return 3;
    for (unsigned i = 0; i < 5; ++i) if (back[i] < blk[i] || blk[i] < back[i] || back[i].P != blk[i].P || back[i].controller != blk[i].controller) return 4;
    if (back.unpack(bytes.data(), bytes.size() - 1)) return 5;
    fwrite(bytes.data(), 1, bytes.size(), stdout);
    return 0;
}
''')
    exe = str(tmp_path / "client")
    r = subprocess.run([shutil.which("g++") or "g++", "-std=c++11", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), str(src), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True)
    assert r.returncode == 0
    n, m = np.frombuffer(r.stdout[:8], dtype="<u4")
    assert (n, m) == (5, 500) and len(r.stdout) == 8 + 5 * 106
    recs = np.frombuffer(r.stdout[8:], dtype=hostlib.TWO_DTYPE)
    assert [(int(x["ridA"]), int(x["ridB"]), int(x["packA"]) >> 2, int(x["packB"]) >> 2) for x in recs] == \
        [(0, 0, 3, 1073741823), (0, 0, 500, 10), (0, 0, 500, 20), (0, 1, 7, 8), (1, 0, 900, 50)]
    assert all(int(x["packA"]) & 3 == 2 for x in recs) and [int(x["packB"]) & 3 for x in recs] == [0, 1, 0, 1, 0]
    assert [int(x["controller"]) for x in recs] == [3, 3, 3 | 8192, 1, 1]
    assert recs["cnt"][:, 0].tolist() == [14, 13, 12, 11, 10] and recs["P"].tolist() == [1e-300] * 5
    assert recs["ChiSqFisher"].tolist() == [4] * 5 and recs["ChiSqModel"].tolist() == [3] * 5 and recs["Dprime"].tolist() == [1] * 5


@pytest.mark.parametrize("n_threads,b_size", [(1, 50), (3, 128), (8, 1000)])
def test_record_stream_blocks_follow_the_flush_rule(tmp_path, n_threads, b_size):
    """hostlib.TwoStream (RecordEmitter: worker threads expand + compress, a writer thread appends in order, at most
    6 x threads blocks in flight): many appends of ragged sizes, sorted and unsorted, on three contigs.  Every record
    comes out twice (forward, then reverse with rid / pos swapped), in (idxA, idxB) order within an append; a block
    holds one (ridA, ridB) pair and at most b_size records, and only a block followed by a contig change (or the
    last one) is short (ld_engine.cpp:1270-1281)."""
    M = 400
    rng = np.random.default_rng(5)
    rid = np.repeat([0, 1, 2], [150, 150, 100]).astype(np.uint32)
    pos = (1000 + 10 * np.arange(M)).astype(np.uint32)
    path = str(tmp_path / "s.two")
    st = hostlib.TwoStream(path, 10, rid, pos, n_contigs=3, b_size=b_size, n_threads=n_threads)
    want = []
    for k, n in enumerate([0, 1, 7, 5000, 3, 12000, 2 * b_size, 40]):
        a = rng.integers(0, M - 1, n).astype(np.uint32)
        b = (a + 1 + rng.integers(0, M, n) % (M - 1 - a)).astype(np.uint32)
        key = np.unique(a.astype(np.uint64) << 32 | b)                       # distinct pairs, sorted
        rec = np.zeros(len(key), dtype=T.RECORD_DTYPE)
        rec["idxA"] = key >> 32; rec["idxB"] = key & 0xFFFFFFFF
        rec["flags"] = 3; rec["R2"] = rng.random(len(key)); rec["cnt"] = rng.integers(0, 20, (len(key), 4))
        want.append(rec.copy())
        if k % 2:
            rec = rec[rng.permutation(len(rec))]                               # unsorted input is put in order
        st.append(rec)
    n_written = st.close()
    want = np.concatenate(want)
    got, info = hostlib.read_two(path)
    assert n_written == 2 * len(want) == len(got)
    state, ent, _ = hostlib.two_index(path)
    assert state == 0 and ent[:, 2].sum() == len(got) and len(ent) % 2 == 0
    # blocks come in forward / reverse pairs; walk them
    o = 0
    fwd = []
    for k in range(0, len(ent), 2):
        n = int(ent[k, 2])
        assert n == ent[k + 1, 2] and 0 < n <= b_size
        f, v = got[o:o + n], got[o + n:o + 2 * n]
        o += 2 * n
        assert (f["ridA"] == f["ridA"][0]).all() and (f["ridB"] == f["ridB"][0]).all()
        assert np.array_equal(v["ridA"], f["ridB"]) and np.array_equal(v["ridB"], f["ridA"])
        assert np.array_equal(v["packA"], f["packB"]) and np.array_equal(v["packB"], f["packA"])
        assert np.array_equal(v["R2"], f["R2"]) and np.array_equal(v["cnt"], f["cnt"])
        fwd.append(f)
    # only a contig change (or the end) closes a block early
    for k in range(len(fwd) - 1):
        if len(fwd[k]) < b_size:
            assert (fwd[k]["ridA"][0], fwd[k]["ridB"][0]) != (fwd[k + 1]["ridA"][0], fwd[k + 1]["ridB"][0])
    fwd = np.concatenate(fwd)
    assert np.array_equal(fwd["ridA"], rid[want["idxA"]]) and np.array_equal(fwd["ridB"], rid[want["idxB"]])
    assert np.array_equal(fwd["packA"] >> 2, pos[want["idxA"]]) and np.array_equal(fwd["packB"] >> 2, pos[want["idxB"]])
    assert np.array_equal(fwd["R2"], want["R2"]) and np.array_equal(fwd["cnt"], want["cnt"])


def test_mapped_and_stream_output_write_the_same_file(tmp_path):
    """TwoWriter's mapped mode (map_output: frames given their place by an ordered placing step, copied into a shared mapping
    by the emitter's workers in parallel, file cut to size at close) against the stream (one append at a time): the same
    bytes, for ragged appends on three contigs, and the index at the end reads back.  The stream is the default - the
    mapping measured slower on the GPU box (profiles/r04_writer_ab.txt) - but stays selectable (engine option map_output)."""
    import hashlib
    M = 600
    rng = np.random.default_rng(9)
    rid = np.repeat(np.arange(3), M // 3).astype(np.uint32)
    pos = np.concatenate([1000 + 10 * np.arange(M // 3)] * 3).astype(np.uint32)
    recs = np.zeros(70_000, dtype=T.RECORD_DTYPE)
    a = np.sort(rng.integers(0, M - 1, len(recs)).astype(np.uint32))
    recs["idxA"] = a
    recs["idxB"] = np.minimum(a + 1 + rng.integers(0, 40, len(recs)).astype(np.uint32), M - 1)
    recs = recs[np.lexsort((recs["idxB"], recs["idxA"]))]
    for f in ("D", "Dprime", "R", "R2", "P", "ChiSqFisher"):
        recs[f] = rng.random(len(recs))
    recs["cnt"] = rng.integers(0, 500, (len(recs), 4))
    digests = {}
    for mapped in (False, True, 2):                        # 2: direct mode (one pwritev a frame, space reserved ahead; twk_format.h)
        path = str(tmp_path / f"m{int(mapped)}.two")
        st = hostlib.TwoStream(path, 50, rid, pos, n_contigs=3, b_size=1000, n_threads=5, map_output=mapped)
        k = 0
        for n in (1, 999, 1000, 1001, 7, 30_000, 0, 12_345):
            st.append(recs[k:k + n]); k += n
        st.append(recs[k:])
        assert st.close() == 2 * len(recs)
        digests[mapped] = hashlib.sha256(open(path, "rb").read()).hexdigest()
        back, info = hostlib.read_two(path)
        assert len(back) == 2 * len(recs) and info["n_blocks"] > 100
    assert digests[False] == digests[True] == digests[2]


def test_record_emitter_and_hand_off_queue_are_clean_under_tsan():
    """`make tsan`: the format + emitter sources built with -fsanitize=thread around csrc/tools/emitter_tsan.cpp - ragged
    pieces of sorted survivors through RecordEmitter with 1 / 5 / 16 workers, with and without a backlog of expanded
    blocks, the mapped output and the hand-off queue (RecordHandOff, what `tomahawk calc` puts between the engine's thread
    and the emitter); a data race fails the run (halt_on_error), so does a record count that is off."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "ASAN_OPTIONS", "UBSAN_OPTIONS")}     # (`make asan-test` preloads another sanitiser's runtime)
    r = subprocess.run(["make", "-C", root, "tsan"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "24 configurations, 0 bad" in r.stdout


def test_usable_cpus_respects_affinity_and_the_cgroup_quota():
    """Thread pools on the host side are sized by util::usable_cpus (twk_util.h): hardware threads, cut to the scheduler
    affinity mask and to the container's CFS quota - beyond the quota every thread of the container is frozen until the
    next period, the one feeding the GPU included (the GPU boxes of this pool: 256 hardware threads, a quota of 16)."""
    n = hostlib.usable_cpus()
    want = len(os.sched_getaffinity(0))
    for quota_file, period_file in (("/sys/fs/cgroup/cpu.max", None), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            words = open(quota_file).read().split()
            quota = -1 if words[0] == "max" else int(words[0])
            period = int(words[1]) if period_file is None else int(open(period_file).read().split()[0])
        except (OSError, IndexError, ValueError):
            continue
        if quota > 0 and period > 0:
            want = min(want, -(-quota // period))
        break
    assert n == max(1, want)
