"""The records' own zstd encoder (csrc/host/twk_repcodec.h; engine option record_codec): every frame it writes must be a frame
that libzstd decodes to the input - which is all the reference's reader asks of a block (lib/zstd_codec.cpp:156-168:
ZSTD_decompress) - for any input, not only for .two records; a .two written with it must read back record for record through
this repo's reader and through the compiled reference's `view`."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import tomahawk_amd as T
from oracle import oracle as O
from tomahawk_amd import hostlib

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _zstd():
    z = C.CDLL("libzstd.so.1")
    z.ZSTD_decompress.restype = C.c_size_t
    z.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    z.ZSTD_isError.argtypes = [C.c_size_t]
    z.ZSTD_getFrameContentSize.restype = C.c_ulonglong
    z.ZSTD_getFrameContentSize.argtypes = [C.c_void_p, C.c_size_t]
    z.ZSTD_compress.restype = C.c_size_t
    z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    z.ZSTD_compressBound.restype = C.c_size_t
    z.ZSTD_compressBound.argtypes = [C.c_size_t]
    return z


def _roundtrip(z, data: bytes, stride: int) -> int:
    frame = hostlib.record_codec_compress(data, stride)
    assert z.ZSTD_getFrameContentSize(frame, len(frame)) == len(data)
    out = C.create_string_buffer(len(data) + 8)
    r = z.ZSTD_decompress(out, len(data) + 8, frame, len(frame))
    assert not z.ZSTD_isError(r) and r == len(data), (len(data), stride, r)
    assert out.raw[:r] == data
    assert len(frame) <= len(data) + 12 + 3 * (len(data) // (1 << 17) + 1)          # never worse than stored
    return len(frame)


def _pattern(rng, n, stride, kind):
    if kind == "random":
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == "zeros":
        return np.zeros(n, np.uint8)
    rec = rng.integers(0, 256 if kind != "small alphabet" else 4, stride, dtype=np.uint8)
    a = np.tile(rec, n // stride + 1)[:n].copy()
    if kind == "sparse flips" and n:
        a[rng.integers(0, n, n // 20)] ^= 1
    elif kind in ("dense noise", "small alphabet"):
        m = rng.random(n) < 0.3
        a[m] = rng.integers(0, 256, int(m.sum()), dtype=np.uint8)
    return a


@pytest.mark.parametrize("kind", ["random", "zeros", "sparse flips", "dense noise", "small alphabet"])
def test_every_frame_decodes_with_libzstd(kind):
    """Sizes around the frame's 128 KiB block edge and the stride; data without a match, with one match that spans blocks
    (zeros: 130 K-byte match lengths, the last match-length code), with thousands of short ones (every literal-length and
    match-length code class, sequence counts in the 1-, 2- and 3-byte forms), and data on which the sequences do not pay, so
    that a block falls back to a raw block while the ones behind it still need their tables and their first offset."""
    z = _zstd()
    rng = np.random.default_rng(3)
    for n in (0, 1, 5, 105, 106, 107, 114, 212, 1000, (1 << 17) - 1, 1 << 17, (1 << 17) + 1, (1 << 17) + 106, 400_000, 1_060_008):
        _roundtrip(z, _pattern(rng, n, 106, kind).tobytes(), 106)
    for stride in (1, 2, 3, 8, 61, 200, 4000, 70_000, 200_000):
        _roundtrip(z, _pattern(rng, 300_000, stride, kind).tobytes(), stride)


def test_long_literal_runs_and_long_matches_take_the_wide_codes():
    """Literal lengths and match lengths beyond 64 K (the 16-bit extra fields) next to short ones in one block; a first block
    without a single match, so that the tables and the explicit offset move to the second."""
    z = _zstd()
    rng = np.random.default_rng(4)
    a = rng.integers(0, 256, 700_000, dtype=np.uint8)
    a[70_000:170_000] = np.tile(a[70_000 - 106:70_000], 100_000 // 106 + 1)[:100_000]      # one 100,000-byte match behind 70,000 literals
    a[300_000:300_010] = a[300_000 - 106:300_010 - 106].copy()
    a[500_000:] = np.tile(a[500_000 - 106:500_000], 200_000 // 106 + 1)[:200_000]
    assert _roundtrip(z, a.tobytes(), 106) < 420_000


def _records(n, seed=1):
    """Records shaped like a calc run's: sorted pairs, slowly varying fields, noisy mantissas."""
    rng = np.random.default_rng(seed)
    r = np.zeros(n, dtype=hostlib.TWO_DTYPE)
    a = np.sort(rng.integers(0, 5000, n))
    r["controller"] = 3
    r["ridA"] = 0; r["ridB"] = 0
    r["packA"] = (1000 + 50 * a).astype(np.uint32) << 2
    r["packB"] = (1000 + 50 * (a + 1 + rng.integers(0, 200, n))).astype(np.uint32) << 2
    r["cnt"] = rng.integers(0, 5000, (n, 4)).astype(np.float64)
    for f in ("D", "Dprime", "R", "R2", "P", "ChiSqFisher", "ChiSqModel"):
        r[f] = rng.random(n)
    return r


def test_two_file_written_with_the_record_codec_reads_back(tmp_path):
    """write_two at RECORD_CODEC_LEVEL: the reader of this repo (frame content size checked against the block header) and the
    compiled reference's `view` return what the same records written through libzstd return; the file is within 15 % of it."""
    recs = _records(45_000)
    a, b = str(tmp_path / "zstd.two"), str(tmp_path / "codec.two")
    hostlib.write_two(a, recs, n_samples=10, c_level=1)
    hostlib.write_two(b, recs, n_samples=10, c_level=hostlib.RECORD_CODEC_LEVEL)
    ra, ia = hostlib.read_two(a)
    rb, ib = hostlib.read_two(b)
    assert ra.tobytes() == rb.tobytes() == recs.tobytes() and ia["n_blocks"] == ib["n_blocks"]
    assert os.path.getsize(b) < 1.15 * os.path.getsize(a)
    if O.have_ref():
        va = O.run_ref(["view", "-i", a]).stdout
        vb = O.run_ref(["view", "-i", b]).stdout
        body = lambda s: [l for l in s.splitlines() if not l.startswith("##tomahawk_viewCommand")]
        assert body(va) == body(vb) and len(body(vb)) > 45_000


def test_record_stream_with_the_codec_writes_the_same_records(tmp_path):
    """hostlib.TwoStream (the emitter calc uses) at RECORD_CODEC_LEVEL against level 1: same blocks, same records, same index
    but for the compressed sizes."""
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
    got = {}
    for level in (1, hostlib.RECORD_CODEC_LEVEL):
        path = str(tmp_path / f"s{level}.two")
        st = hostlib.TwoStream(path, 50, rid, pos, n_contigs=3, c_level=level, b_size=1000, n_threads=5)
        k = 0
        for n in (1, 999, 1000, 1001, 7, 30_000, 0, 12_345):
            st.append(recs[k:k + n]); k += n
        st.append(recs[k:])
        assert st.close() == 2 * len(recs)
        back, info = hostlib.read_two(path)
        state, ent, _ = hostlib.two_index(path)
        got[level] = (back.tobytes(), info["n_blocks"], ent[:, :5].tolist())
    assert got[1] == got[hostlib.RECORD_CODEC_LEVEL]


def test_blocks_the_encoder_is_not_made_for_go_through_libzstd(tmp_path):
    """Records of a 64-sample run (statistics that take few distinct values: libzstd level 1 finds 2.4 x what the encoder
    finds) and records in shuffled order (nothing one record back): the writer tries the head of every block both ways and
    leaves blocks whose sample comes out a quarter larger to libzstd, so no file is more than 15 % larger than the level-1 file; a block of a large cohort's records stays
    with the encoder (and the file within 15 % of level 1: the bar the round-4 review set)."""
    z = _zstd()
    small, _ = hostlib.read_two(os.path.join(GOLDEN, "ref_n64_small_p.two"))
    small = np.concatenate([small] * 4)                    # > 128 KiB a block, so that the choice is made at all
    shuffled = _records(30_000)[np.random.default_rng(2).permutation(30_000)]
    sizes = {}
    for name, recs in (("small", small), ("shuffled", shuffled), ("cohort", _records(30_000))):
        for level in (1, hostlib.RECORD_CODEC_LEVEL):
            path = str(tmp_path / f"{name}{level}.two")
            hostlib.write_two(path, recs, n_samples=10, c_level=level)
            assert hostlib.read_two(path)[0].tobytes() == recs.tobytes()
            sizes[name, level] = os.path.getsize(path)
    assert sizes["small", hostlib.RECORD_CODEC_LEVEL] == sizes["small", 1]
    assert sizes["shuffled", hostlib.RECORD_CODEC_LEVEL] < 1.15 * sizes["shuffled", 1]
    assert sizes["cohort", 1] < sizes["cohort", hostlib.RECORD_CODEC_LEVEL] < 1.15 * sizes["cohort", 1]
    # the encoder alone on the small-cohort records: valid frames, far larger - which is why the writer checks
    blk = np.uint32(len(small)).tobytes() * 2 + small.tobytes()
    dst = C.create_string_buffer(z.ZSTD_compressBound(len(blk)))
    assert _roundtrip(z, blk, 106) > 1.5 * z.ZSTD_compress(dst, len(dst), blk, len(blk), 1)


def test_randomised_inputs_decode(tmp_path):
    """300 random inputs: strides 1 .. 400, sizes around one to three blocks, a random mix of segments that repeat the bytes one
    stride back, flip a few of them, or are noise - so that matches start and end anywhere relative to the 128 KiB block edges,
    follow each other with no literal between them, and blocks with and without sequences alternate within a frame."""
    z = _zstd()
    rng = np.random.default_rng(12)
    for case in range(300):
        stride = int(rng.integers(1, 401))
        n = int(rng.choice([rng.integers(0, 5000), (1 << 17) + rng.integers(-300, 300), 2 * (1 << 17) + rng.integers(-300, 300), rng.integers(1 << 17, 3 << 17)]))
        a = rng.integers(0, 256, n, dtype=np.uint8)
        pos = stride
        while pos < n:
            seg = int(rng.choice([1, 2, 3, 4, 5, 8, 40, 300, 5000, 140_000]))
            end = min(n, pos + seg)
            kind = rng.integers(0, 3)
            if kind == 0:                                  # equal to the bytes one stride back (propagating: a long match)
                for q in range(pos, end, stride):
                    e2 = min(end, q + stride)
                    a[q:e2] = a[q - stride:e2 - stride]
            elif kind == 1 and end - pos > 8:              # equal but for a few bytes
                for q in range(pos, end, stride):
                    e2 = min(end, q + stride)
                    a[q:e2] = a[q - stride:e2 - stride]
                a[rng.integers(pos, end, max(1, (end - pos) // 50))] ^= 0x5A
            pos = end
        _roundtrip(z, a.tobytes(), stride)


def test_sort_writes_the_same_sorted_file_with_the_record_codec(tmp_path):
    """`tomahawk sort -c 1048577` (compression level 2^20 + k: the records' encoder, libzstd level k where it does not fit): the sorted
    records, the blocks and the index (but for the compressed sizes) are those of `-c 1`."""
    recs = _records(60_000, seed=3)
    recs = recs[np.random.default_rng(4).permutation(len(recs))]
    src = str(tmp_path / "in.two")
    hostlib.write_two(src, recs, n_samples=10)
    got = {}
    for level in (1, hostlib.RECORD_CODEC_LEVEL):
        out = str(tmp_path / f"sorted{level}.two")
        r = subprocess.run([hostlib.CLI_PATH, "sort", "-i", src, "-o", out, "-c", str(level), "-t", "4"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        back, info = hostlib.read_two(out)
        state, ent, contigs = hostlib.two_index(out)
        got[level] = (back.tobytes(), info["n_blocks"], state, ent[:, :5].tolist(), contigs.tolist())
    assert got[1] == got[hostlib.RECORD_CODEC_LEVEL] and got[1][2] == 2
