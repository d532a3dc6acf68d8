"""Is there a cheaper way to write level-1 zstd for .two blocks?  Realistic records (the compiled reference's own
`calc -p -w 1000000` on 2,504 x 12,000 cohort-shaped variants), cut into blocks of 10,000 records as the writer does, through
libzstd with one advanced parameter changed at a time: MB/s on one core, ratio.  (CPU only.)
  python tests/sweeps/zstd_params.py"""
import ctypes as C, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from tomahawk_amd import hostlib as H
twk, two = "/tmp/zstd_params.twk", "/tmp/zstd_params.two"
if not os.path.exists(two):
    H.write_cohort_twk(twk, 2504, 12000, seed=12, n_threads=8, block_size=500, spacing=100)
    subprocess.run([O.REF_BIN, "calc", "-i", twk, "-o", two, "-p", "-w", "1000000", "-t", "8"], check=True, capture_output=True)
raw = H.read_two(two)[0].tobytes()
B = 10000 * 106
blocks = [b"\x10\x27\x00\x00\x10\x27\x00\x00" + raw[i:i + B] for i in range(0, len(raw) - B, B)][:120]
total = sum(len(b) for b in blocks)
z = C.CDLL("/usr/lib/x86_64-linux-gnu/libzstd.so.1")
z.ZSTD_createCCtx.restype = C.c_void_p
z.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]; z.ZSTD_CCtx_setParameter.restype = C.c_size_t
z.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]; z.ZSTD_compress2.restype = C.c_size_t
z.ZSTD_compressBound.restype = C.c_size_t; z.ZSTD_compressBound.argtypes = [C.c_size_t]
z.ZSTD_isError.argtypes = [C.c_size_t]; z.ZSTD_versionString.restype = C.c_char_p
print("zstd", z.ZSTD_versionString().decode(), len(blocks), "blocks,", round(total / 1e6, 1), "MB")
P = dict(level=100, windowLog=101, hashLog=102, chainLog=103, searchLog=104, minMatch=105, targetLength=106, strategy=107, literalCompressionMode=1002)
dst = C.create_string_buffer(z.ZSTD_compressBound(len(blocks[0])))


def run(name, **kw):
    cctx = C.c_void_p(z.ZSTD_createCCtx())
    for k, v in kw.items():
        if z.ZSTD_isError(z.ZSTD_CCtx_setParameter(cctx, P[k], v)):
            print("  parameter refused:", k, v); return
    best = None
    for _ in range(3):
        t = time.perf_counter(); out = 0
        for b in blocks:
            r = z.ZSTD_compress2(cctx, dst, len(dst), b, len(b)); assert not z.ZSTD_isError(r); out += r
        best = min(best or 1e9, time.perf_counter() - t)
    print(f"{name:36s} {total / best / 1e6:7.1f} MB/s  ratio {total / out:5.2f}  {out / 1e6:7.2f} MB")


run("level 1 (the default)", level=1)
for lv in (-1, -3, 2, 3): run(f"level {lv}", level=lv)
for k, vals in (("minMatch", (5, 7)), ("hashLog", (12, 16, 17)), ("windowLog", (17, 21)), ("targetLength", (2, 4)), ("literalCompressionMode", (2,)), ("strategy", (2,))):
    for v in vals: run(f"level 1, {k} {v}", level=1, **{k: v})


# ---- round 5: would a GPU encoder that only entropy-codes (Huffman literals, no sequences) be close enough? ------------------
# A zstd frame whose blocks carry all bytes as Huffman-compressed literals and zero sequences is valid and needs no match search -
# the part of the format a GPU can produce at memory speed (histogram -> code lengths -> four interleaved streams per 128 KiB
# block).  Its size is the order-0 Huffman cost of every 128 KiB block (code lengths limited to 11 bits, as zstd's literals
# are) plus ~150 bytes of table and headers per block.  Measured on the same records, also with the records' bytes transposed
# (byte k of every record together: what a columnar pre-pass could feed the coder - not the reference's format, shown for scale).
import heapq
import numpy as np


def huffman_bits(hist, limit=11):
    """Total bits of a length-limited Huffman code for the histogram (package-merge replaced by the usual heuristic: lengths
    above the limit are clamped and the Kraft sum repaired by lengthening the rarest symbols still below it)."""
    sym = [(int(c), i) for i, c in enumerate(hist) if c]
    if len(sym) == 1:
        return sym[0][0]
    heap = [(c, i, None, None) for c, i in sym]
    heapq.heapify(heap)
    nxt = 256
    while len(heap) > 1:
        a = heapq.heappop(heap); b = heapq.heappop(heap)
        heapq.heappush(heap, (a[0] + b[0], nxt, a, b)); nxt += 1
    lengths = {}
    stack = [(heap[0], 0)]
    while stack:
        (c, i, l, r), d = stack.pop()
        if l is None:
            lengths[i] = max(d, 1)
        else:
            stack.append((l, d + 1)); stack.append((r, d + 1))
    ln = {i: min(l, limit) for i, l in lengths.items()}
    kraft = sum(2.0 ** -l for l in ln.values())
    order = sorted(ln, key=lambda i: hist[i])          # rarest first
    k = 0
    while kraft > 1.0 + 1e-12:
        i = order[k % len(order)]
        if ln[i] < limit:
            kraft -= 2.0 ** -ln[i] / 2; ln[i] += 1
        k += 1
    return sum(int(hist[i]) * l for i, l in ln.items())


def entropy_only(name, data_blocks):
    out = 0
    t = time.perf_counter()
    for b in data_blocks:
        a = np.frombuffer(b, dtype=np.uint8)
        for k in range(0, len(a), 131072):
            h = np.bincount(a[k:k + 131072], minlength=256)
            out += (huffman_bits(h) + 7) // 8 + 150
    print(f"{name:36s} {'(estimate)':>12s}  ratio {total / out:5.2f}  {out / 1e6:7.2f} MB   ({time.perf_counter() - t:.1f} s in python)")


entropy_only("Huffman literals only, record order", blocks)
transposed = []
for b in blocks:
    a = np.frombuffer(b[8:], dtype=np.uint8).reshape(-1, 106)
    transposed.append(b[:8] + a.T.tobytes())
entropy_only("Huffman literals only, bytes transposed", transposed)
run("level 1 on transposed bytes (for scale)", level=1) if False else None


# ---- round 5, second look: what level 1 finds in these records is runs of bytes that equal the previous record's -----------
# (tomahawk_amd/csrc/host/twk_repcodec.h: matches only against the byte 106 back, raw literals, FSE tables per frame)
from tomahawk_amd import hostlib as H2
zd = C.CDLL("/usr/lib/x86_64-linux-gnu/libzstd.so.1")
zd.ZSTD_decompress.restype = C.c_size_t; zd.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
frames = [H2.record_codec_compress(b, 106) for b in blocks]
back = C.create_string_buffer(len(blocks[0]))
for b, f in zip(blocks, frames):
    assert zd.ZSTD_decompress(back, len(back), f, len(f)) == len(b) and back.raw[:len(b)] == b
src = [np.frombuffer(b, np.uint8) for b in blocks]
dstb = np.zeros(int(H2.lib().twk_record_codec_bound(len(blocks[0]))), np.uint8)
best = None
for _ in range(3):
    t = time.perf_counter()
    for s in src: H2.lib().twk_record_codec_compress(s.ctypes.data, len(s), 106, dstb.ctypes.data, len(dstb))
    best = min(best or 1e9, time.perf_counter() - t)
out = sum(len(f) for f in frames)
print(f"{'the records own encoder (decoded by libzstd)':36s} {total / best / 1e6:7.1f} MB/s  ratio {total / out:5.2f}  {out / 1e6:7.2f} MB")
