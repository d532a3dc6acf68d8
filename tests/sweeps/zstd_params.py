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
