"""TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/liboracle.so (oracle/ld_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_BIN = os.path.join(_HERE, "_ref", "tomahawk_ref")

VARIANT_DTYPE = np.dtype([("ac", "<u4"), ("an", "<u4"), ("pos", "<u4"), ("rid", "<u4"),
                          ("gt_missing", "<u4"), ("gt_phase", "<u4"), ("hwe", "<f8")])
RECORD_DTYPE = np.dtype([("controller", "<u4"), ("ridA", "<u4"), ("ridB", "<u4"), ("Apos", "<u4"),
                         ("Bpos", "<u4"), ("_pad", "<u4"), ("cnt", "<f8", (4,)), ("D", "<f8"),
                         ("Dprime", "<f8"), ("R", "<f8"), ("R2", "<f8"), ("P", "<f8"),
                         ("ChiSqFisher", "<f8"), ("ChiSqModel", "<f8")])
assert VARIANT_DTYPE.itemsize == 32 and RECORD_DTYPE.itemsize == 112


class Settings(C.Structure):
    _fields_ = [("minR2", C.c_double), ("maxR2", C.c_double), ("minDprime", C.c_double),
                ("maxDprime", C.c_double), ("minP", C.c_double), ("force_phased", C.c_int),
                ("forced_unphased", C.c_int), ("keep_low_ac", C.c_int), ("ref_compat", C.c_int)]


_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "oracle"], check=True, stdout=subprocess.DEVNULL)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        p = C.c_void_p
        L.orc_default_settings.argtypes = [C.POINTER(Settings)]
        L.orc_words64.restype = C.c_uint32
        L.orc_words64.argtypes = [C.c_uint32]
        L.orc_build_bitvector.argtypes = [p, p, p, C.c_uint32, C.c_uint32, p, p]
        L.orc_count_phased.argtypes = [p, p, p, p, C.c_uint32, p]
        L.orc_count_phased_rle.argtypes = [p, p, p, p, C.c_uint32, p]
        L.orc_count_phased_k3_as_is.argtypes = [p, p, p, p, C.c_uint32, p]
        L.orc_count_unphased.argtypes = [p, p, p, p, C.c_uint32, p]
        L.orc_fisher_exact.restype = C.c_double
        L.orc_fisher_exact.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_double)] * 3
        L.orc_phased_math.argtypes = [p, p, p, C.POINTER(Settings), p]
        L.orc_unphased_math.argtypes = [p, p, p, C.POINTER(Settings), p]
        L.orc_all_pairs.restype = C.c_uint64
        L.orc_all_pairs.argtypes = [p, p, p, C.c_uint32, C.c_uint32, C.POINTER(Settings), C.c_int, p]
        L.orc_all_pairs_rows.restype = C.c_uint64
        L.orc_all_pairs_rows.argtypes = [p, p, p, C.c_uint32, C.c_uint32, C.POINTER(Settings), C.c_int, C.c_uint32, C.c_uint32, p]
        L.orc_pack_record.argtypes = [p, p]
        _lib = L
    return _lib


def settings(minR2=0.1, maxR2=100.0, minDprime=0.0, maxDprime=100.0, minP=1.0, phased=False, unphased=False,
             keep_low_ac=False, ref_compat=False) -> Settings:
    return Settings(minR2, maxR2, minDprime, maxDprime, minP, int(phased), int(unphased), int(keep_low_ac), int(ref_compat))


def words64(n_samples: int) -> int:
    return (2 * n_samples + 63) // 64


def bitvectors_from_alleles(alleles: np.ndarray):
    """alleles int8 [M, N, 2] in {0,1,2} -> (data, mask|None) uint64 [M, words64] (twk_igt_vec layout)."""
    M, N, _ = alleles.shape
    w = words64(N)
    flat = alleles.reshape(M, 2 * N)
    miss_sample = (alleles == 2).any(axis=2)
    bits = (flat == 1).astype(np.uint8)
    mbits = np.repeat(miss_sample, 2, axis=1).astype(np.uint8)
    pad = w * 64 - 2 * N

    def pack(b):
        b = np.pad(b, ((0, 0), (0, pad)))
        return np.packbits(b, axis=1, bitorder="little").view(np.uint64).reshape(M, w).copy()

    data = pack(bits)
    mask = pack(mbits) if miss_sample.any() else None
    return data, mask


def variants_from_alleles(alleles: np.ndarray, pos=None, rid=None, phase=1, hwe=None) -> np.ndarray:
    M, N, _ = alleles.shape
    v = np.zeros(M, dtype=VARIANT_DTYPE)
    v["ac"] = (alleles == 1).sum(axis=(1, 2))
    v["an"] = (alleles == 2).sum(axis=(1, 2))
    v["pos"] = np.arange(M) * 100 + 1000 if pos is None else pos
    v["rid"] = 0 if rid is None else rid
    v["gt_missing"] = (alleles == 2).any(axis=(1, 2))
    v["gt_phase"] = phase
    v["hwe"] = 1.0 if hwe is None else hwe
    return v


def count_phased(a, ma, b, mb, n_samples):
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_count_phased(a.ctypes.data, None if ma is None else ma.ctypes.data, b.ctypes.data,
                           None if mb is None else mb.ctypes.data, n_samples, out.ctypes.data)
    return out


def count_unphased(a, ma, b, mb, n_samples):
    out = np.zeros(9, dtype=np.uint64)
    lib().orc_count_unphased(a.ctypes.data, None if ma is None else ma.ctypes.data, b.ctypes.data,
                             None if mb is None else mb.ctypes.data, n_samples, out.ctypes.data)
    return out


def fisher(n11, n12, n21, n22):
    l, r, t = C.c_double(), C.c_double(), C.c_double()
    lib().orc_fisher_exact(n11, n12, n21, n22, C.byref(l), C.byref(r), C.byref(t))
    return l.value, r.value, t.value


def phased_math(c4, A, B, st):
    rec = np.zeros(1, dtype=RECORD_DTYPE)
    c4 = np.ascontiguousarray(c4, dtype=np.uint64)
    A = np.ascontiguousarray(A); B = np.ascontiguousarray(B)
    ok = lib().orc_phased_math(c4.ctypes.data, A.ctypes.data, B.ctypes.data, C.byref(st), rec.ctypes.data)
    return (rec[0] if ok else None)


def unphased_math(c9, A, B, st):
    rec = np.zeros(1, dtype=RECORD_DTYPE)
    c9 = np.ascontiguousarray(c9, dtype=np.uint64)
    A = np.ascontiguousarray(A); B = np.ascontiguousarray(B)
    ok = lib().orc_unphased_math(c9.ctypes.data, A.ctypes.data, B.ctypes.data, C.byref(st), rec.ctypes.data)
    return (rec[0] if ok else None)


def _threads() -> int:
    """Host threads for all_pairs: the CPUs this process may run on, 8 at most (TWK_ORACLE_THREADS overrides)."""
    env = os.environ.get("TWK_ORACLE_THREADS")
    if env:
        return max(1, int(env))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(8, n))


def all_pairs(data, mask, variants, n_samples, st, vector_only=True) -> np.ndarray:
    """Records of all pairs i < j in pair order.  Large problems are cut into row ranges of equal pair
    count and run on several host threads (ctypes drops the GIL; orc_all_pairs_rows keeps no state);
    the ranges are joined in row order, so the result is the serial loop's."""
    M = data.shape[0]
    data = np.ascontiguousarray(data, dtype=np.uint64)
    variants = np.ascontiguousarray(variants, dtype=VARIANT_DTYPE)
    total = M * (M - 1) // 2
    recs = np.zeros(max(total, 1), dtype=RECORD_DTYPE)
    mptr = None
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint64)
        mptr = mask.ctypes.data
    L = lib()
    nt = _threads()
    if nt == 1 or total * data.shape[1] < (1 << 18):
        n = L.orc_all_pairs(data.ctypes.data, mptr, variants.ctypes.data, M, n_samples, C.byref(st),
                            int(vector_only), recs.ctypes.data)
        return recs[:n].copy()
    before = lambda r: r * (2 * M - r - 1) // 2             # pairs of the rows in front of row r
    cuts = [0]
    for k in range(1, 4 * nt):
        r = cuts[-1]
        while r < M and before(r) < total * k // (4 * nt):
            r += 1
        cuts.append(r)
    cuts.append(M)
    ranges = [(a, b) for a, b in zip(cuts, cuts[1:]) if b > a]

    def run(rng):
        r0, r1 = rng
        return L.orc_all_pairs_rows(data.ctypes.data, mptr, variants.ctypes.data, M, n_samples, C.byref(st),
                                    int(vector_only), r0, r1, recs[before(r0):].ctypes.data)

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(nt) as pool:
        counts = list(pool.map(run, ranges))
    return np.concatenate([recs[before(r0):before(r0) + n] for (r0, _), n in zip(ranges, counts)] or [recs[:0]])


def pack_record(rec) -> bytes:
    out = (C.c_uint8 * 106)()
    r = np.ascontiguousarray(np.array([rec], dtype=RECORD_DTYPE))
    lib().orc_pack_record(r.ctypes.data, out)
    return bytes(out)


def have_ref() -> bool:
    return os.path.exists(REF_BIN) and os.access(REF_BIN, os.X_OK)


def run_ref(args, **kw):
    """Run the compiled reference (oracle/_ref/tomahawk_ref)."""
    return subprocess.run([REF_BIN] + list(args), check=True, capture_output=True, text=True, **kw)
