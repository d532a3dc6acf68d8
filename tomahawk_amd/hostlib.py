"""ctypes binding of the flat C API of lib/libtomahawk_amd.so (tomahawk_amd/csrc/host/twk_capi.cpp):
.twk / .two file helpers and twk_ld::Compute, for tests, tools and the benchmark harness."""
import ctypes as C
import os

import numpy as np

from .hip import META_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))
# TWK_HOST_LIB / TWK_CLI: run the same tests against another build of the host side (`make asan-test`)
LIB_PATH = os.environ.get("TWK_HOST_LIB") or os.path.join(_HERE, "lib", "libtomahawk_amd.so")
CLI_PATH = os.environ.get("TWK_CLI") or os.path.join(_HERE, "bin", "tomahawk")

# twk1_two_t as serialised (reference lib/core.cpp:470-490): 106 bytes, packed
TWO_DTYPE = np.dtype({"names": ["controller", "ridA", "ridB", "packA", "packB", "cnt", "D", "Dprime", "R", "R2", "P",
                                "ChiSqFisher", "ChiSqModel"],
                      "formats": ["<u2", "<u4", "<u4", "<u4", "<u4", ("<f8", (4,)), "<f8", "<f8", "<f8", "<f8", "<f8",
                                  "<f8", "<f8"],
                      "offsets": [0, 2, 6, 10, 14, 18, 50, 58, 66, 74, 82, 90, 98], "itemsize": 106})

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not built: run `make host`")
        L = C.CDLL(LIB_PATH)
        p = C.c_void_p
        L.twk_file_write_twk.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, p, p, p, p, p, C.c_uint32, C.c_uint32, C.c_int]
        L.twk_file_write_synthetic_twk.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_uint32,
                                                   C.c_int, C.c_int]
        L.twk_file_write_cohort_twk.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_double, C.c_double,
                                                C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_uint32, C.c_uint32,
                                                C.c_uint32, C.c_int, C.c_int]
        L.twk_file_read_twk.argtypes = [C.c_char_p, p, p, p, p, p, p]
        L.twk_file_read_two.argtypes = [C.c_char_p, p, C.c_uint64, p, p]
        L.twk_file_write_two.argtypes = [C.c_char_p, p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.twk_file_two_index.argtypes = [C.c_char_p, p, C.c_uint64, p, C.c_uint64, p]
        L.twk_two_sort.argtypes = [C.c_char_p, C.c_char_p, C.c_double, C.c_int, C.c_int]
        L.twk_import_vcf.argtypes = [C.c_char_p, C.c_char_p, C.c_double, C.c_double, C.c_int, C.c_uint32, C.c_int, C.c_int, p]
        L.twk_hwe_exact.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.twk_hwe_exact.restype = C.c_double
        L.twk_record_codec_bound.restype = C.c_uint64
        L.twk_record_codec_bound.argtypes = [C.c_uint64]
        L.twk_record_codec_compress.restype = C.c_uint64
        L.twk_record_codec_compress.argtypes = [p, C.c_uint64, C.c_uint32, p, C.c_uint64]
        L.twk_file_header_literals.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
        L.twk_two_stream_open.restype = C.c_void_p
        L.twk_two_stream_open.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, p, p, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int]
        L.twk_two_stream_append.argtypes = [p, p, C.c_uint64]
        L.twk_two_stream_close.argtypes = [p, C.POINTER(C.c_uint64)]
        L.twk_ld_compute.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, p, p]
        _lib = L
    return _lib


def write_twk(path, alleles, pos, rid, phased, n_contigs=1, block_size=500, hwe=None, c_level=1):
    M, N, _ = alleles.shape
    al = np.ascontiguousarray(alleles.reshape(M, 2 * N), dtype=np.int8)
    pos = np.ascontiguousarray(pos, dtype=np.uint32)
    rid = np.ascontiguousarray(rid, dtype=np.uint32)
    ph = np.ascontiguousarray(phased, dtype=np.uint8)
    hw = None if hwe is None else np.ascontiguousarray(hwe, dtype=np.float64)
    rc = lib().twk_file_write_twk(path.encode(), N, M, al.ctypes.data, pos.ctypes.data, rid.ctypes.data, ph.ctypes.data,
                                  None if hw is None else hw.ctypes.data, n_contigs, block_size, c_level)
    if rc != 0:
        raise RuntimeError(f"twk_file_write_twk failed: {rc}")


def write_synthetic_twk(path, n_samples, n_variants, seed=42, phased=False, block_size=50, c_level=1, n_threads=8):
    rc = lib().twk_file_write_synthetic_twk(path.encode(), n_samples, n_variants, seed, int(phased), block_size, c_level,
                                            n_threads)
    if rc != 0:
        raise RuntimeError(f"twk_file_write_synthetic_twk failed: {rc}")


def write_cohort_twk(path, n_samples, n_variants, seed=1, n_founders=12, p_switch=0.02, p_mut=0.0005, rare_frac=0.7,
                     max_rare_af=0.01, miss_variants=0.0, miss_rate=0.0, phased=True, n_contigs=1, spacing=100, block_size=200,
                     c_level=1, n_threads=8):
    """A .twk shaped like real cohort data (founder mosaics, 1/x spectrum of rare variants, optional missing samples)."""
    rc = lib().twk_file_write_cohort_twk(path.encode(), n_samples, n_variants, seed, n_founders, p_switch, p_mut, rare_frac,
                                         max_rare_af, miss_variants, miss_rate, int(phased), n_contigs, spacing, block_size,
                                         c_level, n_threads)
    if rc != 0:
        raise RuntimeError(f"twk_file_write_cohort_twk failed: {rc}")


def read_twk(path):
    n_s, n_v = C.c_uint32(), C.c_uint32()
    rc = lib().twk_file_read_twk(path.encode(), C.byref(n_s), C.byref(n_v), None, None, None, None)
    if rc != 0:
        raise RuntimeError(f"twk_file_read_twk failed: {rc}")
    N, M = n_s.value, n_v.value
    w = (2 * N + 63) // 64
    data = np.zeros((M, w), dtype=np.uint64)
    mask = np.zeros((M, w), dtype=np.uint64)
    meta = np.zeros(M, dtype=META_DTYPE)
    extra = np.zeros((M, 4), dtype=np.uint32)
    rc = lib().twk_file_read_twk(path.encode(), None, None, data.ctypes.data, mask.ctypes.data, meta.ctypes.data,
                                 extra.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"twk_file_read_twk failed: {rc}")
    return N, data, mask, meta, extra


def read_two(path):
    n = C.c_uint64()
    info = (C.c_uint64 * 4)()
    rc = lib().twk_file_read_two(path.encode(), None, 0, C.byref(n), info)
    if rc != 0:
        raise RuntimeError(f"twk_file_read_two failed: {rc}")
    recs = np.zeros(n.value, dtype=TWO_DTYPE)
    if n.value:
        rc = lib().twk_file_read_two(path.encode(), recs.ctypes.data, n.value, C.byref(n), info)
        if rc != 0:
            raise RuntimeError(f"twk_file_read_two failed: {rc}")
    return recs, dict(n_samples=info[0], n_contigs=info[1], n_blocks=info[2], state=info[3])


def write_two(path, recs, n_samples=1, n_contigs=1, block_records=10000, c_level=1):
    """Unsorted .two from TWO_DTYPE records (the shape `calc` writes)."""
    recs = np.ascontiguousarray(recs, dtype=TWO_DTYPE)
    rc = lib().twk_file_write_two(path.encode(), recs.ctypes.data, len(recs), n_samples, n_contigs, block_records, c_level)
    if rc != 0:
        raise RuntimeError(f"twk_file_write_two failed: {rc}")


def two_index(path):
    """Index of a .two: (state, entries [n,6] = rid, ridB, n, minpos, maxpos, b_unc; contigs [m,5] = rid, n, minpos, maxpos, nn)."""
    cnt = (C.c_uint64 * 3)()
    rc = lib().twk_file_two_index(path.encode(), None, 0, None, 0, cnt)
    if rc != 0:
        raise RuntimeError(f"twk_file_two_index failed: {rc}")
    ent = np.zeros((cnt[1], 6), dtype=np.int64)
    ctg = np.zeros((cnt[2], 5), dtype=np.int64)
    rc = lib().twk_file_two_index(path.encode(), ent.ctypes.data, cnt[1], ctg.ctypes.data, cnt[2], cnt)
    if rc != 0:
        raise RuntimeError(f"twk_file_two_index failed: {rc}")
    return int(cnt[0]), ent, ctg


def sort_two(path_in, path_out, memory_limit_gb=0.5, c_level=1, n_threads=0):
    """`tomahawk sort` (two_reader::Sort): sorted .two with a sorted index."""
    rc = lib().twk_two_sort(path_in.encode(), path_out.encode(), memory_limit_gb, c_level, n_threads)
    if rc != 0:
        raise RuntimeError(f"twk_two_sort failed: {rc}")


IMPORT_COUNTERS = ("invariant", "missing_threshold", "insufficient_samples", "mixed_ploidy", "no_genotypes", "no_format",
                   "not_biallelic", "not_snp", "hwe", "duplicates", "sites", "written")


def import_vcf(path_in, path_out, threshold_miss=0.9, hwe=0.0, remove_univariate=True, block_size=500, c_level=1,
               n_threads=0):
    """`tomahawk import` (twk_variant_importer::Import): VCF text (plain / gzip) -> .twk. Returns the site counters."""
    cnt = (C.c_uint64 * 12)()
    rc = lib().twk_import_vcf(path_in.encode(), path_out.encode(), threshold_miss, hwe, int(remove_univariate), block_size,
                              c_level, n_threads, cnt)
    if rc != 0:
        raise RuntimeError(f"twk_import_vcf failed: {rc}")
    return dict(zip(IMPORT_COUNTERS, (int(x) for x in cnt)))


def usable_cpus() -> int:
    """CPUs the host side sizes its thread pools by (hardware threads, affinity mask, the container's CFS quota)."""
    return int(lib().twk_usable_cpus())


RECORD_CODEC_LEVEL = (1 << 20) + 1  # as a compression level of a .two writer: the records' own zstd encoder (twk_repcodec.h), libzstd level 1 where it does not fit


def record_codec_compress(data: bytes, stride: int = 106) -> bytes:
    """One zstd frame of `data` from the records' own encoder (matches only against the byte `stride` back)."""
    src = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    dst = np.zeros(int(lib().twk_record_codec_bound(len(data))), dtype=np.uint8)
    n = int(lib().twk_record_codec_compress(src.ctypes.data, len(data), stride, dst.ctypes.data, len(dst)))
    if n == 0:
        raise RuntimeError("twk_record_codec_compress failed")
    return dst[:n].tobytes()


def hwe_exact(hom1, het, hom2):
    return float(lib().twk_hwe_exact(hom1, het, hom2))


def header_literals(path, is_two=True):
    buf = C.create_string_buffer(1 << 16)
    rc = lib().twk_file_header_literals(path.encode(), int(is_two), buf, len(buf))
    if rc != 0:
        raise RuntimeError(f"twk_file_header_literals failed: {rc}")
    return buf.value.decode()


def two_as_matrix(recs):
    """TWO_DTYPE records -> float64 [n, 16] in the column order of `tomahawk_ref dump`."""
    out = np.zeros((len(recs), 16))
    out[:, 0] = recs["controller"]; out[:, 1] = recs["ridA"]; out[:, 2] = recs["packA"] >> 2
    out[:, 3] = recs["ridB"]; out[:, 4] = recs["packB"] >> 2
    out[:, 5:9] = recs["cnt"]
    for i, f in enumerate(("D", "Dprime", "R", "R2", "P", "ChiSqFisher", "ChiSqModel")):
        out[:, 9 + i] = recs[f]
    return out


class TwoStream:
    """The writer side of a calc run: engine records (tomahawk_amd.RECORD_DTYPE, variant indices) -> forward +
    reverse blocks with the reference's flush rule -> a .two file.  What rank 0 of a multi-process run does
    with the records gathered from the other ranks."""

    def __init__(self, path, n_samples, rid, pos, n_contigs=1, c_level=1, b_size=10000, n_threads=4, map_output=False):
        rid = np.ascontiguousarray(rid, dtype=np.uint32)
        pos = np.ascontiguousarray(pos, dtype=np.uint32)
        assert rid.shape == pos.shape
        self._h = lib().twk_two_stream_open(path.encode(), n_samples, n_contigs, rid.ctypes.data, pos.ctypes.data, len(rid),
                                            c_level, b_size, n_threads, int(map_output))      # 0 stream, 1 (True) mapped, 2 direct (pwritev, space reserved ahead)
        if not self._h:
            raise RuntimeError(f"twk_two_stream_open failed for {path}")

    def append(self, recs):
        from .hip import RECORD_DTYPE
        recs = np.ascontiguousarray(recs, dtype=RECORD_DTYPE)
        rc = lib().twk_two_stream_append(self._h, recs.ctypes.data, len(recs))
        if rc != 0:
            raise RuntimeError(f"twk_two_stream_append failed: {rc}")

    def close(self):
        """-> number of records written (forward + reverse)."""
        n = C.c_uint64()
        h, self._h = self._h, None
        rc = lib().twk_two_stream_close(h, C.byref(n))
        if rc != 0:
            raise RuntimeError(f"twk_two_stream_close failed: {rc}")
        return n.value
