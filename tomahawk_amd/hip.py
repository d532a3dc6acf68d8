"""ctypes binding of the C ABI in include/twk_hip.h (lib/libtwk_hip.so).

Fails loudly when the HIP library is missing or no device is usable: there is
no CPU path behind these calls.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtwk_hip.so")

ABI_VERSION = 5                 # TWK_HIP_ABI_VERSION of include/twk_hip.h (struct layouts below)
MODE_PHASED, MODE_UNPHASED, MODE_AUTO = 1, 2, 3
# option bits of twk_hip_tile_desc.window / the `window` argument of ld_all / ld_region (TWK_HIP_OPT_*)
OPT_WINDOW, OPT_KEEP_LOW_AC, OPT_REF_COMPAT, OPT_R2_SCREEN = 1, 2, 4, 8
E_OVERFLOW = -4

# twk_hip_record (include/twk_hip.h): 104 bytes
RECORD_DTYPE = np.dtype([("idxA", "<u4"), ("idxB", "<u4"), ("flags", "<u4"), ("_pad", "<u4"),
                         ("cnt", "<f8", (4,)), ("D", "<f8"), ("Dprime", "<f8"), ("R", "<f8"),
                         ("R2", "<f8"), ("P", "<f8"), ("ChiSqFisher", "<f8"), ("ChiSqModel", "<f8")])
assert RECORD_DTYPE.itemsize == 104

# twk_hip_rle_desc: 16 bytes
RLE_DESC_DTYPE = np.dtype([("offset", "<u8"), ("n_runs", "<u4"), ("width", "u1"), ("missing", "u1"), ("_pad", "<u2")])
assert RLE_DESC_DTYPE.itemsize == 16

# twk_hip_variant_meta: 32 bytes
META_DTYPE = np.dtype([("ac", "<u4"), ("an", "<u4"), ("pos", "<u4"), ("rid", "<u4"),
                       ("missing", "<u4"), ("_pad", "<u4"), ("hwe", "<f8")])
assert META_DTYPE.itemsize == 32


class _Filters(C.Structure):
    _fields_ = [("minR2", C.c_double), ("maxR2", C.c_double), ("minDprime", C.c_double),
                ("maxDprime", C.c_double), ("minP", C.c_double)]


class _Tile(C.Structure):
    _fields_ = [("rowA0", C.c_uint32), ("nA", C.c_uint32), ("rowB0", C.c_uint32), ("nB", C.c_uint32),
                ("diag", C.c_int32), ("window", C.c_int32), ("l_window", C.c_uint32), ("_pad", C.c_uint32)]


class _Timing(C.Structure):
    _fields_ = [("count_ms", C.c_double), ("stats_ms", C.c_double), ("count_launches", C.c_uint64),
                ("stats_launches", C.c_uint64), ("row_pairs", C.c_uint64), ("variant_pairs", C.c_uint64),
                ("words_per_row", C.c_uint64), ("fused_launches", C.c_uint64), ("candidates", C.c_uint64),
                ("list_ms", C.c_double), ("list_launches", C.c_uint64), ("list_pairs", C.c_uint64),
                ("probe_ms", C.c_double), ("probe_launches", C.c_uint64), ("probe_pairs", C.c_uint64),
                ("count_shader_cycles", C.c_uint64), ("count_wall_ticks", C.c_uint64),
                ("three_launches", C.c_uint64), ("three_row_pairs", C.c_uint64), ("recount_candidates", C.c_uint64),
                ("outlier_launches", C.c_uint64), ("finish_ms", C.c_double)]


class _PlanEnv(C.Structure):         # twk_hip_plan_env
    _fields_ = [("n_samples", C.c_uint32), ("planes_per_variant", C.c_int32), ("k_chunks", C.c_uint32), ("resident_blocks", C.c_uint32),
                ("screen", C.c_int32), ("fused", C.c_int32), ("phased_math", C.c_int32), ("band_launch", C.c_int32), ("band_reverse", C.c_int32),
                ("_pad", C.c_int32), ("band_work_log2", C.c_int64), ("band_max_launches", C.c_int64), ("minR2", C.c_double)]


TILE_DTYPE = np.dtype([("rowA0", "<u4"), ("nA", "<u4"), ("rowB0", "<u4"), ("nB", "<u4"), ("diag", "<i4"), ("window", "<i4"),
                       ("l_window", "<u4"), ("_pad", "<u4")])       # twk_hip_tile_desc


def plan_region(meta, n_samples, a0, nA, b0, nB, triangle=True, part=0, n_parts=1, tile_variants=0, window=0, l_window=0,
                popc=None, screen=0, minR2=0.1, planes_per_variant=1, k_chunks=1, fused=False, phased_math=True,
                resident_blocks=512, band_launch=True, band_reverse=True, band_work_log2=19, band_max_launches=8):
    """The engine's planner (twk_hip_plan_region: host arithmetic, needs no GPU) -> dict(tiles=TILE_DTYPE array, n_band_launches,
    row_begin, row_end, n_pairs, lo, hi).  meta: META_DTYPE per position of the index space."""
    lib = load_library()
    meta = np.ascontiguousarray(meta, dtype=META_DTYPE)
    env = _PlanEnv(n_samples, planes_per_variant, k_chunks, resident_blocks, screen, int(fused), int(phased_math), int(band_launch),
                   int(band_reverse), 0, band_work_log2, band_max_launches, float(minR2))
    pc = None if popc is None else np.ascontiguousarray(popc, dtype=np.uint32)
    lo = np.zeros(nA, dtype=np.uint32); hi = np.zeros(nA, dtype=np.uint32)
    n_tiles, n_bands, r0, r1, pairs = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_uint64(0)
    cap = 1024
    while True:
        tiles = np.zeros(cap, dtype=TILE_DTYPE)
        rc = lib.twk_hip_plan_region(C.byref(env), meta.ctypes.data, None if pc is None else pc.ctypes.data, len(meta), a0, nA, b0, nB,
                                     int(bool(triangle)), part, n_parts, tile_variants, window, l_window, tiles.ctypes.data, cap,
                                     C.byref(n_tiles), C.byref(n_bands), C.byref(r0), C.byref(r1), C.byref(pairs), lo.ctypes.data, hi.ctypes.data)
        if rc == -4 and n_tiles.value > cap:
            cap = n_tiles.value
            continue
        if rc != 0:
            raise HipError(rc, "twk_hip_plan_region", lib.twk_hip_strerror(rc).decode())
        break
    return dict(tiles=tiles[:n_tiles.value], n_band_launches=n_bands.value, row_begin=r0.value, row_end=r1.value, n_pairs=pairs.value, lo=lo, hi=hi)


class Plant(C.Structure):            # twk_hip_plant: LD planted in the synthetic input (include/twk_hip.h)
    _fields_ = [("n_planted", C.c_uint32), ("half", C.c_uint32), ("mult", C.c_uint32), ("offset", C.c_uint32), ("max_eps", C.c_double)]

    @classmethod
    def spread(cls, n_variants: int, n_planted: int | None = None, max_eps: float = 0.3, mult: int = 2_654_435_761, offset: int = 12_345):
        """Copies whose sources are scattered over the whole data set: j = (k mult + offset) mod half with the golden-ratio prime
        2^32 / phi - coprime with every half it does not divide, and far from a small multiple of any half in sight, so that
        neighbouring copies have sources far apart."""
        import math
        half = n_variants // 2
        assert half < 2 or math.gcd(mult, half) == 1, (mult, half)
        return cls(half if n_planted is None else n_planted, half, mult, offset % max(half, 1), max_eps)

    @classmethod
    def near(cls, n_variants: int, distance: int, n_planted: int | None = None, max_eps: float = 0.3):
        """Copies at a fixed odd distance in front of their sources (window runs): copy 2k + 1 <- source 2k + 1 + distance."""
        assert distance % 2 == 1
        half = n_variants // 2
        return cls(half if n_planted is None else n_planted, half, 1, (distance + 1) // 2, max_eps)


class _LaunchStat(C.Structure):      # twk_hip_launch_stat
    _fields_ = [("ms", C.c_double), ("shader_mhz", C.c_double), ("xcd_finish_spread_us", C.c_double), ("row_pairs", C.c_uint64),
                ("candidates", C.c_uint64), ("words_per_row", C.c_uint32), ("kind", C.c_uint32), ("outlier", C.c_uint32), ("_pad", C.c_uint32)]


@dataclass
class Filters:
    """twk_ld_settings filter defaults (reference lib/core.cpp:304)."""
    minR2: float = 0.1
    maxR2: float = 100.0
    minDprime: float = 0.0
    maxDprime: float = 100.0
    minP: float = 1.0

    def _c(self) -> _Filters:
        return _Filters(self.minR2, self.maxR2, self.minDprime, self.maxDprime, self.minP)


class HipError(RuntimeError):
    def __init__(self, code: int, what: str, detail: str = ""):
        super().__init__(f"{what}: error {code}" + (f" ({detail})" if detail else ""))
        self.code = code


_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64)
_lib = None


def load_library() -> C.CDLL:
    """Load lib/libtwk_hip.so (built by `make hip` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built: run `make hip` (no CPU fallback exists)")
    lib = C.CDLL(LIB_PATH)
    p = C.c_void_p
    lib.twk_hip_abi_version.restype = C.c_int
    if lib.twk_hip_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {lib.twk_hip_abi_version()}, this binding was written for {ABI_VERSION}: rebuild (`make hip`)")
    lib.twk_hip_device_count.restype = C.c_int
    lib.twk_hip_strerror.restype = C.c_char_p
    lib.twk_hip_strerror.argtypes = [C.c_int]
    lib.twk_hip_last_error.restype = C.c_char_p
    lib.twk_hip_last_error.argtypes = [p]
    lib.twk_hip_ctx_create.argtypes = [C.c_int, C.POINTER(p)]
    lib.twk_hip_ctx_destroy.argtypes = [p]
    lib.twk_hip_set_problem.argtypes = [p, C.c_uint32, C.c_uint32]
    lib.twk_hip_upload_bitvectors.argtypes = [p, C.c_uint32, C.c_uint32, p, p, C.c_size_t, p]
    lib.twk_hip_upload_rle.argtypes = [p, C.c_uint32, C.c_uint32, p, C.c_size_t, p, p]
    lib.twk_hip_download_bitvectors.argtypes = [p, C.c_uint32, C.c_uint32, p, p, C.c_size_t]
    lib.twk_hip_host_alloc.argtypes = [C.c_size_t, C.POINTER(p)]
    lib.twk_hip_host_free.argtypes = [p]
    lib.twk_hip_generate_synthetic.argtypes = [p, C.c_uint64]
    lib.twk_hip_generate_synthetic_range.argtypes = [p, C.c_uint64, C.c_uint32]
    lib.twk_synth_bitvector.restype = C.c_uint32
    lib.twk_synth_bitvector.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, p]
    lib.twk_hip_generate_synthetic_planted.argtypes = [p, C.c_uint64, C.c_uint32, C.POINTER(Plant)]
    lib.twk_synth_planted_bitvector.restype = C.c_uint32
    lib.twk_synth_planted_bitvector.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(Plant), p]
    lib.twk_synth_plant_source.argtypes = [C.c_uint64, C.POINTER(Plant), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    lib.twk_hip_get_marginals.argtypes = [p, p, p, p, p]
    lib.twk_hip_count_tile.argtypes = [p, C.c_int, C.POINTER(_Tile), p]
    lib.twk_hip_ld_tile.argtypes = [p, C.c_int, C.POINTER(_Tile), C.POINTER(_Filters), p, C.c_uint64,
                                    C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.twk_hip_ld_all.argtypes = [p, C.c_int, C.POINTER(_Filters), C.c_uint32, C.c_uint32, C.c_uint32,
                                   C.c_int32, C.c_uint32, _SINK, p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.twk_hip_ld_region.argtypes = [p, C.c_int, C.POINTER(_Filters), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32, _SINK, p,
                                      C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.twk_hip_shard_rows.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32, C.c_uint32,
                                       C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    lib.twk_hip_plan_region.argtypes = [p, p, p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.c_int32, C.c_uint32, p, C.c_uint32, p, p, p, p, p, p, p]
    lib.twk_hip_launch_log.argtypes = [p, p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    lib.twk_hip_fisher_exact.argtypes = [p, p, C.c_uint64, p, C.c_int32, C.POINTER(C.c_float)]
    lib.twk_hip_set_device_sink.argtypes = [p, C.c_int]
    lib.twk_hip_device_records.argtypes = [p, C.POINTER(p), C.POINTER(C.c_uint64)]
    lib.twk_hip_set_option.argtypes = [p, C.c_char_p, C.c_int64]
    lib.twk_hip_get_option.argtypes = [p, C.c_char_p, C.POINTER(C.c_int64)]
    lib.twk_hip_gather_records.argtypes = [C.POINTER(p), C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    lib.twk_hip_gather_backend.restype = C.c_char_p
    lib.twk_hip_drain_device_sink.argtypes = [p, _SINK, p, C.POINTER(C.c_uint64)]
    lib.twk_hip_option_describe.argtypes = [C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_char_p)]
    lib.twk_hip_timing_reset.argtypes = [p]
    lib.twk_hip_timing_get.argtypes = [p, C.POINTER(_Timing)]
    _lib = lib
    return lib


def option_table():
    """The engine's option table (twk_hip_option_describe: needs no device) -> list of (key, default, lowest, highest, meaning)."""
    lib = load_library()
    out, i = [], 0
    while True:
        key, doc = C.c_char_p(), C.c_char_p()
        d, lo, hi = C.c_int64(), C.c_int64(), C.c_int64()
        if lib.twk_hip_option_describe(i, C.byref(key), C.byref(d), C.byref(lo), C.byref(hi), C.byref(doc)) != 0:
            return out
        out.append((key.value.decode(), d.value, lo.value, hi.value, doc.value.decode()))
        i += 1


def option_table_markdown() -> str:
    """... as the table INTEGRATION.md 1 carries between its `<!-- engine options -->` markers (tests/test_docs_consistency.py)."""
    def num(x):
        for k in (20, 30, 32, 40):
            if x == 1 << k:
                return f"2^{k}"
        return str(x)
    rows = ["| key | default | range | meaning |", "|---|---|---|---|"]
    rows += [f"| `{k}` | {num(d)} | {num(lo)} .. {num(hi)} | {doc} |" for k, d, lo, hi, doc in option_table()]
    return "\n".join(rows)


GATHER_SELF_LOOP = 1


def gather_backend() -> str:
    """What twk_hip_gather_records moves records with: "rccl <version>" or "unavailable (<why>)"."""
    return load_library().twk_hip_gather_backend().decode()


def gather_records(engines, dst: int = 0, self_loop: bool = False):
    """One process, one HipLd per GPU, each with its device sink on: gather their survivors into engines[dst]'s sink over RCCL
    (twk_hip_gather_records) -> (records now held by engines[dst], milliseconds of the transfers)."""
    lib = load_library()
    arr = (C.c_void_p * len(engines))(*[e._ctx for e in engines])
    n, ms = C.c_uint64(0), C.c_double(0.0)
    rc = lib.twk_hip_gather_records(arr, len(engines), dst, GATHER_SELF_LOOP if self_loop else 0, C.byref(n), C.byref(ms))
    engines[dst]._check(rc, "twk_hip_gather_records")
    return n.value, ms.value


def device_count() -> int:
    return load_library().twk_hip_device_count()


def shard_rows(n_variants: int, part: int, n_parts: int, n_cols: int | None = None, triangle: bool = True):
    """Row band [r0, r1) and pair count of shard `part` (the partition ld_all uses). Host-only."""
    r0, r1, n = C.c_uint32(), C.c_uint32(), C.c_uint64()
    rc = load_library().twk_hip_shard_rows(n_variants, n_variants if n_cols is None else n_cols, int(triangle), part,
                                           n_parts, C.byref(r0), C.byref(r1), C.byref(n))
    if rc != 0:
        raise HipError(rc, "twk_hip_shard_rows")
    return r0.value, r1.value, n.value


def words64(n_samples: int) -> int:
    return (2 * n_samples + 63) // 64


def synth_bitvector(seed: int, n_samples: int, v: int, plant: "Plant | None" = None) -> tuple[np.ndarray, int]:
    """Host twin of the device generator (twk_synth_bitvector / twk_synth_planted_bitvector)."""
    out = np.zeros(words64(n_samples), dtype=np.uint64)
    ac = load_library().twk_synth_planted_bitvector(seed, n_samples, v, None if plant is None else C.byref(plant), out.ctypes.data)
    return out, int(ac)


def plant_source(seed: int, plant: "Plant", v: int):
    """-> (source variant, flip probability) when global variant v is a planted copy, else None (twk_synth_plant_source)."""
    src, eps = C.c_uint32(0), C.c_double(0.0)
    if load_library().twk_synth_plant_source(seed, C.byref(plant), v, C.byref(src), C.byref(eps)):
        return int(src.value), float(eps.value)
    return None


class HipLd:
    """One engine context on one GPU (twk_hip_ctx)."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        rc = self._lib.twk_hip_ctx_create(device, C.byref(self._ctx))
        if rc != 0:
            raise HipError(rc, "twk_hip_ctx_create", self._lib.twk_hip_strerror(rc).decode())
        self.n_samples = 0
        self.n_variants = 0
        self.device = device
        self._sink_on = False
        self._defaults = {}

    # ---- switches (twk_hip_set_option: the library reads no environment variable) ----
    def get_option(self, key: str) -> int:
        v = C.c_int64(0)
        self._check(self._lib.twk_hip_get_option(self._ctx, key.encode(), C.byref(v)), f"twk_hip_get_option({key})")
        return int(v.value)

    def set_option(self, key: str, value: int):
        self._defaults.setdefault(key, self.get_option(key))
        self._check(self._lib.twk_hip_set_option(self._ctx, key.encode(), int(value)), f"twk_hip_set_option({key})")

    def unset_option(self, key: str):
        """Back to the value the context was created with."""
        if key in self._defaults:
            self._check(self._lib.twk_hip_set_option(self._ctx, key.encode(), self._defaults[key]), f"twk_hip_set_option({key})")

    def close(self):
        if self._ctx:
            self._lib.twk_hip_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            detail = self._lib.twk_hip_last_error(self._ctx).decode() or self._lib.twk_hip_strerror(rc).decode()
            raise HipError(rc, what, detail)

    # ---- input ----
    def set_problem(self, n_samples: int, n_variants: int):
        self._check(self._lib.twk_hip_set_problem(self._ctx, n_samples, n_variants), "twk_hip_set_problem")
        self.n_samples, self.n_variants = n_samples, n_variants

    def upload(self, data: np.ndarray, meta: np.ndarray, mask: np.ndarray | None = None, first: int = 0):
        """data/mask: uint64 [count, stride64] reference-layout bitvectors; meta: META_DTYPE[count]."""
        data = np.ascontiguousarray(data, dtype=np.uint64)
        meta = np.ascontiguousarray(meta, dtype=META_DTYPE)
        assert data.ndim == 2 and data.shape[0] == meta.shape[0]
        mptr = None
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint64)
            assert mask.shape == data.shape
            mptr = mask.ctypes.data
        self._check(self._lib.twk_hip_upload_bitvectors(self._ctx, first, data.shape[0], data.ctypes.data, mptr,
                                                        data.shape[1], meta.ctypes.data), "twk_hip_upload_bitvectors")

    def upload_rle(self, run_bytes: np.ndarray, desc: np.ndarray, meta: np.ndarray, first: int = 0):
        """Run-length genotype words as stored in a .twk block (uint8 buffer) + RLE_DESC_DTYPE[count]:
        expanded to bitvector + mask by a HIP kernel (twk_hip_upload_rle)."""
        run_bytes = np.ascontiguousarray(run_bytes, dtype=np.uint8)
        desc = np.ascontiguousarray(desc, dtype=RLE_DESC_DTYPE)
        meta = np.ascontiguousarray(meta, dtype=META_DTYPE)
        assert desc.shape == meta.shape
        self._check(self._lib.twk_hip_upload_rle(self._ctx, first, len(desc), run_bytes.ctypes.data, run_bytes.size,
                                                 desc.ctypes.data, meta.ctypes.data), "twk_hip_upload_rle")

    def download(self, first: int = 0, count: int | None = None):
        """-> (data, mask) uint64 [count, words64] in the reference layout, as held on the device."""
        count = self.n_variants - first if count is None else count
        w = words64(self.n_samples)
        data = np.zeros((count, w), dtype=np.uint64)
        mask = np.zeros((count, w), dtype=np.uint64)
        self._check(self._lib.twk_hip_download_bitvectors(self._ctx, first, count, data.ctypes.data, mask.ctypes.data, w),
                    "twk_hip_download_bitvectors")
        return data, mask

    def generate_synthetic(self, seed: int = 42, first_variant: int = 0, plant: "Plant | None" = None):
        """Synthetic benchmark input for global variants [first_variant, first_variant + n_variants); plant: LD pairs planted in it."""
        self._check(self._lib.twk_hip_generate_synthetic_planted(self._ctx, seed, first_variant, None if plant is None else C.byref(plant)),
                    "twk_hip_generate_synthetic_planted")

    def marginals(self):
        M = self.n_variants
        ac, het, hom, miss = (np.zeros(M, dtype=np.uint32) for _ in range(4))
        self._check(self._lib.twk_hip_get_marginals(self._ctx, ac.ctypes.data, het.ctypes.data, hom.ctypes.data,
                                                    miss.ctypes.data), "twk_hip_get_marginals")
        return ac, het, hom, miss

    # ---- compute ----
    @staticmethod
    def _tile(a0, nA, b0, nB, diag, window=0, l_window=0) -> _Tile:
        return _Tile(a0, nA, b0, nB, int(bool(diag)), int(window), l_window, 0)

    def count_tile(self, mode: int, a0: int, nA: int, b0: int, nB: int, diag: bool = False) -> np.ndarray:
        ncell = 4 if mode == MODE_PHASED else 9
        out = np.zeros((nA, nB, ncell), dtype=np.uint64)
        t = self._tile(a0, nA, b0, nB, diag)
        self._check(self._lib.twk_hip_count_tile(self._ctx, mode, C.byref(t), out.ctypes.data), "twk_hip_count_tile")
        return out

    def ld_tile(self, mode: int, a0: int, nA: int, b0: int, nB: int, diag: bool, filters: Filters,
                capacity: int | None = None, window: int = 0, l_window: int = 0):
        cap = capacity if capacity is not None else nA * nB
        out = np.zeros(max(cap, 1), dtype=RECORD_DTYPE)
        n, npairs = C.c_uint64(0), C.c_uint64(0)
        t = self._tile(a0, nA, b0, nB, diag, window, l_window)
        f = filters._c()
        rc = self._lib.twk_hip_ld_tile(self._ctx, mode, C.byref(t), C.byref(f), out.ctypes.data, cap,
                                       C.byref(n), C.byref(npairs))
        if rc == E_OVERFLOW:
            raise HipError(rc, "twk_hip_ld_tile", f"need capacity {n.value}")
        self._check(rc, "twk_hip_ld_tile")
        return out[: n.value].copy(), npairs.value

    def ld_all(self, mode: int, filters: Filters, part: int = 0, n_parts: int = 1, tile_variants: int = 0,
               window: int = 0, l_window: int = 0, collect: bool = True):
        """All-vs-all over shard `part` of `n_parts`. Returns (records, n_pairs, n_records).
        `window` carries the TWK_HIP_OPT_* bits unchanged (OPT_WINDOW = 1, OPT_KEEP_LOW_AC = 2)."""
        chunks = []

        def sink(_user, recs, n):
            if collect and n:
                buf = (C.c_char * (n * RECORD_DTYPE.itemsize)).from_address(recs)
                chunks.append(np.frombuffer(buf, dtype=RECORD_DTYPE).copy())
            return 0

        cb = _SINK(sink)
        npairs, nrec = C.c_uint64(0), C.c_uint64(0)
        f = filters._c()
        # (with the device sink on the records stay in HBM: the returned array is empty, see device_records_tensor)
        self._check(self._lib.twk_hip_ld_all(self._ctx, mode, C.byref(f), part, n_parts, tile_variants,
                                             int(window), l_window, cb, None, C.byref(npairs), C.byref(nrec)),
                    "twk_hip_ld_all")
        recs = np.concatenate(chunks) if chunks else np.zeros(0, dtype=RECORD_DTYPE)
        return recs, npairs.value, nrec.value

    def ld_region(self, mode: int, filters: Filters, a0: int, nA: int, b0: int, nB: int, triangle: bool,
                  part: int = 0, n_parts: int = 1, tile_variants: int = 0, window: int = 0, l_window: int = 0,
                  collect: bool = True):
        """LD over rows [a0,a0+nA) x cols [b0,b0+nB) (triangle: col > row, nB >= nA). Returns (records, n_pairs, n_records)."""
        chunks = []

        def sink(_user, recs, n):
            if collect and n:
                buf = (C.c_char * (n * RECORD_DTYPE.itemsize)).from_address(recs)
                chunks.append(np.frombuffer(buf, dtype=RECORD_DTYPE).copy())
            return 0

        cb = _SINK(sink)
        npairs, nrec = C.c_uint64(0), C.c_uint64(0)
        f = filters._c()
        self._check(self._lib.twk_hip_ld_region(self._ctx, mode, C.byref(f), a0, nA, b0, nB, int(bool(triangle)), part,
                                                n_parts, tile_variants, int(window), l_window, cb, None,
                                                C.byref(npairs), C.byref(nrec)), "twk_hip_ld_region")
        recs = np.concatenate(chunks) if chunks else np.zeros(0, dtype=RECORD_DTYPE)
        return recs, npairs.value, nrec.value

    def fisher_exact(self, tables: np.ndarray, ordered: bool = True):
        """Two-sided Fisher P of int32 tables [n, 4] = (n11, n12, n21, n22) through the engine's Fisher kernels
        (twk_hip_fisher_exact): as the engine runs them, the walks in the order of their length, or (ordered=False) in
        the order given - same P.  -> (P float64[n], kernel milliseconds)."""
        t = np.ascontiguousarray(tables, dtype=np.int32).reshape(-1, 4)
        out = np.zeros(len(t), dtype=np.float64)
        ms = C.c_float(0)
        self._check(self._lib.twk_hip_fisher_exact(self._ctx, t.ctypes.data, len(t), out.ctypes.data, 0 if ordered else 1,
                                                   C.byref(ms)), "twk_hip_fisher_exact")
        return out, float(ms.value)

    # ---- multi-GPU: survivors stay in HBM until the gather ----
    def set_device_sink(self, on: bool = True):
        """Region / all-vs-all calls keep their survivors on the device (twk_hip_set_device_sink); empties the buffer."""
        self._check(self._lib.twk_hip_set_device_sink(self._ctx, int(bool(on))), "twk_hip_set_device_sink")
        self._sink_on = bool(on)

    def device_records(self):
        """-> (device pointer, n records) of what the device sink holds (twk_hip_device_records)."""
        ptr, n = C.c_void_p(), C.c_uint64(0)
        self._check(self._lib.twk_hip_device_records(self._ctx, C.byref(ptr), C.byref(n)), "twk_hip_device_records")
        return (ptr.value or 0), n.value

    def drain_device_sink(self) -> np.ndarray:
        """The device sink's records on the host, in the order they lie in the sink; the sink is empty afterwards (twk_hip_drain_device_sink)."""
        chunks = []

        def sink(_user, recs, n):
            buf = (C.c_char * (n * RECORD_DTYPE.itemsize)).from_address(recs)
            chunks.append(np.frombuffer(buf, dtype=RECORD_DTYPE).copy())
            return 0

        cb = _SINK(sink)
        n = C.c_uint64(0)
        self._check(self._lib.twk_hip_drain_device_sink(self._ctx, cb, None, C.byref(n)), "twk_hip_drain_device_sink")
        out = np.concatenate(chunks) if chunks else np.zeros(0, dtype=RECORD_DTYPE)
        assert len(out) == n.value
        return out

    def device_records_tensor(self):
        """The device sink's records as a torch uint8 tensor [n * 104] that aliases the engine's HBM buffer (no copy):
        what tomahawk_amd.dist.gather_records sends over RCCL.  Valid until the next compute call."""
        import torch
        ptr, n = self.device_records()
        dev = torch.device("cuda", self.device)          # the context's own device, whatever torch's current one is
        if n == 0:
            return torch.empty(0, dtype=torch.uint8, device=dev)

        class _Span:          # the CUDA array interface is how torch wraps foreign device memory without owning it
            __cuda_array_interface__ = {"shape": (n * RECORD_DTYPE.itemsize,), "typestr": "|u1", "data": (ptr, False), "version": 2}
        return torch.as_tensor(_Span(), device=dev)

    # ---- measurement ----
    def timing_reset(self):
        self._check(self._lib.twk_hip_timing_reset(self._ctx), "twk_hip_timing_reset")

    def launch_log(self, capacity: int = 4096):
        """The count launches since the last timing_reset, oldest first (twk_hip_launch_log) -> (list of dicts, launches seen)."""
        buf = (_LaunchStat * capacity)()
        n, total = C.c_uint32(0), C.c_uint64(0)
        self._check(self._lib.twk_hip_launch_log(self._ctx, buf, capacity, C.byref(n), C.byref(total)), "twk_hip_launch_log")
        return [{k: getattr(buf[i], k) for k, _ in _LaunchStat._fields_ if k != "_pad"} for i in range(n.value)], total.value

    def timing(self) -> dict:
        t = _Timing()
        self._check(self._lib.twk_hip_timing_get(self._ctx, C.byref(t)), "twk_hip_timing_get")
        return {k: getattr(t, k) for k, _ in _Timing._fields_}
