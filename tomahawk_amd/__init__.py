"""MI355X-native pairwise-LD engine, drop-in for the `tomahawk calc` hot path.

The product is native code: HIP kernels behind the C ABI of ``include/twk_hip.h``
(``lib/libtwk_hip.so``) and the C++ host side (``lib/libtomahawk_amd.so``,
``bin/tomahawk``) that mirrors the reference's ``tomahawk::twk_ld`` /
``tomahawk calc``.  This Python package only binds those libraries with ctypes
for tests, benchmarks and scripting; it contains no compute and no CPU fallback.
"""
from .hip import (HipLd, HipError, Filters, RECORD_DTYPE, MODE_PHASED, MODE_UNPHASED, MODE_AUTO,
                  OPT_WINDOW, OPT_KEEP_LOW_AC, OPT_REF_COMPAT, OPT_R2_SCREEN, RLE_DESC_DTYPE, META_DTYPE,
                  device_count, load_library, synth_bitvector, shard_rows, plan_region, TILE_DTYPE, Plant, plant_source, gather_records, gather_backend)

__all__ = ["HipLd", "HipError", "Filters", "RECORD_DTYPE", "MODE_PHASED", "MODE_UNPHASED",
           "MODE_AUTO", "OPT_WINDOW", "OPT_KEEP_LOW_AC", "OPT_REF_COMPAT", "OPT_R2_SCREEN", "RLE_DESC_DTYPE", "META_DTYPE", "device_count", "load_library", "synth_bitvector", "shard_rows", "plan_region", "TILE_DTYPE", "Plant", "plant_source", "gather_records", "gather_backend"]
