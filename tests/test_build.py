"""Checks on the device code as compiled (CPU only: hipcc cross-compiles gfx950 without a GPU)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def shift64_amount_in_the_last_vgpr(asm: str):
    """-> [(kernel, instruction)]: 64-bit shifts (v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64) whose shift amount sits in the
    last VGPR the kernel's waves own (VGPRs are allocated in blocks of 8)."""
    counts = dict(re.findall(r"\.name:\s+(\S+)\n(?:(?!\.name:).*\n)*?\s+\.vgpr_count:\s+(\d+)", asm))
    hits, func = [], None
    for line in asm.split("\n"):
        t = line.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            func = m.group(1)
            continue
        m = re.match(r"^(v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64)\s+v\[\d+:\d+\],\s*v(\d+),", t)
        if m and func in counts:
            owned = -(-int(counts[func]) // 8) * 8
            if int(m.group(2)) == owned - 1:
                hits.append((func, t))
    return hits, counts


def test_scanner_finds_the_pattern_it_is_for():
    asm = """
_ZN1a1kEv:
	v_lshlrev_b64 v[24:25], v31, v[24:25]
	v_lshlrev_b64 v[22:23], v15, v[16:17]
_ZN1a1gEv:
	v_lshlrev_b64 v[18:19], v23, v[18:19]
amdhsa.kernels:
  - .name:           _ZN1a1kEv
    .sgpr_count:     10
    .vgpr_count:     32
  - .name:           _ZN1a1gEv
    .sgpr_count:     10
    .vgpr_count:     26
"""
    hits, counts = shift64_amount_in_the_last_vgpr(asm)
    assert counts == {"_ZN1a1kEv": "32", "_ZN1a1gEv": "26"}
    assert hits == [("_ZN1a1kEv", "v_lshlrev_b64 v[24:25], v31, v[24:25]")]      # v15 of 32 and v23 of 26 (-> 32 owned) are not the last


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_kernel_shifts_64_bits_by_an_amount_in_its_last_vgpr(tmp_path):
    """On the MI355X boxes of this pool a 64-bit shift whose amount the register allocator puts into the last VGPR a wave owns
    returns wrong results (the "shift64 high register" erratum; LLVM works around it for gfx90a only).  It cost round 4 an
    afternoon: an unrolled unphased probe loop had its amount in v31 of 32 (v47 of 48) and gave one group of rows per run wrong
    counts, while every form that kept the amount elsewhere was right (ld_list.hip.h; stand-alone reproducer:
    csrc/tools/shift64_probe.hip, profiles/r04_shift64_probe.txt - 88 % of a million lanes wrong).  The kernels that shifted
    64-bit counters by a variable now use 32-bit ones; this test compiles the library's device code the way `make hip` does
    and looks at every kernel for the pattern, so that a later change (or a later compiler) cannot bring it back unseen."""
    out = str(tmp_path / "twk_hip.s")
    make = open(os.path.join(ROOT, "Makefile")).read()
    flags = re.search(r"^HIPFLAGS\s*:=\s*(.*)$", make, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    flags = [f for f in flags if f not in ("-fPIC",)]
    r = subprocess.run([HIPCC] + flags + ["-Iinclude", "-S", "--cuda-device-only", "-o", out, "tomahawk_amd/csrc/hip/twk_hip.hip"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env={k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "ASAN_OPTIONS", "UBSAN_OPTIONS")})
    assert r.returncode == 0, r.stderr[-2000:]
    asm = open(out).read()
    hits, counts = shift64_amount_in_the_last_vgpr(asm)
    assert len(counts) > 20 and any("k_count_list_t" in k for k in counts)         # the scan saw the library's kernels
    assert not hits, hits


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_count_kernels_use_no_scratch_memory(tmp_path):
    """Every count kernel (plain, three-product, fused) must keep its loop state in registers: a spilled per-lane offset is
    reloaded from scratch *inside* the K loop (round 5: an epilogue that grew by a dozen registers cost the fused phased
    kernel 16 spills and 10 % of its rate before the offsets were recomputed per unit, profiles/r05_fused_epilogue.txt)."""
    out = str(tmp_path / "twk_hip.s")
    make = open(os.path.join(ROOT, "Makefile")).read()
    flags = re.search(r"^HIPFLAGS\s*:=\s*(.*)$", make, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    flags = [f for f in flags if f not in ("-fPIC",)]
    r = subprocess.run([HIPCC] + flags + ["-Iinclude", "-S", "--cuda-device-only", "-o", out, "tomahawk_amd/csrc/hip/twk_hip.hip"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env={k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "ASAN_OPTIONS", "UBSAN_OPTIONS")})
    assert r.returncode == 0, r.stderr[-2000:]
    asm = open(out).read()
    seen = 0
    for name, body in re.findall(r"\.name:\s+(\S+)\n((?:(?!\s*\.name:).*\n)*)", asm):
        if "k_count" not in name:
            continue
        seen += 1
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", body).group(1))
        spills = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", body).group(1))
        vgprs = int(re.search(r"\.vgpr_count:\s+(\d+)", body).group(1))
        assert scratch == 0 and spills == 0, (name, scratch, spills)
        assert vgprs <= 128, (name, vgprs)            # four waves per SIMD
    assert seen >= 5
