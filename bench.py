#!/usr/bin/env python3
"""Headline benchmark: all-vs-all pairwise LD, variant pairs per second (BASELINE.json).

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[2] -- 1,000,000 diploid samples x 50,000 biallelic
variants, all-vs-all *unphased* genotype LD (`calc -u`, default filters r2 >= 0.1), synthetic iid
genotypes generated directly in HBM (bit-identical host twin feeds the CPU baseline).
A step is one pass over the whole upper triangle: 1,249,975,000 variant pairs.  With N GPUs the
triangle is cut into equal-area row bands, one per rank (no data-path collective); inside the timed
region the survivors are gathered to rank 0 over RCCL (counts all-gather + grouped send/recv of exact
sizes) and rank 0 packs them into a real .two file; `value` is total pairs / max-over-ranks time:
total work is fixed, so "scaling" is "strong".

The JSON line also carries
  roofline     the dominant kernel (twk::k_count3_list_t: the three-product form of the unphased contraction - HH and
               S = QH + HQ + 2 QQ per pair, all UnphasedMath's r2 screen reads; the four products of the pairs that pass are
               recounted; twk::k_count_list_t where that form does not apply), timed live with HIP events on the engine's
               own stream.  `frac` counts the lane-ops the kernel EXECUTED (v_and or v_bitop3 + v_bcnt per product; no v_or since
               round 6); `algorithmic_frac` SURVEY 8(d)'s four products per unphased pair over the same time.  The kernel tiles 128 x 128 plane rows through LDS, so each streamed row is
               reused 128x and HBM is not what binds it: the binding unit is the VALU (v_and_b32 +
               v_bcnt_u32_b32 per 32-bit word pair, no MFMA as the north star requires).  bound="valu":
               achieved = algorithmic lane-ops (2*ceil(2N/32) per pair phased,
               8*ceil(N/32) unphased) x variant pairs of the launches / kernel time, against the
               SIMD lane peak 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz = 78.6 T lane-ops/s.
               `and_bcnt_ceiling_frac` prices the same work against what the two instructions can
               issue at best (v_and_b32 2 cycles + v_bcnt_u32_b32 4 cycles per wave64: 2.62e13 word
               pairs/s; microbenchmarks under profiles/).  `hbm_algorithmic` keeps SURVEY 8(d)'s
               HBM-read accounting (N/4 bytes per pair against 8 TB/s; > 1 by construction because
               of the LDS reuse).  `traffic` (bytes per launch between the L2s and memory): at N=1 two rocprofv3
               counter passes over one step of the same workload are run as child processes after the timed
               region (FETCH_SIZE, WRITE_SIZE; --no-traffic skips them, then it is null).
  cpu_baseline the compiled reference (oracle/_ref, SSE4.2) on this box's host cores, on the first
               M_s variants of the same synthetic input (rank 0, N=1 only).
  e2e          (N=1 only, after the timed region like cpu_baseline) `tomahawk calc`, default mode, from a cohort-shaped
               .twk of the headline size written to /tmp: load (pread + zstd + device run-length inflate), r2 screen,
               count / math kernels, device sort, .two writer - the parts of the path the survivor-free headline step
               does not exercise, under the driver's clock.
  extra        (N=1, cfg3 only, after the timed region) the other regimes the repository makes claims about, under the same
               clock: "cfg2" (BASELINE configs[1], 20 steps), "cfg5_shard" (configs[4], shard 3 of 8 emulated on this GPU, one
               step), "e2e_u" (`tomahawk calc -u` from the same cohort .twk as "e2e") and "kg" - the reference's only
               published workloads (docs/tutorial.md:177-199, 252-253): `calc -p` r2 >= 0.1 over all 141.2 G pairs of a
               cohort-shaped 2,504 x 531,500 .twk, and `calc -p -w 4000000` on the same file.  Each with pairs/s, the dominant
               kernel, its average launch, its own roofline fraction, records, load / compute + write split, and the shader
               clock the count kernel's blocks really ran at (roofline.shader_mhz likewise).  --no-extra skips them.
For N > 1 the line also carries per_rank_ms (each rank's compute time per step), gather_ms / write_ms (rank 0: the
gather of the survivors incl. waiting for the slowest rank; packing the .two) and ranks_seen (an all-gather of the rank
ids over the group that carried the gather), gather_bytes / gather_GBps (record bytes rank 0 received per step, and their
rate over the transfers alone - from the moment every rank had arrived).
"""
import argparse
import ctypes
import datetime
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (n_samples, n_variants, mode)
    "cfg3": (1_000_000, 50_000, "unphased"),   # BASELINE.json configs[2]: the metric's workload
    "cfg2": (100_000, 10_000, "phased"),       # configs[1]
    "cfg1": (1_000, 1_000, "unphased"),        # configs[0] (plumbing size)
    # configs[4]: windowed (+-500 kb at 100 bp spacing = 5,000 partners per variant), Fisher P <= 1e-6.
    # 500 GB of bitvectors: every rank holds only its band of rows plus the halo its window reaches.
    "cfg5": (10_000_000, 200_000, "unphased"),
}
NAMES = {"cfg1": "configs[0]", "cfg2": "configs[1]", "cfg3": "configs[2]", "cfg5": "configs[4]"}
WINDOW_BP = {"cfg5": 500_000}
MIN_P = {"cfg5": 1e-6}
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_LANE_PEAK = 256 * 4 * 32 * 2.4e9         # SIMD-32 lane-ops/s (MI355X_MICROARCH.md: v_fma_f32 wave64 = 2 cycles)
VALU_PAIR_PEAK = 256 * 4 * 64 / 6.0 * 2.4e9   # and(2 cyc)+bcnt(4 cyc) per wave64 word pair, 2.4 GHz


def executed_work(tm):
    """What the count launches issued, from the engine's timing: AND+popcount products (one per word of a plane-row pair; three for
    every four of them in the three-product form of UnphasedMath's contraction, whose Q & (H | Q) terms are one v_bitop3_b32 each
    since round 6: no v_or any more) -> (products, other VALU ops of the loop (0), name of the kernel, name of the form)."""
    words = tm["words_per_row"]
    four = (tm["row_pairs"] - tm["three_row_pairs"]) * words
    three = tm["three_row_pairs"] * words * 0.75
    kernel = "twk::k_count_list_t"
    form = "four products per pair (HH, HQ, QH, QQ)"
    if tm["three_launches"]:
        kernel = "twk::k_count3_list_t" if not tm["fused_launches"] else "twk::k_count3_screen_unphased_t"
        form = "three products per pair (HH and S = QH + HQ + 2 QQ; the four products of screened-in pairs recounted)"
        if tm["three_launches"] < tm["count_launches"]:
            form += f" in {tm['three_launches']} of {tm['count_launches']} launches"
    return four + three, 0.0, kernel, form


def launch_spread(eng):
    """The count launches of the timed region one by one (twk_hip_launch_log): how far the slowest launch's time per unit of work lies
    above the median's - a run that silently took 1.7 x as long (round 4 saw such runs) shows here - and what the engine's outlier watch
    flagged.  -> dict for the JSON line."""
    stats, seen = eng.launch_log()
    cost = [x["ms"] / (x["row_pairs"] * x["words_per_row"] * (0.75 if x["kind"] in (1, 4) else 1.0)) for x in stats if x["row_pairs"] and x["ms"] >= 0.3]
    if not cost:
        return {"launches": seen}
    srt = sorted(cost)
    med = srt[len(srt) // 2]
    ms = sorted(x["ms"] for x in stats)
    # the same in shader cycles (cost x the clock the launch's blocks ran at): a launch that is slow in milliseconds and ordinary in cycles ran
    # at a low clock - the one cause the watch has caught so far (profiles/r05_outliers.txt: 1.1-1.6 GHz for seconds, 1.47-1.76 x the time)
    cyc = sorted(x["ms"] * x["shader_mhz"] / (x["row_pairs"] * x["words_per_row"] * (0.75 if x["kind"] in (1, 4) else 1.0))
                 for x in stats if x["row_pairs"] and x["ms"] >= 0.3 and x["shader_mhz"])
    return {"launches": seen, "launch_ms_max": ms[-1], "launch_ms_median": ms[len(ms) // 2],
            "launch_cost_max_over_median": srt[-1] / med if med > 0 else None, "launch_cost_min_over_median": srt[0] / med if med > 0 else None,
            "launch_cycles_max_over_median": (cyc[-1] / cyc[len(cyc) // 2]) if cyc and cyc[len(cyc) // 2] > 0 else None,
            "outlier_launches": sum(1 for x in stats if x["outlier"]),
            "shader_mhz_min": min((x["shader_mhz"] for x in stats if x["shader_mhz"]), default=None),
            "xcd_finish_spread_us_max": max((x["xcd_finish_spread_us"] for x in stats), default=None),
            "note": "cost = launch time per AND+popcount issued (launches differ in size: diagonal tiles, band edges); outlier: > 1.4 x the median of its peers"}


def cpu_baseline(n_samples, mode, seed, log):
    """Time the reference's SSE4.2 calc path (oracle/_ref) on a bounded sample of the same workload."""
    from oracle import oracle as O
    from tomahawk_amd import hostlib
    # 32 threads maximise the reference on the 256-thread GPU-box host (tests/sweeps/cpu_baseline_threads.py,
    # profiles/r02_cpu_baseline_threads.txt: 223 k pairs/s at 32 threads, 203 k at 64, 187 k at 128, 137 k at 256 -
    # its block-pair ticket is a spinlock and its output path a second one - and, found in round 4, the container has a CFS
    # quota of 16 CPUs: what those thread counts share; `host_cpus_usable` in the result says so)
    cores = min(os.cpu_count() or 1, 32)
    if not O.have_ref():
        return None
    # ~15 s of CPU work: SURVEY 6: 5.4 G (unphased) / 52 G (phased) genotype-pairs/s/core
    per_core = (5.4e9 if mode == "unphased" else 52e9) / n_samples
    pairs_target = per_core * cores * 15.0
    m = int(min(1500, max(200, (2 * pairs_target) ** 0.5)))
    block = max(10, min(50, m // (4 * min(cores, 32)) or 10))
    twk = os.path.join(tempfile.gettempdir(), f"twk_bench_{n_samples}_{m}_{seed}_{mode}.twk")
    if not os.path.exists(twk):
        t0 = time.time()
        hostlib.write_synthetic_twk(twk + ".tmp", n_samples, m, seed=seed, phased=(mode == "phased"),
                                    block_size=block, n_threads=min(cores, 32))
        os.replace(twk + ".tmp", twk)
        log(f"cpu_baseline: wrote {twk} ({os.path.getsize(twk) / 1e6:.1f} MB) in {time.time() - t0:.1f}s")
    out = os.path.join(tempfile.gettempdir(), f"twk_bench_{os.getpid()}.two")
    flag = "-u" if mode == "unphased" else "-p"
    t0 = time.time()
    r = subprocess.run([O.REF_BIN, "calc", "-i", twk, "-o", out, flag, "-t", str(cores)], capture_output=True, text=True)
    wall = time.time() - t0
    try:
        os.remove(out)
    except OSError:
        pass
    if r.returncode != 0:
        log("cpu_baseline: reference failed: " + r.stderr[-300:])
        return None
    pairs = m * (m - 1) // 2
    rate = None
    mo = re.search(r"\] ([0-9,]+) variants/s", r.stderr)     # ld_progress.h:94 ("variants" = pairs)
    if mo:
        rate = float(mo.group(1).replace(",", ""))
    if not rate:
        rate = pairs / wall
    return {"value": rate, "unit": "variant-pairs/s", "cores": cores, "kind": "reference",
            # CPUs the container may really use (affinity mask, CFS quota): `cores` threads share them
            "host_cpus_usable": hostlib.usable_cpus(),
            "sample": f"first {m} variants of the same synthetic input ({pairs} pairs, N={n_samples}, calc {flag} "
                      f"-t {cores}, {block} variants/block, SSE4.2 build of the reference, {wall:.1f}s wall)"}


def measure_traffic(config_args, log):
    """Memory-side traffic of the dominant kernel, per launch, from rocprofv3 counter passes over this very workload (one
    step, no warm-up; child processes started after the timed region): FETCH_SIZE and WRITE_SIZE in separate passes, as
    MI355X_MICROARCH.md prescribes, FETCH_SIZE doubled (gfx950: 128-byte requests tallied at 64 bytes).  FETCH_SIZE counts
    what the L2s request from the fabric - Infinity-Cache hits included - so this is an upper bound on HBM reads."""
    import csv
    import glob
    import shutil
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    got = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="twk_pmc_")
        cmd = [rocprof, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
               "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e", "--no-traffic", "--no-extra", "--no-planted"] + config_args
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=tempfile.gettempdir(), env=dict(os.environ, TMPDIR=tempfile.gettempdir()))
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                log(f"traffic: rocprofv3 --pmc {ctr} failed (rc {r.returncode}): {r.stderr[-200:]}")
                return None
            total, launches = 0.0, set()
            with open(files[0], newline="") as fh:
                for row in csv.DictReader(fh):
                    name = row.get("Kernel_Name") or row.get("kernel_name") or ""
                    if ", 1>(" in name:          # k_count3_list_t<8, 1>: the engine's half-millisecond density samples, not launches of the run
                        continue
                    if ("k_count_list_t" in name or "k_count_screen" in name or "k_count3" in name) and row.get("Counter_Name") == ctr:
                        total += float(row["Counter_Value"])
                        launches.add(row.get("Dispatch_Id") or row.get("dispatch_id"))
            if not launches:
                return None
            got[ctr] = (total * 1024.0, len(launches))       # KiB -> bytes
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch_b, n_f = got["FETCH_SIZE"]
    write_b, n_w = got["WRITE_SIZE"]
    per_launch = 2.0 * fetch_b / n_f + write_b / n_w
    log(f"traffic: FETCH_SIZE x2 {2 * fetch_b / 1e9:.1f} GB over {n_f} launches, WRITE_SIZE {write_b / 1e9:.1f} GB over {n_w}")
    return {"bytes_per_launch": per_launch, "read_bytes_per_launch": 2.0 * fetch_b / n_f, "write_bytes_per_launch": write_b / n_w, "launches": n_f}


def _secs(txt):
    """The CLI prints [Hh][Mm]S.sss"s" (twk_util.h elapsed_string)."""
    mo = re.fullmatch(r"(?:(\d+)h)?(?:(\d+)m)?([0-9.]+)s\.?", txt or "")
    return (int(mo.group(1) or 0) * 3600 + int(mo.group(2) or 0) * 60 + float(mo.group(3))) if mo else None


def run_cli(twk, flags, threads, out, keep_out=False):
    """One `tomahawk calc` run -> dict parsed from its log (None if it failed): wall, load, compute + write, pairs, records,
    the kernels' own times (HIP events, as the engine reports them) and the writer's share."""
    from tomahawk_amd import hostlib
    try:
        if not keep_out:           # (keep_out: `out` is a symlink to /dev/null, tests/sweeps/record_codec_floor.py)
            os.remove(out)         # (dropping a multi-gigabyte file of the previous run from the page cache is not part of this run)
    except OSError:
        pass
    t0 = time.time()
    r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", twk, "-o", out, "-t", str(threads)] + list(flags), capture_output=True, text=True)
    wall = time.time() - t0
    if r.returncode != 0:
        return {"error": r.stderr[-300:]}
    lg = r.stderr
    stamps = []
    for m in re.findall(r"^\[(\d{4}-\d\d-\d\d \d\d:\d\d:\d\d,\d{3})\]", lg, re.M):
        try:
            stamps.append(datetime.datetime.strptime(m, "%Y-%m-%d %H:%M:%S,%f").timestamp())
        except ValueError:
            pass
    load = re.search(r"Unpacked and uploaded .* variants\. (\S+)", lg)
    fin = re.search(r"Finished in (\S+)\. Variants: ([0-9,]+), genotypes: [0-9,]+, output: ([0-9,]+)", lg)
    eng = re.search(r"count kernel ([0-9.e+]+) ms in (\d+) launches \(([0-9.e+-]+) % [^)]*\), math kernels ([0-9.e+]+) ms", lg)
    mhz = re.search(r"its blocks ran at (\d+) MHz", lg)
    lst = re.search(r"carrier-list kernel ([0-9.e+]+) ms in (\d+) launches over ([0-9,]+) rare pairs", lg)
    prb = re.search(r"probe kernel ([0-9.e+]+) ms in (\d+) launches over ([0-9,]+) rare x common pairs", lg)
    fus = re.search(r"(\d+) launches fused count -> r2 screen, ([0-9,]+) candidate", lg)
    wri = re.search(r"the producer spent ([0-9.e+-]+) s handing", lg)
    pairs = int(fin.group(2).replace(",", "")) if fin else None
    cw = _secs(fin.group(1)) if fin else None
    res = {"wall_s": wall, "load_s": _secs(load.group(1)) if load else None, "compute_write_s": cw,
           "pairs": pairs, "records": int(fin.group(3).replace(",", "")) if fin else None,
           "pairs_per_s_compute_write": pairs / cw if pairs and cw else None,
           "pairs_per_s_wall": pairs / wall if pairs else None,
           "two_bytes": os.path.getsize(out) if os.path.exists(out) else None,
           "dominant_kernel": ("twk::k_count_screen_t / k_count_screen_unphased_t" if fus else "twk::k_count_list_t"),
           "count_kernel_ms": float(eng.group(1)) if eng else None, "count_launches": int(eng.group(2)) if eng else None,
           "avg_launch_ms": float(eng.group(1)) / max(int(eng.group(2)), 1) if eng else None,
           # the engine's own figure: word pairs it contracted / kernel time against 2.62e13 (and+bcnt); x 2/3 = of the lane peak
           "and_bcnt_ceiling_frac": float(eng.group(3)) / 100.0 if eng else None,
           "frac": float(eng.group(3)) / 100.0 * (2.0 / 3.0) if eng else None,
           "math_kernels_ms": float(eng.group(4)) if eng else None,
           "shader_mhz": int(mhz.group(1)) if mhz else None,
           "list_kernel_ms": float(lst.group(1)) if lst else None,
           "pairs_decided_by_carrier_lists": int(lst.group(3).replace(",", "")) if lst else None,
           "probe_kernel_ms": float(prb.group(1)) if prb else None,
           "pairs_decided_by_probes": int(prb.group(3).replace(",", "")) if prb else None,
           "fused_launches": int(fus.group(1)) if fus else 0,
           "producer_handover_s": float(wri.group(1)) if wri else None,
           "writer_line": next((l.split("] ", 1)[-1] for l in lg.splitlines() if "handing its survivors over" in l), None),
           # the wall outside the log's phases: process start -> first log line, last log line -> exit (runtime teardown)
           "wall_before_first_log_line_s": round(stamps[0] - t0, 3) if stamps else None,
           "wall_after_last_log_line_s": round(t0 + wall - stamps[-1], 3) if stamps else None}
    return res


def timed_cli(tag, twk, flags, threads, log, runs=2, warm=True):
    """`runs` timed runs (after an untimed one if `warm`) -> the faster run's dict + both walls."""
    out = os.path.join(tempfile.gettempdir(), f"twk_bench_{tag}_{os.getpid()}.two")
    got = []
    for attempt in (["warm-up"] if warm else []) + [f"timed {i + 1}" for i in range(runs)]:
        res = run_cli(twk, flags, threads, out)
        if res is None or "error" in res:
            log(f"{tag}: tomahawk calc failed: {res}")
            return None
        log(f"{tag} {attempt}: wall {res['wall_s']:.2f}s compute+write {res['compute_write_s']} count {res['count_kernel_ms']} ms "
            f"in {res['count_launches']} launches, records {res['records']}")
        if attempt.startswith("timed"):
            got.append(res)
    try:
        os.remove(out)
    except OSError:
        pass
    best = min(got, key=lambda x: x["wall_s"])
    best["wall_s_of_timed_runs"] = [x["wall_s"] for x in got]
    best["command"] = "tomahawk calc " + " ".join(flags) + f" -t {threads}"
    return best


def cohort_twk(n_samples, n_variants, log, **kw):
    """The cohort-shaped input of the from-disk runs, written once per box under /tmp."""
    from tomahawk_amd import hostlib
    threads = min(os.cpu_count() or 8, 64)
    tag = "_".join(f"{k}{v}" for k, v in sorted(kw.items()))
    twk = os.path.join(tempfile.gettempdir(), f"twk_bench_cohort_{n_samples}_{n_variants}{'_' + tag if tag else ''}.twk")
    if not os.path.exists(twk):
        t0 = time.time()
        args = dict(seed=11, n_threads=threads, block_size=128)
        args.update(kw)
        hostlib.write_cohort_twk(twk + ".tmp", n_samples, n_variants, **args)
        os.replace(twk + ".tmp", twk)
        os.sync()                  # the timed runs should read the file, not compete with its write-back
        log(f"wrote {twk} ({os.path.getsize(twk) / 1e6:.0f} MB) in {time.time() - t0:.1f}s")
    return twk, threads


def e2e_from_disk(n_samples, n_variants, log, flags=(), tag="e2e"):
    """`tomahawk calc` (default mode: r2 screen on; flags: e.g. -u) from a cohort-shaped .twk on disk -> dict for the JSON line."""
    twk, threads = cohort_twk(n_samples, n_variants, log)
    res = timed_cli(tag, twk, list(flags), threads, log)
    if res:
        res["screen"] = "on"
        res["input"] = (f"{n_samples} samples x {n_variants} cohort-shaped variants (founder mosaics, 70 % rare), "
                        f"{os.path.getsize(twk) / 1e6:.0f} MB .twk")
        res["wall_s_of_both_timed_runs"] = res["wall_s_of_timed_runs"]
    return res


# The reference's only published workloads (docs/tutorial.md:177-199, 252-253): 1000 Genomes chr6, 2,504 samples,
# 531,500 variants, `calc -p` r2 >= 0.1 over all 141,245,859,250 pairs (49.9 M records; 26 m 13 s on 8 CPU threads = 89.8 M
# pairs/s), and `calc -p -w 4000000`.  Here on a cohort-shaped synthetic .twk of the same shape: positions 322 bp apart
# (171 Mb / 531,500), founder mosaics that switch between blocks of 500 variants.
# p_switch 0.12 (a haplotype's founder changes at 12 % of the block boundaries): 141.2 G pairs leave about as many records
# as the reference's chr6 run did (tests/sweeps/kg_shape.py: 242 M records at 0.02, 36 M at 0.2, 22 M at 1.0).
KG = dict(n_samples=2504, n_variants=531_500, spacing=322, block_size=500, seed=6, p_switch=0.12)


def extra_kg(log):
    kw = {k: v for k, v in KG.items() if k not in ("n_samples", "n_variants")}
    twk, threads = cohort_twk(KG["n_samples"], KG["n_variants"], log, **kw)
    out = {"input": f"{KG['n_samples']} samples x {KG['n_variants']} cohort-shaped variants, {os.path.getsize(twk) / 1e6:.0f} MB .twk "
                    f"(the shape of the reference's tutorial run, docs/tutorial.md:177-199)",
           "reference_published": {"all_pairs": "89.8 M pairs/s, 26 m 12.8 s, 49,870,388 records (8 CPU threads, SSE4, 1000 Genomes chr6)",
                                   "window_4mb": "72.1 M pairs/s (4,784,608 variants)"}}
    out["all_pairs"] = timed_cli("kg_all", twk, ["-p"], threads, log, runs=1, warm=True)
    out["window_4mb"] = timed_cli("kg_w4m", twk, ["-p", "-w", "4000000"], threads, log, runs=1, warm=False)
    # (the default since round 6: at -k 1 the output blocks' zstd frames come from the records' own encoder, csrc/host/twk_repcodec.h - a file
    # a few per cent larger, written without libzstd's level-1 match search on the critical path)  The same run through libzstd, as in round 5:
    out["window_4mb_libzstd"] = timed_cli("kg_w4m_libzstd", twk, ["-p", "-w", "4000000", "--engine-option", "record_codec=0"], threads, log, runs=1, warm=False)
    return out


def extra_in_process(config, log, steps, warmup, seed=42, emulate_shard=None, options=()):
    """Another BASELINE config on this GPU through the same calls as the timed region (synthetic input in HBM) -> dict."""
    import tomahawk_amd as T
    from tomahawk_amd.dist import window_slab
    n_samples, n_variants, mode = CONFIGS[config]
    hip_mode = T.MODE_UNPHASED if mode == "unphased" else T.MODE_PHASED
    filters = T.Filters(minP=MIN_P.get(config, 1.0))
    window_bp = WINDOW_BP.get(config, 0)
    eng = T.HipLd(0)
    for k_, v_ in options:
        eng.set_option(k_, int(v_))
    try:
        t0 = time.time()
        if window_bp:
            k, n = emulate_shard
            r0, r1, col_end, expected = window_slab(n_variants, window_bp // 100, k, n)
            eng.set_problem(n_samples, col_end - r0)
            eng.generate_synthetic(seed, first_variant=r0)
            call = lambda: eng.ld_region(hip_mode, filters, 0, r1 - r0, 0, col_end - r0, True, window=1, l_window=window_bp)
        else:
            eng.set_problem(n_samples, n_variants)
            eng.generate_synthetic(seed)
            call = lambda: eng.ld_all(hip_mode, filters)
        setup = time.time() - t0
        cold_ms = None
        for i in range(warmup):
            t_w = time.perf_counter()
            call()
            if i == 0:
                cold_ms = (time.perf_counter() - t_w) * 1e3      # the first pass: plane sets built, buffers allocated, pages first touched
        eng.timing_reset()
        t0 = time.perf_counter()
        pairs = recs = 0
        for _ in range(steps):
            _, p, r = call()
            pairs += p; recs += r
        el = time.perf_counter() - t0
        tm = eng.timing()
    finally:
        eng.close()
    lane_ops = 2 * ((2 * n_samples + 31) // 32) if mode == "phased" else 8 * ((n_samples + 31) // 32)
    k_s = tm["count_ms"] * 1e-3
    products, ors, kernel, form = executed_work(tm)
    res = {"workload": f"BASELINE {NAMES[config]}: {n_samples} x {n_variants} {mode}" + (f", +-{window_bp} bp, P<={filters.minP:g}, EMULATED shard {emulate_shard[0]}/{emulate_shard[1]}" if window_bp else ""),
           "steps": steps, "warmup": warmup, "pairs_per_step": pairs // max(steps, 1), "value": pairs / el, "unit": "variant-pairs/s",
           "ms_per_step": el / steps * 1e3, "cold_first_step_ms": cold_ms, "survivors_per_step": recs / steps, "setup_s": setup,
           "dominant_kernel": kernel, "form": form, "count_launches_per_step": tm["count_launches"] / steps,
           "avg_launch_ms": tm["count_ms"] / max(tm["count_launches"], 1), "count_kernel_ms_per_step": tm["count_ms"] / steps,
           "math_kernels_ms_per_step": tm["stats_ms"] / steps,
           # frac: executed lane-ops (v_and or v_bitop3 + v_bcnt per product; `ors` is 0 since round 6) against the lane peak;
           # algorithmic_frac: SURVEY 8(d)'s four products per unphased pair, whatever was executed
           "frac": (2 * products + ors) / k_s / VALU_LANE_PEAK if k_s > 0 else None,
           "algorithmic_frac": pairs * lane_ops / k_s / VALU_LANE_PEAK if k_s > 0 else None,
           "and_bcnt_ceiling_frac": pairs * lane_ops / 2 / k_s / VALU_PAIR_PEAK if k_s > 0 else None,
           "executed_frac_of_and_bcnt_ceiling": products / k_s / VALU_PAIR_PEAK if k_s > 0 else None,
           "executed_frac_of_issue_ceiling": (products + ors / 3.0) / k_s / VALU_PAIR_PEAK if k_s > 0 else None,
           "shader_mhz": (tm["count_shader_cycles"] / tm["count_wall_ticks"] * 100.0) if tm["count_wall_ticks"] else None}
    log(f"extra {config}: {res['value'] / 1e6:.1f} M pairs/s, {res['ms_per_step']:.1f} ms/step, count {res['count_kernel_ms_per_step']:.1f} ms, "
        f"and+bcnt {res['and_bcnt_ceiling_frac']}")
    return res


def record_hashes(recs):
    """One 64-bit hash per 104-byte record (numpy, wrap-around arithmetic): every 8-byte word multiplied by its own odd
    constant, folded, summed over the record's 13 words and mixed again.  A set of records is summarised by the wrap-around SUM
    of these - independent of the order the records arrive in, additive over shards."""
    import numpy as np
    w = np.ascontiguousarray(recs).view(np.uint64).reshape(len(recs), -1)
    k = (np.arange(1, w.shape[1] + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)
    with np.errstate(over="ignore"):
        x = w * k
        x ^= x >> np.uint64(29)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        h = x.sum(axis=1, dtype=np.uint64)
        h ^= h >> np.uint64(32)
        h *= np.uint64(0x94D049BB133111EB)
    return h


def gather_check(eng, rank, world, xdev, gather_group, collective, seed, log):
    """N > 1, after the timed region: does the gather move records, and the right ones?  The headline workload has no survivors
    (iid genotypes, r2 >= 0.1), so its gather carries one count per rank and nothing else.  Here one step of a small, survivor-rich
    workload (2,000 samples x 4,096 variants, r2 >= 0.0005) runs through the same calls as a timed step - row band of this rank,
    survivors kept in HBM, gathered to rank 0 over the group that carried the timed gathers (RCCL when it came up) - and the result is
    checked against what every rank knows by itself: each rank copies its own survivors to its own host memory (plain D2H, no
    collective), hashes them, and the per-rank counts and hash sums travel over the gloo control group; rank 0 hashes every
    rank's slice of the gathered buffer and compares.  -> dict for extra.gather_check (rank 0), None elsewhere."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import tomahawk_amd as T
    from tomahawk_amd.dist import LAST_GATHER, gather_records
    n_samples, n_variants = 2000, 4096
    t0 = time.perf_counter()
    eng.set_problem(n_samples, n_variants)
    eng.generate_synthetic(seed)
    eng.set_device_sink(True)
    filters = T.Filters(minR2=0.0005)
    _, npairs, nrec = eng.ld_all(T.MODE_UNPHASED, filters, part=rank, n_parts=world)
    dev_recs = eng.device_records_tensor()
    assert dev_recs.numel() == nrec * T.RECORD_DTYPE.itemsize
    torch.cuda.synchronize()
    own = dev_recs.cpu().numpy().view(T.RECORD_DTYPE).copy()            # this rank's survivors, by its own D2H copy
    with np.errstate(over="ignore"):
        own_sum = int(record_hashes(own).sum(dtype=np.uint64)) if len(own) else 0
    dist.barrier()
    t1 = time.perf_counter()
    got = gather_records(dev_recs, dst=0, device=xdev, group=gather_group)
    t2 = time.perf_counter()
    xfer_s, xfer_bytes = LAST_GATHER["seconds"], LAST_GATHER["bytes"]
    # what every rank knows of its own shard, over the control group (int64 carries the 64 bits)
    mine = torch.tensor([len(own), own_sum - (1 << 64) if own_sum >= (1 << 63) else own_sum, npairs], dtype=torch.int64)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    me = torch.tensor([rank], dtype=torch.int64, device=xdev)
    seen = [torch.zeros_like(me) for _ in range(world)]
    dist.all_gather(seen, me, group=gather_group)
    eng.set_device_sink(False)
    if rank != 0:
        return None
    counts = [int(e[0].item()) for e in every]
    sums = [int(e[1].item()) % (1 << 64) for e in every]
    pairs = sum(int(e[2].item()) for e in every)
    equal_per_rank, off = [], 0
    ok_len = got is not None and len(got) == sum(counts)
    for r in range(world):
        if not ok_len:
            equal_per_rank.append(False)
            continue
        sl = got[off:off + counts[r]]
        off += counts[r]
        with np.errstate(over="ignore"):
            h = int(record_hashes(sl).sum(dtype=np.uint64)) if len(sl) else 0
        equal_per_rank.append(h == sums[r])
    res = {"workload": f"{n_samples} samples x {n_variants} variants, calc -u, r2>={filters.minR2:g}: one step after the timed region",
           "backend": collective, "ranks_seen": sorted(int(x.item()) for x in seen), "pairs": pairs, "records": sum(counts),
           "records_per_rank": counts, "bytes": xfer_bytes, "seconds": xfer_s, "GBps": (xfer_bytes / xfer_s / 1e9) if xfer_s > 0 and xfer_bytes else None,
           "equal": bool(ok_len and all(equal_per_rank)), "equal_per_rank": equal_per_rank,
           "hash": "per rank: wrap-around sum of a 64-bit hash per 104-byte record, own D2H copy vs that rank's slice of the gathered buffer",
           "wall_s": time.perf_counter() - t0, "gather_wall_s": t2 - t1}
    log(f"gather_check: {res['records']} records, {xfer_bytes / 1e6:.1f} MB over {collective.split(' ')[0]} in {xfer_s * 1e3:.1f} ms, equal={res['equal']}")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--variants", type=int, default=0, help="override the number of variants (debug)")
    ap.add_argument("--samples", type=int, default=0, help="override the number of samples (debug)")
    ap.add_argument("--tile", type=int, default=0, help="super-tile edge in variants (0 = engine default)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (debug: gloo lets "
                    "several ranks share one GPU, to exercise the N > 1 orchestration on a single-GPU box)")
    ap.add_argument("--min-r2", type=float, default=None, help="override the r2 cut-off (debug: produce survivors)")
    ap.add_argument("--min-p", type=float, default=None, help="override the Fisher P cut-off (debug)")
    ap.add_argument("--keep-two", default="", help="rank 0 keeps the .two file of the last timed step at this path")
    ap.add_argument("--no-e2e", action="store_true", help="skip the from-disk `tomahawk calc` measurement (N=1, cfg3)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 counter passes behind roofline.traffic (N=1)")
    ap.add_argument("--no-extra", action="store_true", help="skip the other regimes reported under \"extra\" (N=1, cfg3): configs[1], a "
                    "configs[4] shard, calc -u from disk, the reference's published 2,504 x 531,500 runs")
    ap.add_argument("--no-planted", action="store_true", help="skip the survivor-bearing steps on the same problem with planted LD that follow the timed region (extra.<config>_planted)")
    ap.add_argument("--no-gather-check", action="store_true", help="skip the survivor-rich gather self-check that follows the timed region when N > 1")
    ap.add_argument("--e2e-variants", type=int, default=0, help="variants of the e2e input (0: the config's own count)")
    ap.add_argument("--engine-option", action="append", default=[], metavar="KEY=INT", help="a switch of the engine "
                    "(twk_hip_set_option, include/twk_hip.h; measurement runs of profiles/collect.sh)")
    ap.add_argument("--emulate-shard", default="", help="K/N: run shard K of N on this one GPU (validation of the "
                    "sharded configs on a single-GPU box; the value then covers that shard only)")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Libraries underneath do not know that - RCCL prints a banner of its version, the host name
    # and its own path to stdout the first time a communicator comes up (in every rank of an N > 1 run, and in the native-gather leg at N = 1),
    # through C stdio, flushed when the process ends: behind the JSON line.  So file descriptor 1 is pointed at stderr for the life of the
    # process and the line goes out through a duplicate of the real stdout made before that.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    os.environ.setdefault("NCCL_DEBUG", "NONE")       # (... and RCCL is told not to print it at all, unless the caller wants its log - this pool's
                                                      # image exports NCCL_DEBUG=VERSION, so here the redirection is what keeps the line alone)

    import torch
    import torch.distributed as dist
    import numpy as np
    import tomahawk_amd as T
    from tomahawk_amd.dist import LAST_GATHER, gather_records, init_groups, window_slab, window_total_pairs

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    if not torch.cuda.is_available() or T.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    # one GPU per rank: LOCAL_RANK indexes the visible devices, unless the launcher already narrowed
    # visibility to one device per process
    n_vis = torch.cuda.device_count()
    dev_index = local_rank if local_rank < n_vis else local_rank % max(n_vis, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # Control traffic (barriers, the statistics all-reduce) runs over a gloo group, the gather of the survivors over
    # RCCL - after every rank has agreed that RCCL came up (tomahawk_amd/dist.py init_groups).
    xdev = torch.device("cpu")          # where the gather's transfer buffers live
    gather_group = None
    collective = "none (one rank)"
    if world > 1:
        gather_group, xdev, collective = init_groups(args.backend, dev, force_rccl_failure=bool(os.environ.get("TWK_BENCH_FORCE_RCCL_FAILURE")))

    n_samples, n_variants, mode = CONFIGS[args.config]
    if args.variants:
        n_variants = args.variants
    if args.samples:
        n_samples = args.samples
    hip_mode = T.MODE_UNPHASED if mode == "unphased" else T.MODE_PHASED
    filters = T.Filters(minP=MIN_P.get(args.config, 1.0))   # reference defaults: r2 >= 0.1, P <= 1
    if args.min_r2 is not None:
        filters.minR2 = args.min_r2
    if args.min_p is not None:
        filters.minP = args.min_p
    window_bp = WINDOW_BP.get(args.config, 0)
    shard_rank, shard_world = rank, world
    if args.emulate_shard:
        assert world == 1, "--emulate-shard is a single-process facility"
        shard_rank, shard_world = (int(x) for x in args.emulate_shard.split("/"))

    eng = T.HipLd(dev_index)
    for kv in args.engine_option:
        k, v = kv.split("=", 1)
        eng.set_option(k, int(v))
    t0 = time.time()
    slab = None
    if window_bp:
        # Band of rows with 1/world of the in-window pairs + the halo its last row reaches (positions are
        # 1000 + 100 v: SURVEY 8(d)); every rank derives the same partition from the positions alone.
        wv = window_bp // 100
        r0, r1, col_end, my_expected = window_slab(n_variants, wv, shard_rank, shard_world)
        slab = (r0, r1, col_end)
        eng.set_problem(n_samples, col_end - r0)
        eng.generate_synthetic(args.seed, first_variant=r0)
        total_pairs = window_total_pairs(n_variants, wv)
    else:
        eng.set_problem(n_samples, n_variants)
        eng.generate_synthetic(args.seed)          # every rank generates the same bits in its own HBM
        total_pairs = n_variants * (n_variants - 1) // 2
    torch.cuda.synchronize()
    log(f"{args.config}: N={n_samples} M={n_variants} {mode}{' slab ' + str(slab) if slab else ''}; "
        f"input resident in HBM after {time.time() - t0:.1f}s")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Rank 0 is the writer rank: it packs the survivors of all ranks into a real .two file (forward + reverse
    # blocks, the reference's flush rule, index) inside the timed region.  The files are created - header
    # only - before the clock starts, one per step.
    from tomahawk_amd import hostlib
    out_dir = tempfile.mkdtemp(prefix="twk_bench_") if rank == 0 else None
    all_pos = 1000 + 100 * np.arange(n_variants, dtype=np.uint32)
    all_rid = np.zeros(n_variants, dtype=np.uint32)

    def open_stream(i):
        if rank != 0:
            return None
        return hostlib.TwoStream(os.path.join(out_dir, f"step{i}.two"), n_samples, all_rid, all_pos,
                                 n_threads=min(os.cpu_count() or 1, 32))

    written = {"records": 0}
    last = {"recs": None}
    phase = {"compute": 0.0, "gather": 0.0, "write": 0.0,       # seconds, this rank, summed over the timed steps
             "xfer": 0.0, "xfer_bytes": 0}                      # rank 0: the transfers alone (after every rank has arrived) and their bytes

    def step(stream):
        """One pass of the hot path over this rank's shard + the gather of survivors to rank 0 + the .two blocks."""
        t_a = time.perf_counter()
        if world > 1:
            eng.set_device_sink(True)       # survivors stay in HBM: they leave this GPU over the gather, not over PCIe
        if slab:
            r0, r1, col_end = slab
            recs, npairs, nrec = eng.ld_region(hip_mode, filters, 0, r1 - r0, 0, col_end - r0, True,
                                               tile_variants=args.tile, window=1, l_window=window_bp)
            assert npairs == my_expected, (npairs, my_expected)
        else:
            recs, npairs, nrec = eng.ld_all(hip_mode, filters, part=shard_rank, n_parts=shard_world, tile_variants=args.tile)
        if world > 1:
            recs = eng.device_records_tensor()
            assert recs.numel() == nrec * T.RECORD_DTYPE.itemsize, (recs.numel(), nrec)
            if slab and nrec and slab[0]:        # slab-local variant indices -> global (idxA, idxB: the first two u32 of a record)
                idx = recs.view(torch.int32).view(-1, T.RECORD_DTYPE.itemsize // 4)
                idx[:, 0:2] += slab[0]
            torch.cuda.synchronize()
        elif slab and len(recs) and slab[0]:
            recs["idxA"] += slab[0]; recs["idxB"] += slab[0]
        t_b = time.perf_counter()
        if world > 1:
            # RCCL: all_gather(counts) + grouped send/recv of exact sizes, HBM to HBM; rank 0 copies to host once
            recs = gather_records(recs, dst=0, device=xdev, group=gather_group)
            phase["xfer"] += LAST_GATHER["seconds"]; phase["xfer_bytes"] += LAST_GATHER["bytes"]
        t_c = time.perf_counter()
        if rank == 0:
            stream.append(recs)
            written["records"] += stream.close()
            last["recs"] = recs            # (rank 0: every rank's survivors of this step, as the writer saw them)
        t_d = time.perf_counter()
        phase["compute"] += t_b - t_a; phase["gather"] += t_c - t_b; phase["write"] += t_d - t_c
        return npairs, nrec

    for i in range(args.warmup):
        step(open_stream(f"w{i}"))
    streams = [open_stream(i) for i in range(args.steps)]
    written["records"] = 0
    for k in phase:
        phase[k] = 0
    barrier()
    eng.timing_reset()
    t0 = time.perf_counter()
    my_pairs = my_recs = 0
    for i in range(args.steps):
        p, r = step(streams[i])
        my_pairs += p
        my_recs += r
    barrier()
    elapsed = time.perf_counter() - t0
    if rank == 0:
        import shutil
        if args.keep_two and args.steps:
            shutil.copyfile(os.path.join(out_dir, f"step{args.steps - 1}.two"), args.keep_two)
    tm = eng.timing()
    spread = launch_spread(eng)
    written_timed, phase_timed = dict(written), dict(phase)          # the timed region's own figures (the legs below go through the same step())

    stats = torch.tensor([elapsed, tm["count_ms"], tm["stats_ms"]], dtype=torch.float64)
    sums = torch.tensor([my_pairs, my_recs, tm["count_launches"], tm["row_pairs"]], dtype=torch.float64)
    per_rank_ms, ranks_seen = [phase_timed["compute"] / max(args.steps, 1) * 1e3], [rank]
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        mine = torch.tensor([phase_timed["compute"] / max(args.steps, 1) * 1e3], dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [float(x.item()) for x in every]
        # which ranks the group that carried the gather really connects: an all-gather of the rank ids over it
        me = torch.tensor([rank], dtype=torch.int64, device=xdev)
        seen = [torch.zeros_like(me) for _ in range(world)]
        dist.all_gather(seen, me, group=gather_group)
        ranks_seen = sorted(int(x.item()) for x in seen)

    def planted_leg():
        """A survivor-bearing step of the SAME problem under the same clock (since round 1 the timed region has had 0 survivors: iid
        genotypes hold no pair near r2 = 0.1): the input regenerated in place with LD planted in it (twk_hip_plant: every odd variant a
        noisy copy of an even one, flip probability 0 .. 0.4 - r2 from 1 down to below the cut-off; n_variants / 2 pairs spread over
        the triangle, or 7 variants apart in window runs), one warm-up + two steps through the same step() - count, screen, recount of
        the candidates' four products, UnphasedMath, Fisher, sort, copy-back (N > 1: gather), .two writer - and one more step in the
        four-product form, which screens nothing on (HH, S): every pair's full table goes through the math.  Self-check: the survivors of
        the two forms are the same bytes, every survivor is a planted pair, and planted_found == planted_expected (the planted pairs the
        four-product step kept).  The ORACLE's verdict on this very data set (same generator, seed and plant) is
        tests/test_gpu_full_size.py::test_three_product_form_at_the_headline_size_against_the_oracle."""
        plant = T.Plant.near(n_variants, 7, max_eps=0.4) if window_bp else T.Plant.spread(n_variants, max_eps=0.4)
        t_g = time.perf_counter()
        eng.generate_synthetic(args.seed, first_variant=slab[0] if slab else 0, plant=plant)
        gen_s = time.perf_counter() - t_g

        def run(n_steps, tag):
            for k in phase:
                phase[k] = 0
            written["records"] = 0
            streams = [open_stream(f"{tag}{i}") for i in range(n_steps)]
            barrier()
            eng.timing_reset()
            t_s = time.perf_counter()
            pr = rc = 0
            for i in range(n_steps):
                p_, r_ = step(streams[i])
                pr += p_; rc += r_
            barrier()
            el = time.perf_counter() - t_s
            st = torch.tensor([el, float(pr), float(rc)], dtype=torch.float64)
            if world > 1:
                mx = st.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
                dist.all_reduce(st, op=dist.ReduceOp.SUM)
                st[0] = mx[0]
            return float(st[0]), int(round(float(st[1]))), int(round(float(st[2]))), eng.timing(), dict(phase), last["recs"]

        step(open_stream("pw"))                                   # warm-up: plane set rebuilt from the new rows
        el3, pairs3, recs3, tm3, ph3, r3 = run(2, "p")
        eng.set_option("three", 0)
        el4, pairs4, recs4, tm4, ph4, r4 = run(1, "q")
        eng.unset_option("three")
        native = None
        if world == 1:
            # One GPU: what can be shown of the C++ product's own gather (twk_hip_gather_records: dlopen'd librccl, ncclCommInitAll, one group of
            # exact-size ncclSend / ncclRecv, device sink to device sink) - one more step with the survivors kept in HBM, then round RCCL's loop
            # from the sink to itself, then to the host once (twk_hip_drain_device_sink), and compared with the step above.
            try:
                eng.set_device_sink(True)
                t_n = time.perf_counter()
                if slab:
                    _, _, n_rec = eng.ld_region(hip_mode, filters, 0, slab[1] - slab[0], 0, slab[2] - slab[0], True, tile_variants=args.tile, window=1, l_window=window_bp)
                else:
                    _, _, n_rec = eng.ld_all(hip_mode, filters, part=shard_rank, n_parts=shard_world, tile_variants=args.tile)
                t_c = time.perf_counter()
                n_g, xfer_ms = T.gather_records([eng], self_loop=True)
                back = eng.drain_device_sink()
                if slab and len(back) and slab[0]:
                    back["idxA"] += slab[0]; back["idxB"] += slab[0]
                order_n = ["idxA", "idxB"]
                native = {"backend": T.gather_backend(), "records": int(n_g), "bytes": int(n_g) * T.RECORD_DTYPE.itemsize, "transfer_ms": xfer_ms,
                          "GBps": (n_g * T.RECORD_DTYPE.itemsize / (xfer_ms * 1e-3) / 1e9) if xfer_ms > 0 and n_g else None,
                          "compute_ms": (t_c - t_n) * 1e3, "gather_and_drain_ms": (time.perf_counter() - t_c) * 1e3,
                          "equal_to_streamed_step": bool(r3 is not None and n_g == n_rec == len(r3) and np.sort(back, order=order_n).tobytes() == np.sort(np.asarray(r3), order=order_n).tobytes()),
                          "note": "one GPU: the one context's records go from its device sink to itself through the same group of ncclSend / ncclRecv a multi-GPU run uses"}
            except Exception as e:           # never take the leg down with it
                native = {"error": repr(e)[:300]}
            finally:
                eng.set_device_sink(False)
        if rank != 0:
            return None
        key = lambda r: set(zip(r["idxA"].tolist(), r["idxB"].tolist())) if r is not None and len(r) else set()
        k3, k4 = key(r3), key(r4)
        planted = {}
        for v in range(1, 2 * plant.n_planted, 2):
            src, eps = T.plant_source(args.seed, plant, v)
            if max(src, v) < n_variants:
                planted[(min(src, v), max(src, v))] = eps
        found, expected = k3 & set(planted), k4 & set(planted)
        order = ["idxA", "idxB"]
        same = (r3 is not None and r4 is not None and len(r3) == len(r4)
                and np.sort(np.asarray(r3), order=order).tobytes() == np.sort(np.asarray(r4), order=order).tobytes())
        missed = [e for kk, e in planted.items() if kk not in found]
        products, ors, kernel, form = executed_work(tm3)
        k_s = tm3["count_ms"] * 1e-3
        res = {"workload": f"the headline's problem with LD planted in it: {plant.n_planted} noisy copies (flip probability 0 .. {plant.max_eps:g}, "
                           + ("7 variants from their sources" if window_bp else "sources spread over the whole data set") + f"), same filters, 1 warm-up + 2 steps through the timed region's step()",
               "steps": 2, "warmup": 1, "generate_s": gen_s, "pairs_per_step": pairs3 // 2, "value": pairs3 / el3, "unit": "variant-pairs/s",
               "ms_per_step": el3 / 2 * 1e3, "survivors_per_step": recs3 / 2,
               "dominant_kernel": kernel, "form": form,
               "count_kernel_ms_per_step": tm3["count_ms"] / 2, "count_launches_per_step": tm3["count_launches"] / 2,
               "three_product_launches": int(tm3["three_launches"]), "recounted_candidates_per_step": tm3["recount_candidates"] / 2,
               # rank 0's engine, per step: recount + UnphasedMath + Fisher (HIP events: from the count kernel's end to the end of Fisher's walks),
               # and the calling thread's wall from a launch's last kernel to its records' hand-over (device sort, copy-back, the sink), summed over
               # the launches - most of it waiting for the sort's kernels to get registers beside the NEXT launch's persistent count kernel (the wide
               # lane tile fills the register file), and overlapped by it: the pipeline keeps three launches in flight, compute_ms_per_step says what is left
               "math_kernels_ms_per_step": tm3["stats_ms"] / 2, "sort_copyback_sink_wall_ms_per_step_overlapped": tm3.get("finish_ms", 0.0) / 2,
               "compute_ms_per_step": ph3["compute"] / 2 * 1e3, "gather_ms_per_step": ph3["gather"] / 2 * 1e3, "write_ms_per_step": ph3["write"] / 2 * 1e3,
               "gather_bytes_per_step": ph3["xfer_bytes"] / 2, "gather_GBps": (ph3["xfer_bytes"] / ph3["xfer"] / 1e9) if ph3["xfer"] > 0 and ph3["xfer_bytes"] else None,
               "executed_frac_of_issue_ceiling": (products + ors / 3.0) / k_s / VALU_PAIR_PEAK if k_s > 0 else None,
               "four_product_step": {"ms_per_step": el4 * 1e3, "value": pairs4 / el4, "survivors": recs4, "count_kernel_ms": tm4["count_ms"],
                                     "math_kernels_ms": tm4["stats_ms"], "three_product_launches": int(tm4["three_launches"]),
                                     "executed_frac_of_and_bcnt_ceiling": (executed_work(tm4)[0] / (tm4["count_ms"] * 1e-3) / VALU_PAIR_PEAK) if tm4["count_ms"] > 0 else None},
               "native_gather_self_loop": native,
               "planted_pairs": len(planted), "planted_found": len(found), "planted_expected": len(expected),
               "survivors_not_planted": len(k3 - set(planted)), "records_equal_four_product": bool(same),
               "largest_flip_probability_found": max((planted[kk] for kk in found), default=None),
               "smallest_flip_probability_missed": min(missed, default=None),
               "self_check": bool(same and found == expected and len(found) > 0 and (filters.minR2 < 0.01 or not (k3 - set(planted)))),
               "self_check_note": "planted_expected = planted pairs among the survivors of the four-product step (no (HH, S) screen); records compared byte for byte; "
                                  "the oracle's check of this data set: tests/test_gpu_full_size.py"}
        log(f"planted: {res['ms_per_step']:.1f} ms/step, {res['survivors_per_step']:.0f} survivors, found {len(found)} of {len(expected)} expected, "
            f"equal to four products: {same}; four-product step {el4 * 1e3:.1f} ms")
        return res

    planted = None
    if not args.no_planted and args.steps > 0:
        planted = planted_leg()
    if rank == 0:
        shutil.rmtree(out_dir, ignore_errors=True)
    gcheck = None
    if world > 1 and not args.no_gather_check:
        try:
            gcheck = gather_check(eng, rank, world, xdev, gather_group, collective, args.seed, log)
        except Exception as e:           # a failed check is reported, on every rank alike, not hidden
            gcheck = {"equal": False, "error": repr(e)[:300]} if rank == 0 else None
            log(f"gather_check failed: {e!r}")
    elapsed_max, count_ms_max, stats_ms_max = (float(x) for x in stats.tolist())
    pairs_all, recs_all, launches_all, row_pairs_all = (float(x) for x in sums.tolist())
    if not args.emulate_shard:
        assert int(round(pairs_all)) == total_pairs * args.steps, (pairs_all, total_pairs * args.steps)

    if rank == 0:
        value = pairs_all / elapsed_max
        bytes_per_pair = n_samples / 4.0                    # one partner bitvector: 8*ceil(2N/64) = N/4 bytes
        # algorithmic integer work per variant pair: one AND + one popcount per 32-bit word
        lane_ops_per_pair = 2 * ((2 * n_samples + 31) // 32) if mode == "phased" else 8 * ((n_samples + 31) // 32)
        # dominant kernel on rank 0: its own launches, its own HIP-event time
        k_pairs = my_pairs
        k_ms = tm["count_ms"]
        k_s = k_ms * 1e-3
        lane_ops_per_s = k_pairs * lane_ops_per_pair / k_s if k_ms > 0 else 0.0
        hbm_alg = k_pairs * bytes_per_pair / k_s / 1e9 if k_ms > 0 else 0.0
        words = tm["words_per_row"]
        # what the kernel actually executed (whole 128 x 128 tiles: includes the lower half of diagonal
        # tiles, row padding and, in window mode, the tile corners outside the window): AND+popcount products (and, until
        # round 6, the v_or that formed the three-product form's carrier words: `ors`, now 0)
        products, ors, kernel_name, form = executed_work(tm)
        word_pairs_per_s = products / k_s if k_ms > 0 else 0.0
        executed_lane_ops_per_s = (2 * products + ors) / k_s if k_ms > 0 else 0.0
        out = {
            "metric": "variant-pairs/sec all-vs-all LD, 1M samples; achieved HBM GB/s vs roofline",
            "value": value, "unit": "variant-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": (f"BASELINE {NAMES[args.config]}: {n_samples} samples x {n_variants} variants, "
                                    + (f"windowed +-{window_bp} bp" if window_bp else "all-vs-all")
                                    + f" {mode} genotype LD (calc {'-u' if mode == 'unphased' else '-p'}, r2>={filters.minR2:g}"
                                    + (f", P<={filters.minP:g}" if filters.minP < 1 else "") + f"), {total_pairs} pairs/step; contraction: {form}"
                                    + (f"; EMULATED shard {args.emulate_shard} only" if args.emulate_shard else "")),
                       "n_samples": n_samples, "n_variants": n_variants, "mode": mode, "tile_variants": args.tile,
                       "partition": (f"equal-area row bands of the pair triangle over {world} GPU(s), "
                                     + ("no gather (one rank)" if world == 1 else "RCCL gather of survivors (HBM to HBM)" if collective == "nccl"
                                        else f"gather of survivors over {collective.split(' ')[0]} through host memory")
                                     + ", rank 0 writes the .two file"),
                       "collective_backend": collective,
                       "survivors_per_step": recs_all / args.steps,
                       "two_records_written_per_step": written_timed["records"] / args.steps},
            # achieved / frac: lane-ops the kernel EXECUTED (v_and or v_bitop3 + v_bcnt per product; rounds 5 / 6a also counted the three-product form's
            # v_or, which no longer exist - so `frac` fell from 0.621 while pairs/s rose) - it cannot grow by counting work that was not done; algorithmic_*: SURVEY 8(d)'s figure per pair (four products per unphased pair) over the same time
            "roofline": {"bound": "valu", "achieved": executed_lane_ops_per_s / 1e12, "peak": VALU_LANE_PEAK / 1e12,
                         "unit": "Tlane-op/s", "frac": executed_lane_ops_per_s / VALU_LANE_PEAK,
                         "algorithmic_achieved": lane_ops_per_s / 1e12, "algorithmic_frac": lane_ops_per_s / VALU_LANE_PEAK,
                         "and_bcnt_ceiling_frac": (lane_ops_per_s / 2) / VALU_PAIR_PEAK,
                         "contraction_form": form,
                         "traffic": None,
                         "traffic_note": "not measured in this run (rocprofv3 counter passes skipped or unavailable); see profiles/ for the round's figure",
                         # the clock the kernel's blocks really ran at (s_memtime against the constant 100 MHz counter over every block's
                         # life); peak and ceiling above are quoted at the nominal 2.4 GHz
                         "shader_mhz": (tm["count_shader_cycles"] / tm["count_wall_ticks"] * 100.0) if tm["count_wall_ticks"] else None,
                         "kernel": kernel_name, "launches": int(tm["count_launches"]),
                         "three_product_launches": int(tm["three_launches"]), "recounted_candidates": int(tm["recount_candidates"]),
                         "avg_launch_ms": k_ms / max(tm["count_launches"], 1),
                         "launch_spread": spread,
                         "algorithmic_lane_ops_per_pair": lane_ops_per_pair,
                         "executed_word_pairs_per_s": word_pairs_per_s,
                         "executed_frac_of_and_bcnt_ceiling": word_pairs_per_s / VALU_PAIR_PEAK,
                         # ... with the v_or priced at their 2 cycles (a product: 6): the share of the SIMDs' issue cycles the loop's own instructions fill
                         "executed_frac_of_issue_ceiling": (word_pairs_per_s + (ors / k_s if k_ms > 0 else 0.0) / 3.0) / VALU_PAIR_PEAK,
                         "executed_or_ops_per_s": ors / k_s if k_ms > 0 else 0.0,
                         "words_per_row": int(words),
                         "hbm_algorithmic": {"achieved": hbm_alg, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": hbm_alg / HBM_PEAK_GBS, "bytes_per_pair": bytes_per_pair,
                                             "note": "SURVEY 8(d): one partner bitvector per pair; rows are reused 128x "
                                                     "from LDS, so this exceeds 1 and does not bound the kernel"},
                         "note": "integer AND+popcount: bound by VALU issue. peak = 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz; "
                                 "and+bcnt ceiling = 64 lanes / (2 + 4 cycles) per SIMD = 2.62e13 word pairs/s "
                                 "(v_bcnt_u32_b32 is half rate: profiles/*microbench_valu_rate.txt), i.e. frac <= 0.667 for a loop of nothing "
                                 "but products; the three-product form executes 3 products per word pair (HH, Q_A & (H_B | Q_B), Q_B & (H_A | Q_A): "
                                 "the last two one v_bitop3_b32 each, no v_or), algorithmic_* prices SURVEY 8(d)'s four"},
            "kernel_ms": {"count": count_ms_max, "math": stats_ms_max, "wall": elapsed_max * 1e3},
            "per_rank_ms": per_rank_ms,                                   # compute per step, every rank (balance of the bands)
            "gather_ms": phase_timed["gather"] / max(args.steps, 1) * 1e3,      # rank 0, per step: waits for the slowest rank, then the transfers
            "write_ms": phase_timed["write"] / max(args.steps, 1) * 1e3,        # rank 0, per step: survivors -> .two blocks -> file
            "gather_bytes": phase_timed["xfer_bytes"] / max(args.steps, 1),     # rank 0, per step: record bytes received from the other ranks
            "gather_GBps": (phase_timed["xfer_bytes"] / phase_timed["xfer"] / 1e9) if phase_timed["xfer"] > 0 and phase_timed["xfer_bytes"] else None,   # over the transfers alone
            "ranks_seen": ranks_seen,
        }
        if gcheck is not None:
            out["extra"] = {"gather_check": gcheck}
        if planted is not None:
            out.setdefault("extra", {})[f"{args.config}_planted"] = planted
        if world == 1 and not args.no_cpu_baseline:
            try:
                cb = cpu_baseline(n_samples, mode, args.seed, log)
            except Exception as e:  # the baseline must never take the GPU number down with it
                log(f"cpu_baseline failed: {e!r}")
                cb = None
            if cb:
                out["cpu_baseline"] = cb
        if world == 1 and not args.no_traffic and not args.emulate_shard:
            eng.close()                  # the profiled child runs the same workload on the same GPU
            cfg_args = ["--config", args.config, "--seed", str(args.seed)]
            if args.variants:
                cfg_args += ["--variants", str(args.variants)]
            if args.samples:
                cfg_args += ["--samples", str(args.samples)]
            if args.min_r2 is not None:
                cfg_args += ["--min-r2", str(args.min_r2)]
            if args.tile:
                cfg_args += ["--tile", str(args.tile)]
            for kv in args.engine_option:
                cfg_args += ["--engine-option", kv]
            try:
                tr = measure_traffic(cfg_args, log)
            except Exception as e:       # never take the GPU number down with it
                log(f"traffic measurement failed: {e!r}")
                tr = None
            if tr:
                out["roofline"]["traffic"] = tr["bytes_per_launch"]
                out["roofline"]["traffic_detail"] = tr
                out["roofline"]["traffic_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes over one step of this workload, run "
                                                   "as child processes after the timed region), FETCH_SIZE x2 (gfx950), bytes per launch of "
                                                   "the count kernel; FETCH_SIZE counts the L2s' fabric requests (Infinity-Cache hits "
                                                   "included): an upper bound on HBM reads")
        if world == 1 and not args.no_e2e and args.config == "cfg3" and not args.emulate_shard:
            eng.close()                  # the CLI gets the whole GPU
            try:
                e2e = e2e_from_disk(n_samples, args.e2e_variants or n_variants, log)
            except Exception as e:       # never take the GPU number down with it
                log(f"e2e failed: {e!r}")
                e2e = None
            if e2e:
                out["e2e"] = e2e
        if world == 1 and not args.no_extra and args.config == "cfg3" and not args.emulate_shard and not args.variants and not args.samples:
            # Every other regime the repository makes claims about, under the same clock as the headline (after the timed
            # region, like cpu_baseline / e2e): each with its pairs/s, dominant kernel, average launch and roofline fraction.
            eng.close()
            extra = out.get("extra", {})
            for name, fn in (("cfg2", lambda: extra_in_process("cfg2", log, steps=20, warmup=3)),
                             ("cfg5_shard", lambda: extra_in_process("cfg5", log, steps=1, warmup=1, emulate_shard=(3, 8))),
                             ("e2e_u", lambda: e2e_from_disk(n_samples, args.e2e_variants or n_variants, log, flags=("-u",), tag="e2e_u")),
                             ("kg", lambda: extra_kg(log))):
                t_x = time.time()
                try:
                    extra[name] = fn()
                except Exception as e:   # never take the GPU number down with it
                    log(f"extra {name} failed: {e!r}")
                    extra[name] = None
                log(f"extra {name}: {time.time() - t_x:.1f}s")
            out["extra"] = extra
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
