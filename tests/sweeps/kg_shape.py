"""Which cohort generator settings give the reference's published workload (docs/tutorial.md:177-199: 2,504 samples x
531,500 variants, calc -p, r2 >= 0.1: 49.9 M records of 141.2 G pairs) its record density?  Writes the input for several
founder-switch rates, runs `tomahawk calc -p` and `-p -w 4000000`, prints records / times.
  python tests/sweeps/kg_shape.py [n_variants=531500] [p_switch ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from tomahawk_amd import hostlib as H

M = int(sys.argv[1]) if len(sys.argv) > 1 else 531_500
switches = [float(x) for x in sys.argv[2:]] or [0.02, 0.2, 1.0]
threads = min(os.cpu_count() or 8, 64)
log = lambda m: print("[kg_shape] " + m, flush=True)
for ps in switches:
    twk = f"/tmp/kg_shape_{M}_{ps}.twk"
    t = time.time()
    H.write_cohort_twk(twk, 2504, M, seed=6, n_threads=threads, block_size=500, spacing=322, p_switch=ps)
    log(f"p_switch {ps}: wrote {os.path.getsize(twk) / 1e6:.0f} MB in {time.time() - t:.1f} s")
    for flags in (["-p"], ["-p", "-w", "4000000"], ["-p", "--engine-option", "band_launch=0"], ["-p", "-w", "4000000", "--engine-option", "band_launch=0"],
                  ["-u"], ["-u", "-w", "4000000"]):
        r = bench.run_cli(twk, flags, threads, "/tmp/kg_shape.two")
        log(f"p_switch {ps} calc {' '.join(flags)}: " + ", ".join(f"{k}={r.get(k)}" for k in ("wall_s", "load_s", "compute_write_s", "pairs", "records", "two_bytes", "count_kernel_ms", "count_launches", "and_bcnt_ceiling_frac", "math_kernels_ms", "fused_launches", "error") if r.get(k) is not None))
    os.remove(twk)
