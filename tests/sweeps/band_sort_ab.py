"""The reference's published workload shape (2,504 x 531,500, calc -p, -p -w 4000000, -u) through the CLI: the band launches'
order (band_reverse) and the emitter's backlog (emit_backlog_mb: 64 MB ~ the producer waits for the compression) A/B.
  python tests/sweeps/band_sort_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
threads = 64
log = lambda m: print("[band_sort] " + m, flush=True)
big, _ = bench.cohort_twk(bench.KG["n_samples"], bench.KG["n_variants"], log, **{k: v for k, v in bench.KG.items() if k not in ("n_samples", "n_variants")})
for rev, backlog in ((1, 64), (1, 256), (1, 512), (1, 1024), (1, 2048), (0, 512)):
    for flags in (["-p"], ["-p", "-w", "4000000"], ["-u"]):
        best = None
        for _ in range(2):
            r = bench.run_cli(big, flags + ["--engine-option", f"band_reverse={rev}", "--engine-option", f"emit_backlog_mb={backlog}"], threads, "/tmp/band_sort_ab.two")
            if best is None or r["compute_write_s"] < best["compute_write_s"]: best = r
        log(f"band_reverse={rev} emit_backlog_mb={backlog} {' '.join(flags)}: wall {best['wall_s']:.2f} compute+write {best['compute_write_s']:.3f} count {best.get('count_kernel_ms')} ms in {best.get('count_launches')} "
            f"math {best.get('math_kernels_ms')} handover {best['producer_handover_s']} records {best['records']}")
