import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
threads = 64
log = lambda m: print("[emit] " + m, flush=True)
big, _ = bench.cohort_twk(bench.KG["n_samples"], bench.KG["n_variants"], log, **{k: v for k, v in bench.KG.items() if k not in ("n_samples", "n_variants")})
for w in (16, 32, 48, 64):
    for flags in (["-p"], ["-p", "-w", "4000000"]):
        best = None
        for _ in range(2):
            r = bench.run_cli(big, flags + ["--engine-option", f"emit_workers={w}"], threads, "/tmp/emit_ab.two")
            if best is None or r["compute_write_s"] < best["compute_write_s"]: best = r
        log(f"emit_workers={w} {' '.join(flags)}: wall {best['wall_s']:.2f} compute+write {best['compute_write_s']:.3f} handover {best['producer_handover_s']} records {best['records']}")
