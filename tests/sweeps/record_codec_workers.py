import os, sys
sys.path.insert(0, os.getcwd())
import bench
log = lambda m: print("[ew] " + m, flush=True)
big, _ = bench.cohort_twk(bench.KG["n_samples"], bench.KG["n_variants"], log, **{k: v for k, v in bench.KG.items() if k not in ("n_samples", "n_variants")})
for flags in (["-p", "-w", "4000000"],):
    for w in (12, 16, 32):
        for q in (0, 2, 8):
            best = None
            for _ in range(3):
                r = bench.run_cli(big, flags + ["--engine-option", "record_codec=1", "--engine-option", f"emit_workers={w}", "--engine-option", f"emit_queue_pieces={q}"], 64, "/tmp/ew.two")
                if best is None or r["compute_write_s"] < best["compute_write_s"]: best = r
            log(f"emit_workers={w} emit_queue_pieces={q}: wall {best['wall_s']:.2f} compute+write {best['compute_write_s']:.3f} | {best['writer_line']}")
