"""The writer's direct mode (engine option direct_output: one pwritev a frame, space reserved ahead) against the iostream, on the
survivor-rich window run with the record codec on (where the one stream into the file is the floor) and off.
  python tests/sweeps/direct_output_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
log = lambda m: print("[direct] " + m, flush=True)
big, _ = bench.cohort_twk(bench.KG["n_samples"], bench.KG["n_variants"], log, **{k: v for k, v in bench.KG.items() if k not in ("n_samples", "n_variants")})
for flags in (["-p", "-w", "4000000"], ["-u", "-w", "1000000"]):
    for codec in (1, 0):
        for out in ("/tmp/direct_output_ab.two", "/dev/shm/direct_output_ab.two"):
            for direct in (0, 1, 0, 1):
                best = None
                for _ in range(3):
                    r = bench.run_cli(big, flags + ["--engine-option", f"record_codec={codec}", "--engine-option", f"direct_output={direct}"], 64, out)
                    if "error" in r: log(str(r)); break
                    if best is None or r["compute_write_s"] < best["compute_write_s"]: best = r
                if best: log(f"record_codec={codec} direct_output={direct} {' '.join(flags)} -> {out}: compute + write {best['compute_write_s']:.3f} s | {best['writer_line'].split('workers:')[-1]}")
for f in ("/tmp/direct_output_ab.two", "/dev/shm/direct_output_ab.two"):
    try: os.remove(f)
    except OSError: pass
