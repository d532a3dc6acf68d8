"""What is left of a survivor-rich run once the records' own encoder compresses its blocks (record_codec=1): the same command
with the output going to /dev/null (through a symlink: the writer opens it like a file) - the pipeline without the file system -
against a file in /tmp and one in /dev/shm.
  python tests/sweeps/record_codec_floor.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
log = lambda m: print("[floor] " + m, flush=True)
big, _ = bench.cohort_twk(bench.KG["n_samples"], bench.KG["n_variants"], log, **{k: v for k, v in bench.KG.items() if k not in ("n_samples", "n_variants")})
null = "/tmp/record_codec_floor_null.two"
if os.path.lexists(null): os.remove(null)
os.symlink("/dev/null", null)
for flags in (["-p", "-w", "4000000"], ["-p"]):
    for codec in (0, 1):
        for out in ("/tmp/record_codec_floor.two", "/dev/shm/record_codec_floor.two", null):
            best = None
            for _ in range(3):
                r = bench.run_cli(big, flags + ["--engine-option", f"record_codec={codec}"], 64, out, keep_out=(out == null))
                if "error" in r: log(str(r)); break
                if best is None or r["compute_write_s"] < best["compute_write_s"]: best = r
            if best: log(f"record_codec={codec} {' '.join(flags)} -> {out}: compute + write {best['compute_write_s']:.3f} s | {best['writer_line']}")
for f in ("/tmp/record_codec_floor.two", "/dev/shm/record_codec_floor.two", null):
    try: os.remove(f)
    except OSError: pass
