"""The records' own zstd encoder (engine option record_codec, csrc/host/twk_repcodec.h) against libzstd level 1 on the
survivor-rich runs that the host's compression binds: the reference's published workload shape (2,504 x 531,500, `calc -p`),
all pairs and a 4 Mb window, and the 1 M-sample cohort run.  For each: wall, compute + write, file size; the output of the codec
run is read back by this repo's reader through the sorted-record hash of bench.record_hashes where small enough, and by the
compiled reference's `view` (first and last records) when oracle/_ref is there.
  python tests/sweeps/record_codec_ab.py"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from oracle import oracle as O
threads = 64
log = lambda m: print("[codec] " + m, flush=True)
big, _ = bench.cohort_twk(bench.KG["n_samples"], bench.KG["n_variants"], log, **{k: v for k, v in bench.KG.items() if k not in ("n_samples", "n_variants")})
outs = {}
for flags in ([["-p", "-w", "4000000"]] if "--short" in sys.argv else [["-p", "-w", "4000000"], ["-p"], ["-u", "-w", "1000000"]]):
    for codec in (0, 1):
        best = None
        out = f"/tmp/codec_ab_{codec}.two"
        for _ in range(3):
            r = bench.run_cli(big, flags + ["--engine-option", f"record_codec={codec}"], threads, out)
            if "error" in r: log(str(r)); break
            if best is None or r["compute_write_s"] < best["compute_write_s"]: best = r
        if best:
            if True: log(f"  record_codec={codec} writer: " + best.get("writer_line", ""))
            log(f"record_codec={codec} {' '.join(flags)}: wall {best['wall_s']:.2f} s, compute + write {best['compute_write_s']:.3f} s, count kernel {best['count_kernel_ms']:.1f} ms, "
                f"records {best['records']:,}, file {best['two_bytes'] / 1e9:.3f} GB")
    if O.have_ref() and flags == ["-p", "-w", "4000000"]:
        # the reference reads both files: same text
        a = subprocess.run(f"timeout 600 {O.REF_BIN} view -i /tmp/codec_ab_0.two | grep -v '^#' | md5sum", shell=True, capture_output=True, text=True).stdout.split()[0]
        b = subprocess.run(f"timeout 600 {O.REF_BIN} view -i /tmp/codec_ab_1.two | grep -v '^#' | md5sum", shell=True, capture_output=True, text=True).stdout.split()[0]
        log(f"reference `view` of both files: md5 {a} / {b} -> {'equal' if a == b else 'DIFFERENT'}")
