"""Robustness / timing of `tomahawk calc` at large variant counts and a small cohort (1000-Genomes shape):
  2,504 samples x 1,000,000 variants, `-w 100000` (window mode, ~1e9 in-window pairs), and
  2,504 samples x 200,000 variants all-vs-all (2e10 pairs) with and without the r2 screen."""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tomahawk_amd import hostlib as H
threads = min(os.cpu_count() or 8, 64)


def run(tag, twk, args, env=None):
    out = "/tmp/large_m.two"
    t = time.time()
    r = subprocess.run([H.CLI_PATH, "calc", "-i", twk, "-o", out, "-t", str(threads)] + args, capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    wall = time.time() - t
    if r.returncode != 0:
        print(tag, "FAILED", r.stderr[-600:]); return
    fin = re.search(r"Finished in (\S+)\. Variants: ([0-9,]+), genotypes: [0-9,]+, output: ([0-9,]+)", r.stderr)
    load = re.search(r"Unpacked and uploaded .* variants\. (\S+)", r.stderr)
    eng = re.findall(r"count kernel ([0-9.e+]+) ms in (\d+) launches \(([0-9.e+-]+) % [^)]*\), math kernels ([0-9.e+]+) ms", r.stderr)
    eng += re.findall(r"(carrier-list kernel [0-9.e+]+ ms in \d+ launches over [0-9,]+ rare pairs)", r.stderr)
    eng += re.findall(r"(\d+ launches fused count -> r2 screen, [0-9,]+ candidate pairs)", r.stderr)
    print(f"{tag}: wall {wall:.2f} s | load {load.group(1) if load else '?'} | compute+write {fin.group(1)} | pairs {fin.group(2)} | records {fin.group(3)} | engine {eng}", flush=True)
    os.remove(out)


big = "/tmp/kg_2504_1m.twk"
if not os.path.exists(big):
    t = time.time(); H.write_cohort_twk(big, 2504, 1_000_000, seed=21, n_threads=threads, block_size=500, spacing=100, n_contigs=4)
    print(f"wrote {big}: {os.path.getsize(big)/1e6:.0f} MB in {time.time()-t:.1f} s", flush=True)
run("2504 x 1M, calc -w 100000 -r 0.5", big, ["-w", "100000", "-r", "0.5"])
run("2504 x 1M, calc -w 100000 -r 0.5, 2 contexts on one GPU (slabs)", big, ["-w", "100000", "-r", "0.5", "--engine-option", "force_device=0"], {"TWK_HIP_GPUS": "2"})
mid = "/tmp/kg_2504_200k.twk"
if not os.path.exists(mid):
    H.write_cohort_twk(mid, 2504, 200_000, seed=12, n_threads=threads, block_size=500, spacing=100)
run("2504 x 200k all-vs-all, calc -r 0.8", mid, ["-r", "0.8"])
run("2504 x 200k all-vs-all, calc -r 0.8 WITHOUT the r2 screen", mid, ["-r", "0.8"], {"TWK_HIP_NO_SCREEN": "1"})
