"""Which thread count / block size maximises the reference's own calc on this host?  
Runs oracle/_ref/tomahawk_ref calc -u on the first M variants of the bench's synthetic input at N = 1M.
  python tests/sweeps/cpu_baseline_threads.py [M=1500] [blocks=11,25,50] [threads=...]"""
import os, re, subprocess, sys, tempfile, time
sys.path.insert(0, '.')
from oracle import oracle as O
from tomahawk_amd import hostlib
N, M = 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 1500
cores = os.cpu_count()
blocks = (11, 25, 50) if len(sys.argv) <= 2 else tuple(int(b) for b in sys.argv[2].split(","))
threads = sorted({cores, cores // 2, cores // 4, 64, 32}) if len(sys.argv) <= 3 else [int(t) for t in sys.argv[3].split(",")]
print(f"usable CPUs (affinity, cgroup quota): {hostlib.usable_cpus()} of {cores}", flush=True)
for block in blocks:
    twk = os.path.join(tempfile.gettempdir(), f"cpu_sweep_{M}_{block}.twk")
    if not os.path.exists(twk):
        hostlib.write_synthetic_twk(twk, N, M, seed=42, phased=False, block_size=block, n_threads=min(cores, 32))
    for t in threads:
        if t < 1 or t > cores:
            continue
        out = os.path.join(tempfile.gettempdir(), "cpu_sweep.two")
        t0 = time.time()
        r = subprocess.run([O.REF_BIN, "calc", "-i", twk, "-o", out, "-u", "-t", str(t)], capture_output=True, text=True)
        wall = time.time() - t0
        mo = re.search(r"\] ([0-9,]+) variants/s", r.stderr)
        print(f"block={block} threads={t}: {mo.group(1) if mo else '?'} pairs/s (ticker), {M*(M-1)//2/wall:,.0f} pairs/s (wall {wall:.1f}s)", flush=True)
