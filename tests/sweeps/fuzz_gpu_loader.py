"""Byte-level mutations of a .twk through `tomahawk calc` on the GPU box: the pipelined loader (decode threads, record
walk, staging ring) and the device inflate kernels must end every run with exit code 0 or 1 - no crash, no hang."""
import os, random, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import util
from tomahawk_amd import hostlib
tmp = "/tmp/fuzz_gpu"; os.makedirs(tmp, exist_ok=True)
N, M = 700, 900
al = util.mosaic_alleles(M, N, 5, n_founders=6, switch=0.02, mut=0.004, miss_rate=0.03, miss_variants=0.3)
pos = (1000 + 13 * np.arange(M)).astype(np.uint32); rid = (np.arange(M) // 450).astype(np.uint32)
seed = f"{tmp}/seed.twk"
hostlib.write_twk(seed, al, pos, rid, phased=np.ones(M, np.uint8), n_contigs=2, block_size=64)
data = open(seed, "rb").read()
rng = random.Random(3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
codes = {}
for it in range(n):
    b = bytearray(data)
    k = rng.randrange(4)
    lo = 64 + rng.randrange(len(b) - 128)
    if k == 0:
        for _ in range(rng.randrange(1, 40)): b[lo + rng.randrange(60)] = rng.randrange(256)
    elif k == 1:
        b[lo] ^= 1 << rng.randrange(8)
    elif k == 2:
        del b[lo:lo + rng.randrange(1, 200)]
    else:
        b[lo:lo] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 64)))
    p = f"{tmp}/m.twk"; open(p, "wb").write(bytes(b))
    try:
        r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", p, "-o", f"{tmp}/o.two", "-t", "8"], capture_output=True, text=True, timeout=60)
        rc = r.returncode
    except subprocess.TimeoutExpired:
        rc = "timeout"
    codes[rc] = codes.get(rc, 0) + 1
    if rc not in (0, 1):
        print("BAD", it, rc, r.stderr[-400:] if rc != "timeout" else "")
print("runs", n, "exit codes", codes)
