"""Fisher's exact test on the survivors of a real run (2,504 samples x 200,000 cohort-shaped variants, calc -p -w 100000:
10 M tables), through the engine's production kernels (twk_hip_fisher_exact, one table per lane): prepare -> bins -> walks
(ld_math.hip.h) against the same walks in the order the records were appended, the bins on shuffled input, and the oracle
(kt_fisher_exact restated) on a sample, by size of P.
  python tests/sweeps/fisher_order_bench.py"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tomahawk_amd as T
from tomahawk_amd import hostlib as H
from oracle import oracle as O

threads = min(os.cpu_count() or 8, 64)
twk = "/tmp/kg_2504_200k.twk"
if not os.path.exists(twk):
    H.write_cohort_twk(twk, 2504, 200_000, seed=12, n_threads=threads, block_size=500, spacing=100)
out = "/tmp/fisher_bench.two"
r = subprocess.run([H.CLI_PATH, "calc", "-i", twk, "-o", out, "-p", "-w", "100000", "-t", str(threads)], capture_output=True, text=True)
assert r.returncode == 0, r.stderr[-500:]
recs, info = H.read_two(out)
os.remove(out)
cnt = np.round(recs["cnt"]).astype(np.int32)
tables = np.ascontiguousarray(np.stack([cnt[:, 0], cnt[:, 2], cnt[:, 1], cnt[:, 3]], axis=1))      # n11, n12 (REFALT slot), n21, n22
print(f"{len(tables):,} tables from calc -p -w 100000 on 2,504 samples x 200,000 cohort-shaped variants", flush=True)
eng = T.HipLd(0)
eng.set_problem(2504, 64)


def run(name, order, ordered):
    t = np.ascontiguousarray(tables[order])
    best = 1e9
    for _ in range(3):
        p, ms = eng.fisher_exact(t, ordered=ordered)
        best = min(best, ms)
    print(f"  {name:44s} {best:8.3f} ms  {len(t) / best / 1e3:8.1f} M tables/s", flush=True)
    return p


rng = np.random.default_rng(0)
ident = np.arange(len(tables))
p0 = run("walks in the order appended", ident, False)
p1 = run("walks in bin order (production)", ident, True)
print("  same P, bit for bit:", np.array_equal(p0, p1))
run("shuffled input, walks in bin order", rng.permutation(len(tables)), True)
idx = rng.choice(len(tables), 30000, replace=False)
want = np.array([O.fisher(*[int(x) for x in tables[i]])[2] for i in idx])
rel = np.abs(p1[idx] - want) / np.maximum(want, 5e-324)
print(f"against the oracle, {len(idx)} sampled tables:")
for lo, hi in ((1e-100, 2), (1e-250, 1e-100), (1e-290, 1e-250), (1e-300, 1e-290), (0, 1e-300)):
    m = (want >= lo) & (want < hi) & (want > 0)
    if m.any():
        print(f"  P in [{lo:g}, {hi:g}): {m.sum():6d} tables, max relative difference {rel[m].max():.3g}")
z = want == 0
print(f"  P = 0 in the oracle: {z.sum()} tables, device max {p1[idx][z].max() if z.any() else 0:.3g}")
eng.close()

# tables whose P runs through the underflow region: 2 x 1e6 haplotypes, balanced margins, n11 stepping away from independence
print("P across the underflow region (n = 2,000,000 haplotypes, margins 1e6 / 1e6):")
eng = T.HipLd(0)
eng.set_problem(1_000_000, 64)
ks = np.arange(12_600, 14_300, 25)
tabs = np.array([[500_000 + k, 500_000 - k, 500_000 - k, 500_000 + k] for k in ks], dtype=np.int32)
pl, _ = eng.fisher_exact(np.tile(tabs, (64, 1)))
for k, t, b in zip(ks, tabs, pl):
    w = O.fisher(*[int(x) for x in t])[2]
    if w < 1e-280 and (w > 0 or k % 100 == 0):
        print(f"  k={k}: oracle {w:.6e} | device {b:.6e} (relative difference {abs(b - w) / w if w > 0 else float('nan'):.2e})")
eng.close()
