"""Fisher's exact test kernels in isolation (twk_hip_fisher_exact): the production kernel (16 lanes per table) against the
one-lane-per-table walk and the oracle, on the tables of a real survivor-heavy run and on tables whose P crosses the
underflow region.
  python tests/sweeps/fisher_kernel_bench.py [window_bp=50000]"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tomahawk_amd as T
from tomahawk_amd import hostlib as H
from oracle import oracle as O

threads = min(os.cpu_count() or 8, 64)
W = sys.argv[1] if len(sys.argv) > 1 else "50000"
twk = "/tmp/kg_2504_200k.twk"
if not os.path.exists(twk):
    H.write_cohort_twk(twk, 2504, 200_000, seed=12, n_threads=threads, block_size=500, spacing=100)
out = "/tmp/fisher_bench.two"
r = subprocess.run([H.CLI_PATH, "calc", "-i", twk, "-o", out, "-p", "-w", W, "-t", str(threads)], capture_output=True, text=True)
assert r.returncode == 0, r.stderr[-500:]
recs, info = H.read_two(out)
os.remove(out)
cnt = np.round(recs["cnt"]).astype(np.int32)
tables = np.ascontiguousarray(np.stack([cnt[:, 0], cnt[:, 2], cnt[:, 1], cnt[:, 3]], axis=1))      # n11, n12 (REFALT slot), n21, n22
print(f"{len(tables):,} tables from calc -p -w {W} on 2,504 samples x 200,000 cohort-shaped variants "
      f"(support width: median {int(np.median(np.minimum(tables[:,0]+tables[:,1], tables[:,0]+tables[:,2]) - np.maximum(0, tables[:,0]-tables[:,3])))}, "
      f"> 64 in {np.mean((np.minimum(tables[:,0]+tables[:,1], tables[:,0]+tables[:,2]) - np.maximum(0, tables[:,0]-tables[:,3])) > 64) * 100:.1f} %)", flush=True)

from scipy.special import gammaln
def log_q(t):
    n11, n12, n21, n22 = (t[:, k].astype(np.float64) for k in range(4))
    lb = lambda n, k: gammaln(n + 1) - gammaln(k + 1) - gammaln(n - k + 1)
    return lb(n11 + n12, n11) + lb(n21 + n22, n21) - lb(n11 + n12 + n21 + n22, n11 + n21)
lq = log_q(tables) / np.log(10)
print(f"  log10 q (the observed table's own probability): {np.mean(lq < -323.3) * 100:.1f} % underflow to 0, "
      f"{np.mean((lq >= -323.3) & (lq < -270)) * 100:.1f} % in (0, 1e-270) - the band left to the one-lane recurrence, "
      f"{np.mean((lq >= -270) & (lq < -100)) * 100:.1f} % in [1e-270, 1e-100), {np.mean(lq >= -100) * 100:.1f} % above", flush=True)

eng = T.HipLd(0)
eng.set_problem(2504, 64)
for rep in range(3):
    pg, ms_g = eng.fisher_exact(tables)
    pl, ms_l = eng.fisher_exact(tables, one_lane_per_table=True)
    print(f"  group kernel {ms_g:8.3f} ms ({len(tables) / ms_g / 1e3:7.2f} M tables/s) | lane kernel {ms_l:8.3f} ms ({len(tables) / ms_l / 1e3:7.2f} M tables/s)", flush=True)
ok = pl > 1e-300
print(f"  group vs lane: max relative difference {np.max(np.abs(pg[ok] - pl[ok]) / pl[ok]):.3g} over {ok.sum():,} tables with P > 1e-300; "
      f"{(~ok).sum():,} below, max absolute difference there {np.max(np.abs(pg[~ok] - pl[~ok])) if (~ok).any() else 0:.3g}")
idx = np.random.default_rng(1).choice(len(tables), size=min(3000, len(tables)), replace=False)
worst = 0.0
for i in idx:
    w = O.fisher(*[int(x) for x in tables[i]])[2]
    if w > 1e-300:
        worst = max(worst, abs(pg[i] - w) / w)
print(f"  group vs oracle (kt_fisher_exact restated, {len(idx)} sampled tables): max relative difference {worst:.3g}")

# tables whose P runs through the underflow region: 2 x 1e6 haplotypes, balanced margins, n11 stepping away from independence
print("P across the underflow region (n = 2,000,000 haplotypes, margins 1e6 / 1e6):")
eng.set_problem(1_000_000, 64)
ks = np.arange(12_600, 14_300, 25)
tabs = np.array([[500_000 + k, 500_000 - k, 500_000 - k, 500_000 + k] for k in ks], dtype=np.int32)
pg, _ = eng.fisher_exact(tabs)
pl, _ = eng.fisher_exact(tabs, one_lane_per_table=True)
for k, t, a, b in zip(ks, tabs, pg, pl):
    w = O.fisher(*[int(x) for x in t])[2]
    rel = lambda x: abs(x - w) / w if w > 0 else float("nan")
    if w < 1e-280:
        print(f"  k={k}: oracle {w:.6e} | group {a:.6e} (rel {rel(a):.2e}, abs {abs(a - w):.2e}) | lane {b:.6e} (rel {rel(b):.2e})")
eng.close()
