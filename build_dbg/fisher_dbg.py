import os, subprocess, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import tomahawk_amd as T
from tomahawk_amd import hostlib as H
from oracle import oracle as O
twk = "/tmp/kg_2504_200k.twk"
if not os.path.exists(twk):
    H.write_cohort_twk(twk, 2504, 200_000, seed=12, n_threads=64, block_size=500, spacing=100)
out = "/tmp/fisher_bench.two"
r = subprocess.run([H.CLI_PATH, "calc", "-i", twk, "-o", out, "-p", "-w", "20000", "-t", "64"], capture_output=True, text=True)
recs, info = H.read_two(out)
cnt = np.round(recs["cnt"]).astype(np.int32)
tables = np.ascontiguousarray(np.stack([cnt[:, 0], cnt[:, 2], cnt[:, 1], cnt[:, 3]], axis=1))
eng = T.HipLd(0); eng.set_problem(2504, 64)
pg, _ = eng.fisher_exact(tables); pl, _ = eng.fisher_exact(tables, one_lane_per_table=True)
ok = pl > 1e-300
rel = np.zeros(len(pl)); rel[ok] = np.abs(pg[ok] - pl[ok]) / pl[ok]
order = np.argsort(-rel)[:12]
print("n tables", len(tables), "n with rel>1e-6:", (rel > 1e-6).sum())
for i in order:
    t = [int(x) for x in tables[i]]
    w = O.fisher(*t)[2]
    print(t, "group", pg[i], "lane", pl[i], "oracle", w)
