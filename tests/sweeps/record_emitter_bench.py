import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tomahawk_amd as T
from tomahawk_amd import hostlib as H
n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else min(os.cpu_count(), 64)
tile = 465_000
M = 200_000
rng = np.random.default_rng(1)
rec = np.zeros(tile, dtype=T.RECORD_DTYPE)
pos = (1000 + 100 * np.arange(M)).astype(np.uint32); rid = np.zeros(M, np.uint32)
def fill(t):
    a = rng.integers(0, M - 2000, tile).astype(np.uint32)
    rec["idxA"] = a; rec["idxB"] = a + rng.integers(1, 2000, tile).astype(np.uint32)
    rec["flags"] = 3
    c = rng.integers(0, 5008, (tile, 4)).astype(np.float64); rec["cnt"] = c
    for f in ("D", "Dprime", "R", "R2", "P", "ChiSqFisher", "ChiSqModel"): rec[f] = rng.random(tile)
    if len(sys.argv) > 3:
        o = np.lexsort((rec["idxB"], rec["idxA"])); rec[:] = rec[o]
fill(0)
out = "/tmp/emit_bench.two"
st = H.TwoStream(out, 2504, rid, pos, n_threads=threads)
t = time.time(); n = 0
while n < n_total:
    st.append(rec); n += tile
w = st.close(); dt = time.time() - t
print(f"{n} survivors -> {w} records, {os.path.getsize(out)/1e6:.0f} MB in {dt:.2f} s: {n/dt/1e6:.1f} M survivors/s, {w*106/dt/1e9:.2f} GB/s of records ({threads} threads)")
os.remove(out)
