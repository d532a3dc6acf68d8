"""A/B of the .two writer on this box: the record emitter with a mapped output (frames copied into a shared mapping by
the workers, in parallel) against the stream output (one append at a time) - the same records, byte-identical files.
  python tests/sweeps/writer_ab.py [n_records=33000000] [n_threads=32]"""
import sys, time, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tomahawk_amd as T
from tomahawk_amd import hostlib as H
M = 200_000; n = int(sys.argv[1]) if len(sys.argv) > 1 else 33_000_000
rng = np.random.default_rng(1)
recs = np.zeros(n, dtype=T.RECORD_DTYPE)
a = np.sort(rng.integers(0, M - 1, n).astype(np.uint32)); recs["idxA"] = a
recs["idxB"] = np.minimum(a + 1 + rng.integers(0, 5000, n).astype(np.uint32), M - 1)
order = np.lexsort((recs["idxB"], recs["idxA"])); recs = recs[order]
for f in ("D", "Dprime", "R", "R2", "P", "ChiSqFisher", "ChiSqModel"): recs[f] = rng.random(n)
recs["cnt"] = rng.integers(0, 5000, (n, 4))
pos = (1000 + 100 * np.arange(M)).astype(np.uint32); rid = np.zeros(M, np.uint32)
for mo in (True, False, True, False):
    path = "/tmp/wbench.two"
    if os.path.exists(path): os.remove(path)
    t = time.time()
    st = H.TwoStream(path, 2504, rid, pos, n_threads=int(sys.argv[2]) if len(sys.argv) > 2 else 32, map_output=mo)
    for k in range(0, n, 1 << 20): st.append(recs[k:k + (1 << 20)])
    nw = st.close()
    dt = time.time() - t
    print(f"map_output={mo}: {nw} records, {os.path.getsize(path)/1e6:.0f} MB in {dt:.2f} s", flush=True)
    import hashlib; print("  sha", hashlib.sha256(open(path, "rb").read()).hexdigest()[:16])
