import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tomahawk_amd as T
N = 1_000_000
rng = np.random.default_rng(1)
def make_runs(mean):
    lens = []
    left = N
    ls = rng.geometric(1.0 / mean, size=int(N / mean * 1.3) + 10).astype(np.int64)
    ls = np.minimum(ls, 16383)
    cs = np.cumsum(ls); k = int(np.searchsorted(cs, N)); ls = ls[:k + 1]; ls[k] = N - (cs[k - 1] if k else 0)
    if ls[k] == 0: ls = ls[:k]
    a = (np.arange(len(ls)) & 1).astype(np.uint16); b = ((np.arange(len(ls)) >> 1) & 1).astype(np.uint16) & a
    words = (ls.astype(np.uint16) << 2) | (a << 1) | b
    return words.view(np.uint8), len(ls)
eng = T.HipLd(0)
for mean in (12, 200):
    raw, n_runs = make_runs(mean)
    buf = np.zeros(raw.size + 64, np.uint8); buf[3:3 + raw.size] = raw
    print(f"mean run {mean}: {n_runs} runs, {raw.size/1e3:.0f} KB per variant", flush=True)
    for count in (1, 16, 128, 512, 2048):
        eng.set_problem(N, count)
        desc = np.zeros(count, dtype=T.RLE_DESC_DTYPE); desc["offset"] = 3; desc["n_runs"] = n_runs; desc["width"] = 2
        meta = np.zeros(count, dtype=T.META_DTYPE); meta["ac"] = 5
        eng.upload_rle(buf, desc, meta)
        t = time.perf_counter(); eng.upload_rle(buf, desc, meta); dt = time.perf_counter() - t
        print(f"  count {count}: call {dt*1e3:.2f} ms", flush=True)
eng.close()
