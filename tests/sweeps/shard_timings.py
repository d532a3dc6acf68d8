import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tomahawk_amd as T
hip = T.HipLd(0)
hip.set_problem(1_000_000, 50_000)
hip.generate_synthetic(42)
f = T.Filters()
hip.ld_all(T.MODE_UNPHASED, f, part=7, n_parts=8, collect=False)   # warm-up
for n_parts in (8, 4, 2):
    ts = []
    for k in range(n_parts):
        hip.ld_all(T.MODE_UNPHASED, f, part=k, n_parts=n_parts, collect=False)        # warm-up: a rank sizes its buffers once, outside the timed steps
        t = time.perf_counter(); r = hip.ld_all(T.MODE_UNPHASED, f, part=k, n_parts=n_parts, collect=False); ts.append(time.perf_counter() - t)
    tot = 50_000 * 49_999 // 2
    print(n_parts, "shards:", " ".join("%.3f" % x for x in ts), "-> max %.3f, projected %.4g pairs/s, balance %.3f" % (max(ts), tot / max(ts), sum(ts) / len(ts) / max(ts)), flush=True)
