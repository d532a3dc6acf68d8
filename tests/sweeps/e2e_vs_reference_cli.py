import sys, os, subprocess, time, numpy as np
sys.path.insert(0, '.')
from tests import util
from tomahawk_amd import hostlib as H
from oracle import oracle as O
N, M = 2504, 3000
al = util.mosaic_alleles(M, N, 4242, n_founders=8, switch=0.01, mut=0.002, miss_rate=0.01, miss_variants=0.15)
pos = (1000 + 37 * np.arange(M)).astype(np.uint32)
rid = (np.arange(M) >= 2000).astype(np.uint32); pos[2000:] -= pos[2000] - 500
H.write_twk("/tmp/e2e.twk", al, pos, rid, phased=np.ones(M, np.uint8), n_contigs=2, block_size=200)
T = H.CLI_PATH; R = O.REF_BIN
def key(r): return np.lexsort((r["packB"], r["ridB"], r["packA"], r["ridA"]))
for flags in (["-p"], ["-u"], [], ["-p", "-r", "0.8"], ["-u", "-P", "1e-10"], ["-d", "0.9"] if False else ["-r", "0.3", "-u"]):
    t = time.time(); subprocess.run([R, "calc", "-i", "/tmp/e2e.twk", "-o", "/tmp/e2e_ref.two", "-t", "64"] + flags, check=True, capture_output=True, stdin=subprocess.DEVNULL); tr = time.time() - t
    t = time.time(); subprocess.run([T, "calc", "-i", "/tmp/e2e.twk", "-o", "/tmp/e2e_my.two"] + flags, check=True, capture_output=True, stdin=subprocess.DEVNULL); tm = time.time() - t
    a, _ = H.read_two("/tmp/e2e_ref.two"); b, _ = H.read_two("/tmp/e2e_my.two")
    a = a[key(a)]; b = b[key(b)]
    ka = np.stack([a["ridA"], a["packA"] >> 2, a["ridB"], a["packB"] >> 2], 1); kb = np.stack([b["ridA"], b["packA"] >> 2, b["ridB"], b["packB"] >> 2], 1)
    same_keys = ka.shape == kb.shape and np.array_equal(ka, kb)
    msg = f"flags {flags}: ref {len(a)} recs {tr:.1f}s, mine {len(b)} recs {tm:.1f}s, same pair set: {same_keys}"
    if same_keys:
        ok = np.array_equal(a["controller"], b["controller"])
        worst = {}
        for f in ("D", "Dprime", "R", "R2", "P", "ChiSqFisher"):
            d = np.abs(a[f] - b[f]) / np.maximum(np.abs(a[f]), 1e-300)
            d[(np.abs(a[f]) < 1e-12)] = 0
            worst[f] = float(d.max()) if len(d) else 0.0
        cnt = float(np.abs(a["cnt"] - b["cnt"]).max()) if len(a) else 0.0
        msg += f", flags equal {ok}, max |dcnt| {cnt:.3g}, worst rel {worst}"
    else:
        sa = set(map(tuple, ka.tolist())); sb = set(map(tuple, kb.tolist()))
        msg += f", only ref {len(sa - sb)}, only mine {len(sb - sa)}: {sorted(sa - sb)[:3]} {sorted(sb - sa)[:3]}"
    print(msg, flush=True)
