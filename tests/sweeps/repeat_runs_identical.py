import hashlib, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tomahawk_amd import hostlib as H
cases = []
t1 = "/tmp/kg_2504_60k.twk"
if not os.path.exists(t1): H.write_cohort_twk(t1, 2504, 60_000, seed=12, n_threads=64, block_size=500, spacing=100)
cases.append((t1, ["-p", "-w", "1000000", "-t", "64"], 12))
cases.append((t1, ["-w", "300000", "-t", "17"], 10))
cases.append((t1, ["-p", "-r", "0.05", "-t", "64"], 8))        # all pairs: band launches last band first, the hand-off queue
cases.append((t1, ["-u", "-r", "0.2", "-t", "64"], 6))
t2 = "/tmp/c5k.twk"
if not os.path.exists(t2): H.write_cohort_twk(t2, 300_000, 6000, seed=11, n_threads=64, block_size=128)
cases.append((t2, ["-t", "64"], 15))
cases.append((t2, ["-u", "-t", "8"], 10))
for twk, flags, reps in cases:
    hashes = {}
    t0 = time.time()
    for i in range(reps):
        out = "/tmp/soak.two"
        r = subprocess.run([H.CLI_PATH, "calc", "-i", twk, "-o", out] + flags, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-500:]
        recs, info = H.read_two(out); h = hashlib.md5(recs.tobytes()).hexdigest() + str(info["n_blocks"])
        hashes[h] = hashes.get(h, 0) + 1
    print(os.path.basename(twk), flags, f"{reps} runs in {time.time()-t0:.1f} s ->", len(hashes), "distinct output file(s)", os.path.getsize(out) // 1000000, "MB", flush=True)
