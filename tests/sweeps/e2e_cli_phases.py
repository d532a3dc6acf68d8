import os, re, subprocess, sys, time, datetime
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tomahawk_amd import hostlib as H
twk = "/tmp/cohort_1m_50000.twk"
if not os.path.exists(twk):
    t = time.time(); H.write_cohort_twk(twk, 1_000_000, 50000, seed=11, n_threads=64, block_size=128); print("wrote", time.time() - t, flush=True)
subprocess.run([H.CLI_PATH, "calc", "-i", twk, "-o", "/tmp/o.two", "-t", "64"], capture_output=True, text=True)
def ts(line):
    m = re.match(r"\[(\d+-\d+-\d+ \d+:\d+:\d+),(\d+)\]", line)
    return datetime.datetime.strptime(m.group(1), "%Y-%m-%d %H:%M:%S").timestamp() + int(m.group(2)) / 1000 if m else None
for rep in range(3):
    t0 = time.time()
    r = subprocess.run([H.CLI_PATH, "calc", "-i", twk, "-o", "/tmp/o.two", "-t", "64"], capture_output=True, text=True)
    t1 = time.time()
    stamps = [(ts(l), l) for l in r.stderr.splitlines() if ts(l)]
    first, last = stamps[0][0], stamps[-1][0]
    print(f"wall {t1-t0:.3f}: start->first log {first-t0:.3f}, first->last log {last-first:.3f}, last log->exit {t1-last:.3f}")
    for (a, l), (b, _) in zip(stamps, stamps[1:] + [(t1, "")]):
        if b - a > 0.03: print(f"   {b-a:.3f} s after: {l[:110]}")
