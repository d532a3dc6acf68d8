import os, re, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tomahawk_amd import hostlib as H
twk2 = "/tmp/kg_2504_200k.twk"
if not os.path.exists(twk2):
    H.write_cohort_twk(twk2, 2504, 200_000, seed=12, n_threads=64, block_size=500, spacing=100)
for t in ("8", "16", "32", "64", "128"):
    for pre in ([], ["numactl", "--cpunodebind=0", "--membind=0"]):
        try:
            r = subprocess.run(pre + [H.CLI_PATH, "calc", "-i", twk2, "-o", "/tmp/o2.two", "-p", "-w", "1000000", "-t", t], capture_output=True, text=True)
        except FileNotFoundError:
            continue
        fin = re.search(r"Finished in (\S+)\.", r.stderr); w = re.search(r"the producer spent (.*)", r.stderr)
        print(f"-t {t} {'numa0' if pre else '     '}: finished {fin.group(1) if fin else '?'} | {w.group(1) if w else r.stderr[-200:]}", flush=True)
os.system("lscpu | grep -i 'numa\\|socket\\|model name' | head -8")
