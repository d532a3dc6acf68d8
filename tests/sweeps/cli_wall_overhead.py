"""Where does the wall time of a short `tomahawk calc` go outside its own "load" and "compute + write" phases?  Runs the
2,504 x 531,500 `-p` job (a) from a bare python process and (b) from one that holds a HIP context with 20 GB allocated
(what bench.py's extra legs run under), and prints, per run: process start -> first log line, the log's own phases,
last log line -> exit.
  python tests/sweeps/cli_wall_overhead.py"""
import os, re, subprocess, sys, time, datetime
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tomahawk_amd import hostlib
log = lambda m: print("[overhead] " + m, flush=True)
big, _ = bench.cohort_twk(bench.KG["n_samples"], bench.KG["n_variants"], log, **{k: v for k, v in bench.KG.items() if k not in ("n_samples", "n_variants")})
out = "/tmp/overhead.two"


def once(tag, flags):
    try: os.remove(out)
    except OSError: pass
    t0 = time.time()
    r = subprocess.run([hostlib.CLI_PATH, "calc", "-i", big, "-o", out, "-t", "64"] + flags, capture_output=True, text=True)
    t1 = time.time()
    stamps = [datetime.datetime.strptime(m, "%Y-%m-%d %H:%M:%S,%f").timestamp() for m in re.findall(r"^\[(\d{4}-\d\d-\d\d \d\d:\d\d:\d\d,\d{3})\]", r.stderr, re.M)]
    fin = re.search(r"Finished in (\S+)\.", r.stderr)
    load = re.search(r"Unpacked and uploaded .* variants\. (\S+)", r.stderr)
    log(f"{tag} {' '.join(flags)}: wall {t1 - t0:.2f} s = {stamps[0] - t0:.2f} s before the first log line + {stamps[-1] - stamps[0]:.2f} s of log "
        f"(load {load.group(1) if load else '?'}, compute+write {fin.group(1) if fin else '?'}) + {t1 - stamps[-1]:.2f} s after the last line")


for flags in (["-p"], ["-p", "-w", "4000000"]):
    once("bare parent", flags); once("bare parent", flags)
import torch
x = torch.empty(20 << 30, dtype=torch.uint8, device="cuda:0"); x.zero_(); torch.cuda.synchronize()
for flags in (["-p"], ["-p", "-w", "4000000"]):
    once("parent holding a HIP context + 20 GB", flags); once("parent holding a HIP context + 20 GB", flags)
