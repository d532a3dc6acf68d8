"""End-to-end `tomahawk calc` from a .twk on the GPU box: load (pread + zstd + device RLE inflate) vs compute.
  python tests/sweeps/e2e_cli_timings.py [n_variants_at_1M=8192]
Writes cohort-shaped inputs (hostlib.write_cohort_twk) under /tmp, runs the CLI, prints wall / load / engine times."""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tomahawk_amd import hostlib as H

CLI = H.CLI_PATH
threads = min(os.cpu_count() or 8, 64)


def run(tag, twk, args, env=None):
    out = f"/tmp/e2e_{tag}.two"
    t = time.time()
    r = subprocess.run([CLI, "calc", "-i", twk, "-o", out] + args, capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    wall = time.time() - t
    if r.returncode != 0:
        print(tag, "FAILED", r.stderr[-400:]); return
    log = r.stderr
    load = re.search(r"Unpacked and uploaded .* variants\. (\S+)", log)
    fin = re.search(r"Finished in (\S+)\. Variants: ([0-9,]+), genotypes: [0-9,]+, output: ([0-9,]+)", log)
    rate = re.search(r"\] ([0-9,]+) variants/s", log)
    eng = re.findall(r"count kernel ([0-9.e+]+) ms in (\d+) launches \(([0-9.e+-]+) % [^)]*\), math kernels ([0-9.e+]+) ms", log)
    eng += re.findall(r"(carrier-list kernel [0-9.e+]+ ms in \d+ launches over [0-9,]+ rare pairs)", log)
    eng += re.findall(r"(probe kernel [0-9.e+]+ ms in \d+ launches over [0-9,]+ rare x common pairs)", log)
    eng += re.findall(r"(\d+ launches fused count -> r2 screen, [0-9,]+ candidate pairs)", log)
    print(f"{tag}: wall {wall:.2f} s | load {load.group(1) if load else '?'} | compute+write {fin.group(1) if fin else '?'} | pairs {fin.group(2) if fin else '?'} | "
          f"records {fin.group(3) if fin else '?'} | {rate.group(1) if rate else '?'} pairs/s in the compute phase | engine {eng}", flush=True)
    for l in log.splitlines():
        if "[UNPACK]" in l and "batches" in l: print("    " + l.split("[UNPACK]")[1].strip(), flush=True)
    try: os.remove(out)
    except OSError: pass


M1 = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
twk = f"/tmp/cohort_1m_{M1}.twk"
if not os.path.exists(twk):
    t = time.time(); H.write_cohort_twk(twk, 1_000_000, M1, seed=11, n_threads=threads, block_size=128)
    print(f"wrote {twk}: {os.path.getsize(twk)/1e6:.0f} MB in {time.time()-t:.1f} s", flush=True)
subprocess.run([CLI, "calc", "-i", twk, "-o", "/tmp/e2e_warm.two", "-t", str(threads)], capture_output=True)   # untimed: the input was just written, its pages are still being flushed
for args in (["-t", str(threads)], ["-u", "-t", str(threads)], ["-p", "-t", str(threads)]):
    run(f"1M x {M1} cohort calc {' '.join(args[:1]) if args[0] != '-t' else '(default)'}", twk, args)
run(f"1M x {M1} cohort calc (default) WITHOUT the carrier lists (--engine-option lists=0)", twk, ["-t", str(threads), "--engine-option", "lists=0"])
run(f"1M x {M1} cohort calc -u WITHOUT the carrier lists (--engine-option lists=0)", twk, ["-u", "-t", str(threads), "--engine-option", "lists=0"])
run(f"1M x {M1} cohort calc -u, 2 driver threads on one GPU", twk, ["-u", "-t", str(threads), "--engine-option", "force_device=0"], {"TWK_HIP_GPUS": "2"})
for args in (["-t", str(threads)], ["-u", "-t", str(threads)]):
    run(f"1M x {M1} cohort calc {' '.join(args[:1]) if args[0] != '-t' else '(default)'} WITHOUT the r2 screen", twk, args, {"TWK_HIP_NO_SCREEN": "1"})

# the regime the r2 screen cannot help: a cut-off below its threshold (r2 >= 0.001 is where it switches on) - every pair of the
# rare-heavy input is contracted densely (the T2 row of SURVEY 8a: the reference's list kernels are O(carriers) there)
if M1 > 8192:
    twk_s = "/tmp/cohort_1m_8192.twk"
    if not os.path.exists(twk_s):
        H.write_cohort_twk(twk_s, 1_000_000, 8192, seed=11, n_threads=threads, block_size=128)
else:
    twk_s = twk
for args in (["-r", "0.0009", "-t", str(threads)], ["-r", "0.1", "-t", str(threads)]):
    run(f"1M x {min(M1, 8192)} cohort calc (default) {' '.join(args[:2])}" + (" [below the screen's threshold: every pair contracted]" if args[1] == "0.0009" else " [screen on]"), twk_s, args)

twk2 = "/tmp/kg_2504_200k.twk"
if not os.path.exists(twk2):
    t = time.time(); H.write_cohort_twk(twk2, 2504, 200_000, seed=12, n_threads=threads, block_size=500, spacing=100)
    print(f"wrote {twk2}: {os.path.getsize(twk2)/1e6:.0f} MB in {time.time()-t:.1f} s", flush=True)
run("2504 x 200k cohort calc -p -w 1000000", twk2, ["-p", "-w", "1000000", "-t", str(threads)])
run("2504 x 200k cohort calc -w 1000000 (default)", twk2, ["-w", "1000000", "-t", str(threads)])

# K1's rare x common path (probes) and the sorted set below the screen's own threshold: calc -r 0.0009 (every pair is in the band)
for extra, tag in (([], "default: merges + probes"), (["--engine-option", "probe=0"], "probe=0: merges only"), (["--engine-option", "lists=0"], "lists=0: dense")):
    run(f"1M x {M1} cohort calc -r 0.0009 ({tag})", twk, ["-r", "0.0009", "-t", str(threads)] + extra)
    run(f"1M x {M1} cohort calc (default r2 0.1) ({tag})", twk, ["-t", str(threads)] + extra)
    run(f"1M x {M1} cohort calc -u ({tag})", twk, ["-u", "-t", str(threads)] + extra)
