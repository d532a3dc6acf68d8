#!/usr/bin/env python3
"""profiles/<tag>_pmc_hbm_traffic.json from the counter passes collect.sh made (FETCH_SIZE and WRITE_SIZE in separate
runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane streaming reads on gfx950).
  python profiles/make_traffic_summary.py r02"""
import json, os, sys
tag = sys.argv[1]
here = os.path.dirname(os.path.abspath(__file__))
ld = lambda n: json.load(open(os.path.join(here, f"{tag}_{n}")))
fetch, write, run = ld("cfg3_fetch_pmc_sums.json"), ld("cfg3_write_pmc_sums.json"), ld("cfg3_fetch_pmc.json")
k = next(x for x in fetch if "k_count_list" in x)
launches = fetch[k]["launches"]
fetch_b = fetch[k]["FETCH_SIZE"] * 1024 * 2           # KiB -> bytes, x2 (gfx950 correction)
write_b = write[k]["WRITE_SIZE"] * 1024
kernel_s = run["kernel_ms"]["count"] * 1e-3
pairs = run["value"] * run["ms_per_step"] * 1e-3       # variant pairs of the step
n = run["config"]["n_samples"]
alg = pairs * n / 4.0
out = {
 "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- python3 bench.py --steps 1 --warmup 0 --variants 16384 --no-cpu-baseline  (profiles/collect.sh cfg3_pmc)",
 "workload": run["config"]["workload"],
 "kernel": k, "launches": launches, "kernel_seconds": kernel_s,
 "hbm_read_bytes": fetch_b, "hbm_write_bytes": write_b,
 "hbm_read_GBps": fetch_b / kernel_s / 1e9, "hbm_write_GBps": write_b / kernel_s / 1e9,
 "traffic_bytes_per_launch": (fetch_b + write_b) / launches,
 "algorithmic_bytes (pairs x N/4)": alg, "traffic_over_algorithmic": (fetch_b + write_b) / alg,
 "compulsory_input_bytes (every plane row once)": run["config"]["n_variants"] * 2 * ((n + 31) // 32) * 4,
}
json.dump(out, open(os.path.join(here, f"{tag}_pmc_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
