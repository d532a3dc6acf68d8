"""A/B of the probe kernels on the headline-size cohort run: one column per block (round 4, probe_cols=0) against a strip of
columns per block (k_probe_strip_t, the default):   python tests/sweeps/probe_cols_ab.py [n_variants]
`tomahawk calc` default mode (PhasedMath) and -u from a cohort-shaped 1,000,000-sample .twk; prints wall, compute + write,
count / probe / list kernel times and records per run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench

log = lambda m: print("[ab] " + m, flush=True)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
twk, threads = bench.cohort_twk(1_000_000, M, log)
O = lambda *kv: [x for k in kv for x in ("--engine-option", k)]
for flags, tag in (([], "default"), (["-u"], "-u")):
    for cols in ("lds", "lds"):
        r = bench.run_cli(twk, flags + (O("probe_lds=1") if cols == "lds" else O("probe_lds=0", f"probe_cols={cols}")), threads, "/tmp/ab.two")
        if "error" in r:
            log(f"{tag} probe_cols={cols}: FAILED {r['error']}"); continue
        log(f"{tag} probe_cols={cols}: wall {r['wall_s']:.2f} s, load {r['load_s']}, compute+write {r['compute_write_s']:.3f} s, count {r['count_kernel_ms']:.1f} ms in "
            f"{r['count_launches']} launches, probes {r['probe_kernel_ms']} ms over {r['pairs_decided_by_probes']} pairs, lists {r['list_kernel_ms']} ms, math {r['math_kernels_ms']:.1f} ms, records {r['records']}")
