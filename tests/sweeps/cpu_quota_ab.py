"""The container's CPU quota against the emitter's thread count: `tomahawk calc` on the 2,504-sample inputs with
emit_workers = auto (min(-t, 32, usable CPUs)) / 8 / 12 / 14 / 16 / 24 / 32 / 64, printing compute + write and how often the
cgroup was throttled during the run (cpu.stat nr_throttled / throttled_usec).
  python tests/sweeps/cpu_quota_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tomahawk_amd import hostlib as H
log = lambda m: print("[quota] " + m, flush=True)


def cpu_stat():
    out = {}
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, v = line.split()
            out[k] = int(v)
    except OSError:
        pass
    return out


log(f"usable CPUs {H.usable_cpus()} of {os.cpu_count()}; cpu.max: " + (open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a"))
big, _ = bench.cohort_twk(bench.KG["n_samples"], bench.KG["n_variants"], log, **{k: v for k, v in bench.KG.items() if k not in ("n_samples", "n_variants")})
mid = "/tmp/kg_2504_200k.twk"
if not os.path.exists(mid):
    H.write_cohort_twk(mid, 2504, 200_000, seed=12, n_threads=64, block_size=500, spacing=100)
for workers in (0, 8, 12, 14, 16, 24, 32, 64):
    for twk, flags in ((big, ["-p"]), (big, ["-p", "-w", "4000000"]), (mid, ["-p", "-w", "1000000"])):
        best = None
        for _ in range(2):
            a = cpu_stat()
            r = bench.run_cli(twk, flags + (["--engine-option", f"emit_workers={workers}"] if workers else []), 64, "/tmp/quota_ab.two")
            b = cpu_stat()
            r["throttled"] = b.get("nr_throttled", 0) - a.get("nr_throttled", 0)
            r["throttled_ms"] = (b.get("throttled_usec", 0) - a.get("throttled_usec", 0)) / 1e3
            r["cpu_s"] = (b.get("usage_usec", 0) - a.get("usage_usec", 0)) / 1e6
            if best is None or r["compute_write_s"] < best["compute_write_s"]: best = r
        log(f"emit_workers={workers or 'auto'} {'200k' if twk == mid else '531.5k'} {' '.join(flags)}: wall {best['wall_s']:.2f} load {best['load_s']} compute+write {best['compute_write_s']:.3f} "
            f"throttled {best['throttled']} periods / {best['throttled_ms']:.0f} ms of thread time, {best['cpu_s']:.1f} CPU-s; records {best['records']}")
