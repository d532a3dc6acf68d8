"""Is there a launch size beyond which the fused kernels slow down (round 4: "950 000 tiles: 82 %, 4.3 M: 54 %")?  The
2,504 x 531,500 and 2,504 x 200,000 all-pairs runs with 8 (default), 4, 2 and 1 band launches, the count kernel's time, its share
of the ceiling and the shader clock its blocks ran at, three runs each.
  python tests/sweeps/band_launch_size.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
log = lambda m: print("[size] " + m, flush=True)
for nv in (531_500, 200_000):
    twk, _ = bench.cohort_twk(bench.KG["n_samples"], nv, log, **{k: v for k, v in bench.KG.items() if k not in ("n_samples", "n_variants")})
    for mode in (["-p"], ["-u"]):
        for launches in (8, 4, 2, 1):
            rows = []
            for _ in range(3):
                r = bench.run_cli(twk, mode + ["--engine-option", f"band_max_launches={launches}"], 64, "/tmp/band_launch_size.two")
                if "error" in r: log(str(r)); break
                rows.append(f"{r['count_kernel_ms']:.1f} ms in {r['count_launches']} ({100 * r['and_bcnt_ceiling_frac']:.1f} %, {r['shader_mhz']} MHz)")
            log(f"{nv} variants {' '.join(mode)} band_max_launches={launches}: " + "; ".join(rows))
