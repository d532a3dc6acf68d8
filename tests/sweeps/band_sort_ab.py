"""The reference's published workload shape (2,504 x 531,500, calc -p, -p -w 4000000, -u) through the CLI: the band launches'
order (band_reverse) and the hand-off queue between the engine's thread and the emitter (emit_queue_pieces; 0: none) A/B.
  python tests/sweeps/band_sort_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
threads = 64
log = lambda m: print("[band_sort] " + m, flush=True)
big, _ = bench.cohort_twk(bench.KG["n_samples"], bench.KG["n_variants"], log, **{k: v for k, v in bench.KG.items() if k not in ("n_samples", "n_variants")})
mid = "/tmp/kg_2504_200k.twk"
if not os.path.exists(mid):
    from tomahawk_amd import hostlib as H
    H.write_cohort_twk(mid, 2504, 200_000, seed=12, n_threads=threads, block_size=500, spacing=100)
for rev, pieces in ((1, 32), (1, 0), (0, 32), (0, 0), (1, 8), (1, 64)):
    for twk, flags in ((big, ["-p"]), (big, ["-p", "-w", "4000000"]), (big, ["-u"]), (mid, ["-p", "-w", "1000000"])):
        best = None
        for _ in range(2):
            r = bench.run_cli(twk, flags + ["--engine-option", f"band_reverse={rev}", "--engine-option", f"emit_queue_pieces={pieces}"], threads, "/tmp/band_sort_ab.two")
            if best is None or r["compute_write_s"] < best["compute_write_s"]: best = r
        log(f"band_reverse={rev} emit_queue_pieces={pieces} {'200k' if twk == mid else '531.5k'} {' '.join(flags)}: wall {best['wall_s']:.2f} compute+write {best['compute_write_s']:.3f} count {best.get('count_kernel_ms')} ms in {best.get('count_launches')} "
            f"math {best.get('math_kernels_ms')} handover {best['producer_handover_s']} records {best['records']}")
