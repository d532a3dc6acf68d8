"""The outlier watch under load: the same all-pairs run 200 times in one engine context, every count launch logged
(twk_hip_launch_log) and compared with its peers; launches > 1.4 x the median cost of their kind are printed by the engine
itself (option timeline = 1) with the clock their blocks ran at and how far apart the XCDs finished.
    python tests/sweeps/launch_repeat.py [repeats=200]
Round 4 saw a 2,504 x 200,000 run take 263-271 ms of count kernel where it takes 147 ms (profiles/r04_small_n_ab.txt); this
is the hunt for such a launch with instruments on."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import tomahawk_amd as T

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
eng = T.HipLd(0)
eng.set_option("timeline", int(os.environ.get("TIMELINE", "0")))       # 1: the engine prints flagged launches (and the host's steps) as they happen
for name, N, M, mode, n in (("2,504 x 200,000 -p", 2504, 200_000, T.MODE_PHASED, reps), ("2,504 x 200,000 -u", 2504, 200_000, T.MODE_UNPHASED, reps // 2),
                            ("1,000,000 x 12,288 -u", 1_000_000, 12_288, T.MODE_UNPHASED, max(3, reps // 20))):
    eng.set_problem(N, M)
    eng.generate_synthetic(42)
    f = T.Filters(minR2=0.8)
    eng.ld_all(mode, f)                       # warm-up (plane sets, buffers, clock)
    eng.timing_reset()
    per_run, t0 = [], time.time()
    for r in range(n):
        before = eng.timing()["count_ms"]
        eng.ld_all(mode, f)
        per_run.append(eng.timing()["count_ms"] - before)
    tm = eng.timing()
    stats, seen = eng.launch_log()
    cost = np.array([x["ms"] / (x["row_pairs"] * x["words_per_row"] * (0.8125 if x["kind"] in (1, 4) else 1.0)) for x in stats if x["row_pairs"] and x["ms"] >= 0.3])
    pr = np.array(per_run)
    print(f"{name}: {n} runs in {time.time() - t0:.1f} s, {seen} count launches; count kernel per run min {pr.min():.2f} median {np.median(pr):.2f} max {pr.max():.2f} ms "
          f"(max / median {pr.max() / np.median(pr):.3f}); launch cost max / median {cost.max() / np.median(cost):.3f}, min / median {cost.min() / np.median(cost):.3f}; "
          f"outliers flagged {tm['outlier_launches']}; shader clock {min(x['shader_mhz'] for x in stats):.0f} .. {max(x['shader_mhz'] for x in stats):.0f} MHz; "
          f"XCD finish spread max {max(x['xcd_finish_spread_us'] for x in stats):.1f} us", flush=True)
    slow = [i for i, x in enumerate(per_run) if x > 1.2 * np.median(pr)]
    if slow:
        print(f"  runs more than 1.2 x the median: {[(i, round(per_run[i], 2)) for i in slow[:40]]}")
        med = np.median(cost)
        rows = [(i, x) for i, x in enumerate(stats) if x["row_pairs"] and x["ms"] >= 0.3 and x["ms"] / (x["row_pairs"] * x["words_per_row"] * (0.8125 if x["kind"] in (1, 4) else 1.0)) > 1.2 * med]
        print(f"  launches more than 1.2 x the median cost: {len(rows)}; launch #, ms, cost / median, shader MHz, XCD finish spread us, candidates:")
        for i, x in rows[:60]:
            cst = x["ms"] / (x["row_pairs"] * x["words_per_row"] * (0.8125 if x["kind"] in (1, 4) else 1.0))
            print(f"    #{i:5d} {x['ms']:8.3f} ms  {cst / med:5.2f} x  {x['shader_mhz']:6.0f} MHz  {x['xcd_finish_spread_us']:7.1f} us  {x['candidates']}")
        ok = [x for i, x in enumerate(stats) if x["ms"] >= 0.3 and all(i != j for j, _ in rows)]
        if ok:
            print(f"  the other launches: shader clock {min(x['shader_mhz'] for x in ok):.0f} .. {max(x['shader_mhz'] for x in ok):.0f} MHz; "
                  f"cost x clock of the slow launches relative to the median launch: {np.median([x['ms'] / (x['row_pairs'] * x['words_per_row'] * (0.8125 if x['kind'] in (1, 4) else 1.0)) * x['shader_mhz'] for _, x in rows]) / np.median([x['ms'] / (x['row_pairs'] * x['words_per_row'] * (0.8125 if x['kind'] in (1, 4) else 1.0)) * x['shader_mhz'] for x in ok]):.3f}")
eng.close()
