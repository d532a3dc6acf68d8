#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 - <<PY
import sys; sys.path.insert(0, "$R")
import bench
big, _ = bench.cohort_twk(bench.KG["n_samples"], bench.KG["n_variants"], print, **{k: v for k, v in bench.KG.items() if k not in ("n_samples", "n_variants")})
PY
F=$(ls /tmp/twk_bench_cohort_2504_531500_*.twk | head -1)
for codec in 0 1; do
	for rep in 1 2; do $R/tomahawk_amd/bin/tomahawk calc -i $F -o /tmp/tl.two -t 64 -p --engine-option timeline=1 --engine-option record_codec=$codec --engine-option async_delivery=${EARLY:-1} > /tmp/tl.out 2> /tmp/tl.err; done
	echo "== calc -p record_codec=$codec"; grep "timeline\|Finished\|HIP\]\|handing" /tmp/tl.err | cut -c1-300
done
