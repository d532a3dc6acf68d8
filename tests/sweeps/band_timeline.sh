#!/bin/bash
# Host timeline (--engine-option timeline=1) of `calc -p`, `-p -w 4000000` and `-u` over the reference's published shape (2,504 x 531,500)
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 - <<PY
import sys; sys.path.insert(0, "$R")
import bench
big, _ = bench.cohort_twk(bench.KG["n_samples"], bench.KG["n_variants"], print, **{k: v for k, v in bench.KG.items() if k not in ("n_samples", "n_variants")})
print("FILE", big)
PY
F=$(ls /tmp/twk_bench_cohort_2504_531500_*.twk | head -1)
for args in "-p" "-p -w 4000000" "-u"; do
	for rep in 1 2; do $R/tomahawk_amd/bin/tomahawk calc -i $F -o /tmp/tl.two -t 64 $args --engine-option timeline=1 > /tmp/tl.out 2> /tmp/tl.err; done
	echo "== calc $args"; grep "timeline\|Finished\|HIP\]\|handover" /tmp/tl.err | cut -c1-260
done
