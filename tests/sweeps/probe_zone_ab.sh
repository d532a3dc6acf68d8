#!/bin/bash
# Zone x zone pairs as probes (default) against merges of two lists (--engine-option probe_zone=0): 1 M samples x 50 000
# cohort-shaped variants, default mode and -u; the engine's own kernel times.
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 - <<PY
import sys; sys.path.insert(0, "$R")
import bench
twk, _ = bench.cohort_twk(1_000_000, 50_000, print)
PY
F=$(ls /tmp/twk_bench_cohort_1000000_50000*.twk | head -1)
for args in "" "-u" "-r 0.0009" "-u -r 0.0009"; do
	for pz in 1; do
		for rep in 1 2; do $R/tomahawk_amd/bin/tomahawk calc -i $F -o /tmp/pz.two -t 64 $args --engine-option probe_zone=$pz > /dev/null 2> /tmp/pz.err; done
		echo "== calc $args probe_zone=$pz: $(grep -o 'Finished in [0-9.]*s' /tmp/pz.err) output $(grep -o 'output: [0-9,]*' /tmp/pz.err)"
		grep -o "count kernel [0-9.]* ms in [0-9]* launches\|carrier-list kernel.*" /tmp/pz.err | cut -c1-250
	done
done
rm -f /tmp/pz.two
