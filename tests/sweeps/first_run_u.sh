#!/bin/bash
# Why is the first `calc -u` on a box 4 s slower than the second?  1 M samples x 50 000 cohort-shaped variants (bench.py's e2e
# input): one default-mode run, then `-u` three times, each with the host time line (--engine-option timeline=1) and the log's
# own stamps.
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 - <<PY
import sys; sys.path.insert(0, "$R")
import bench
twk, _ = bench.cohort_twk(1_000_000, 50_000, print)
print("FILE", twk)
PY
F=$(ls /tmp/twk_bench_cohort_1000000_50000*.twk | head -1)
$R/tomahawk_amd/bin/tomahawk calc -i $F -o /tmp/fr.two -t 64 > /dev/null 2> /tmp/fr0.err
echo "== default mode: $(grep -o 'Finished in [0-9.]*s' /tmp/fr0.err)"
for rep in 1 2 3; do
	$R/tomahawk_amd/bin/tomahawk calc -i $F -o /tmp/fr.two -t 64 -u --engine-option timeline=1 > /dev/null 2> /tmp/fr.err
	echo "== -u run $rep: $(grep -o 'Finished in [0-9.]*s' /tmp/fr.err)"
	grep "timeline\]" /tmp/fr.err | grep -v "enqueue\|inside" | cut -c1-160
	grep "LOG\]\[HIP\|Unpacked" /tmp/fr.err | cut -c1-400
done
rm -f /tmp/fr.two
