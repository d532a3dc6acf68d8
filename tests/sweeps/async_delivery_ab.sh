#!/bin/bash
# The delivery thread (engine option async_delivery: finished launches' survivors copied aside on the device, a second thread takes them to the host
# while the caller's thread keeps enqueueing launches) off and on, over the reference's published shape (2,504 x 531,500): all pairs -p / -u and the 4 Mb window, with and without the
# record codec; three runs each.
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 - <<PY
import sys; sys.path.insert(0, "$R")
import bench
big, _ = bench.cohort_twk(bench.KG["n_samples"], bench.KG["n_variants"], print, **{k: v for k, v in bench.KG.items() if k not in ("n_samples", "n_variants")})
PY
F=$(ls /tmp/twk_bench_cohort_2504_531500_*.twk | head -1)
for args in "-p" "-u" "-p -w 4000000"; do
for codec in 0 1; do
for early in 0 1; do
	for rep in 1 2 3; do rm -f /tmp/tl.two; $R/tomahawk_amd/bin/tomahawk calc -i $F -o /tmp/tl.two -t 64 $args --engine-option record_codec=$codec --engine-option async_delivery=$early > /tmp/tl.out 2> /tmp/tl.err; echo "calc $args record_codec=$codec async_delivery=$early: $(grep -o 'Finished in [0-9.]*s' /tmp/tl.err) $(grep -o 'count kernel [0-9.]* ms in [0-9]* launches' /tmp/tl.err) $(grep -o 'output: [0-9,]*' /tmp/tl.err) md5 $(tail -c +4096 /tmp/tl.two | md5sum | cut -c1-8)"; done
done
done
done
