#!/bin/bash
# Is the survivor-rich small-N run bound by the file system?  (1) write_probe: what /tmp and /dev/shm take; (2) `calc -p` over the
# reference's published shape (2,504 x 531,500) writing to /tmp, to /dev/shm and to /dev/null (through a symlink).
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/build
g++ -O2 -std=c++17 -pthread $R/tomahawk_amd/csrc/tools/write_probe.cpp -o $R/build/write_probe || exit 1
for p in /tmp/wp.bin /dev/shm/wp.bin; do echo "== write_probe $p"; $R/build/write_probe $p 2 512; done
python3 - <<PY
import sys; sys.path.insert(0, "$R")
import bench
big, _ = bench.cohort_twk(bench.KG["n_samples"], bench.KG["n_variants"], print, **{k: v for k, v in bench.KG.items() if k not in ("n_samples", "n_variants")})
PY
F=$(ls /tmp/twk_bench_cohort_2504_531500_*.twk | head -1)
ln -sf /dev/null /tmp/null.two
for out in /tmp/of.two /dev/shm/of.two /tmp/null.two; do
	for args in "-p" "-p -w 4000000"; do
		for rep in 1 2; do
			[ $out != /tmp/null.two ] && rm -f $out
			$R/tomahawk_amd/bin/tomahawk calc -i $F -o $out -t 64 $args > /dev/null 2> /tmp/of.err
		done
		echo "== calc $args -o $out: $(grep -o 'Finished in [0-9.]*s' /tmp/of.err) $(grep -o 'handover[^;]*' /tmp/of.err | head -1)"
	done
done
rm -f /dev/shm/of.two
# (3) is it the compression or the one append stream?  The window run (count kernel 49 ms: nothing but the host's work) with
# 16 / 32 / 64 emitter threads, into a file and into /dev/null
for out in /tmp/of.two /tmp/null.two; do
	for w in 16 32 64; do
		for rep in 1 2; do
			[ $out != /tmp/null.two ] && rm -f $out
			$R/tomahawk_amd/bin/tomahawk calc -i $F -o $out -t 64 -p -w 4000000 --engine-option emit_workers=$w > /dev/null 2> /tmp/of.err
		done
		echo "== calc -p -w 4000000 -o $out emit_workers=$w: $(grep -o 'Finished in [0-9.]*s' /tmp/of.err) $(grep -o 'workers: expanding.*' /tmp/of.err | cut -c1-120)"
	done
done
rm -f /tmp/of.two
