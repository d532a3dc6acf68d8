#!/bin/bash
# The same fused band-launch run ten times over (2,504 x 200,000, all 2e10 pairs, r2 screen off): does its count kernel always take
# the same time?  (Rounds of measurements in round 4 saw 147 ms most of the time and 263-271 ms now and then, before the fused
# kernels lost their one scratch register.)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TWK_HIP_NO_SCREEN=1
for log2 in 19 21; do
	for i in 1 2 3 4 5 6 7 8 9 10; do
		$R/tomahawk_amd/bin/tomahawk calc -i /tmp/kg_2504_200k.twk -o /tmp/o.two -t 64 -p -r 0.8 --engine-option band_work_log2=$log2 2>&1 > /dev/null | grep "HIP\] count" | sed -e "s/.*count kernel \([0-9.]*\) ms in \([0-9]*\) launches (\([0-9.]*\) %.*ran at \([0-9]*\) MHz.*/log2=$log2 run $i: \1 ms in \2 launches, \3 %, \4 MHz/"
	done
done
