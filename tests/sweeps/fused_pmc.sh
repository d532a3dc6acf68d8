#!/bin/bash
# SQ counter passes over the *fused* count kernels at the reference's published shape (2,504 samples), all pairs (r2 band off, -r 0.8:
# few candidates) and a 1 Mb window: where do the wave cycles go that the K loop does not use?  (round-4 verdict, weak #2)
#   tests/sweeps/fused_pmc.sh <out-dir>        (run through gpurun from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$R/gpurun_out/r05}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
[ -f /tmp/kg_2504_200k.twk ] || python3 - <<PY
import sys; sys.path.insert(0, "$R")
from tomahawk_amd import hostlib as H
H.write_cohort_twk("/tmp/kg_2504_200k.twk", 2504, 200_000, seed=12, n_threads=32, block_size=500, spacing=100)
PY
export TWK_HIP_NO_SCREEN=1
run() {  # run <tag> <flags...>: three timed runs, then the counter passes
	local tag=$1; shift
	for i in 1 2 3; do
		$R/tomahawk_amd/bin/tomahawk calc -i /tmp/kg_2504_200k.twk -o /tmp/o.two -t 32 "$@" 2>&1 > /dev/null | grep "HIP\] count" | sed -e "s/.*\] count/$tag run $i: count/" | cut -c1-330
	done
	for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
	           "SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_SMEM"; do
		rm -rf /tmp/pmc_f
		rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_f -o f -- $R/tomahawk_amd/bin/tomahawk calc -i /tmp/kg_2504_200k.twk -o /tmp/o.two -t 32 "$@" > /dev/null 2>&1
		f=$(find /tmp/pmc_f -name "*counter_collection.csv" | head -1)
		[ -n "$f" ] && python3 $R/profiles/sum_counters.py "$f" | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if 'k_count' in k: print('$tag', k, json.dumps(v))"
	done
}
run "p_all" -p -r 0.8
run "u_all" -u -r 0.8
run "u_all_three0" -u -r 0.8 --engine-option three=0
run "p_w1m" -p -w 1000000
run "u_w1m" -u -w 1000000
