#!/bin/bash
# Per-launch durations of the fused count kernel for `calc -p -r 0.8` over all 2e10 pairs of the 2,504 x 200,000 input
# (r2 screen off), band launches of several sizes against matrix-sized tiles: which launch is the slow one?
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
python3 - <<PY
import sys, os
sys.path.insert(0, "$R")
from tomahawk_amd import hostlib as H
if not os.path.exists("/tmp/kg_2504_200k.twk"):
    H.write_cohort_twk("/tmp/kg_2504_200k.twk", 2504, 200_000, seed=12, n_threads=64, block_size=500, spacing=100)
PY
export TWK_HIP_NO_SCREEN=1
for opts in "band_work_log2=19" "band_work_log2=20" "band_work_log2=21" "band_work_log2=19 --engine-option skip_pad=0" "band_launch=0"; do
	tag=$(echo "$opts" | tr -c 'a-z0-9_=' '_')
	rm -rf /tmp/bt_$tag
	timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/bt_$tag -o t -- $R/tomahawk_amd/bin/tomahawk calc -i /tmp/kg_2504_200k.twk -o /tmp/o.two -t 64 -p -r 0.8 --engine-option $opts > /dev/null 2> /tmp/bt_$tag.log
	echo "== $opts"; grep "HIP\]" /tmp/bt_$tag.log | cut -c1-260
	f=$(find /tmp/bt_$tag -name "*kernel_trace.csv" | head -1)
	python3 - "$f" <<PY
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_count_screen" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print("   launches %d, ms each: %s" % (len(d), " ".join("%.1f" % x for x in d[:40])))
PY
done
# is the slow million-tile launch about its tiles or about the buffers allocated for it?  The same 2 launches with a small candidate list
for opts in "band_work_log2=21 --engine-option band_list_entries=4194304" "band_work_log2=21"; do
	timeout 300 $R/tomahawk_amd/bin/tomahawk calc -i /tmp/kg_2504_200k.twk -o /tmp/o.two -t 64 -p -r 0.8 --engine-option $opts 2>&1 > /dev/null | grep "HIP\]" | cut -c1-260 | sed "s/^/== $opts: /"
done
