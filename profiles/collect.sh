#!/bin/bash
# Collect the measurements behind DESIGN.md on a GPU box (run through gpurun from the repo root):
#   profiles/collect.sh <tag> <stage>...      e.g.  profiles/collect.sh r02 microbench cfg2 cfg5 cfg3
# Raw outputs land in gpurun_out/<tag>/ ; the ones quoted in DESIGN.md are then copied to profiles/<tag>_*.
# rocprofv3 is always given the python3 / tool binary itself after `--` (no shell hop), counters in their
# own passes.
set -u
TAG=${1:?tag}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp

stats() {   # stats <name> <bench args...>: kernel trace + stats of one bench run, CSV summary kept
	local name=$1; shift
	rm -rf /tmp/prof_$name
	timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o $name -- \
		python3 $R/bench.py --no-cpu-baseline --no-e2e --no-traffic --no-extra --no-planted "$@" > $OUT/${name}_under_rocprof.json 2> $OUT/${name}_under_rocprof.log < /dev/null
	echo "$name stats rc=$?"
	local f; f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
	[ -n "$f" ] && cp "$f" $OUT/${name}_kernel_stats.csv && head -5 "$f" | cut -c1-220
}
pmc() {     # pmc <name> <counter list> <bench args...>: one counter pass, per-kernel sums
	local name=$1 ctr=$2; shift 2
	rm -rf /tmp/pmc_$name
	timeout 900 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_$name -o $name -- \
		python3 $R/bench.py --no-cpu-baseline --no-e2e --no-traffic --no-extra --no-planted "$@" > $OUT/${name}_pmc.json 2> $OUT/${name}_pmc.log < /dev/null
	echo "$name pmc rc=$?"
	local f; f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
	[ -n "$f" ] && python3 $R/profiles/sum_counters.py "$f" > $OUT/${name}_pmc_sums.json && cat $OUT/${name}_pmc_sums.json | head -60
}

for stage in "$@"; do
	case $stage in
	microbench)   # the instruction-rate evidence behind the VALU ceiling
		for t in valu_rate issue_test2 bank_test2 hbm_read_bw; do
			echo "== $t"; timeout 300 $R/build/$t > $OUT/microbench_$t.txt 2>&1; cat $OUT/microbench_$t.txt
		done
		echo "== count_microbench (kernel alone, full K of N = 1M unphased planes: 4096 rows x 31264 words)"
		timeout 300 $R/build/count_microbench 4096 31264 3 > $OUT/microbench_count_microbench.txt 2>&1; cat $OUT/microbench_count_microbench.txt
		echo "== count_microbench (configs[1] rows: 10112 rows x 6272 words)"
		timeout 300 $R/build/count_microbench 10112 6272 3 >> $OUT/microbench_count_microbench.txt 2>&1; tail -3 $OUT/microbench_count_microbench.txt
		echo "== per-block finish times of the ticketed kernel (2 rounds of whole-K tiles, N = 1M rows)"
		FINISH=1 timeout 300 $R/build/count_microbench 4096 31264 3 512 > $OUT/microbench_finish_times.txt 2>&1; head -3 $OUT/microbench_finish_times.txt
		;;
	cfg3)  timeout 600 python3 $R/bench.py --steps 2 --warmup 1 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.log; cat $OUT/bench_cfg3.json ;;
	cfg3_stats) stats cfg3 --steps 2 --warmup 1 ;;
	cfg3_sq)   # the work-is-done check of the headline kernel, every round: VALU instructions issued against the products + v_or the launches stand for
		pmc cfg3_sq "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" --steps 1 --warmup 0
		;;
	cfg3_lds)  # how busy the headline kernel keeps the LDS (what is left in the three-product loop)
		pmc cfg3_lds "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS_LOAD SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" --steps 1 --warmup 0
		;;
	cfg3_pmc)
		pmc cfg3_fetch FETCH_SIZE --steps 1 --warmup 0 --variants 16384
		pmc cfg3_write WRITE_SIZE --steps 1 --warmup 0 --variants 16384
		pmc cfg3_sq "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" --steps 1 --warmup 0 --variants 16384
		;;
	patch_pmc)   # HBM read traffic of the count kernel against the shape of the tile patches (engine options patch_rows / patch_cols)
		# and the length of the K segments the tiles of a patch advance by (seg, 0 = whole tiles), 16,384 variants at N = 1 M
		# and, with segments, one unit queue per XCD (xcd_queues=8: patches dealt round robin to the XCDs)
		for combo in ${PATCH_COMBOS:-8x8:0:0 16x32:0:0 8x8:64:0 16x32:64:0 8x8:64:8 8x8:32:8 8x8:16:8 8x8:8:8 4x16:16:8 8x16:16:8}; do
			shape=${combo%%:*}; rest=${combo#*:}; seg=${rest%%:*}; nq=${rest##*:}
			pmc patch_${shape}_seg${seg}_q$nq FETCH_SIZE --steps 1 --warmup 0 --variants 16384 --engine-option patch_rows=${shape%%x*} --engine-option patch_cols=${shape##*x} \
				--engine-option seg=$seg --engine-option xcd_queues=$nq > /dev/null
			python3 - <<PY
import json
f = json.load(open("$OUT/patch_${shape}_seg${seg}_q${nq}_pmc_sums.json")); r = json.load(open("$OUT/patch_${shape}_seg${seg}_q${nq}_pmc.json"))
k = next(x for x in f if "k_count_list" in x)
gb = f[k]["FETCH_SIZE"] * 1024 * 2 / 1e9
print("patch %-6s seg %3s queues %s FETCH_SIZE x2 = %8.1f GB over %d launches (%.1f GB/launch), count kernel %.1f ms, %.1f M pairs/s" % ("$shape", "$seg", "$nq", gb, f[k]["launches"], gb / f[k]["launches"], r["kernel_ms"]["count"], r["value"] / 1e6))
PY
		done
		;;
	cfg2)
		timeout 600 python3 $R/bench.py --config cfg2 --no-traffic --steps 20 --warmup 3 > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.log; cat $OUT/bench_cfg2.json
		stats cfg2 --config cfg2 --steps 20 --warmup 3
		pmc cfg2_sq "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" --config cfg2 --steps 5 --warmup 1
		;;
	cfg5)
		timeout 900 python3 $R/bench.py --config cfg5 --no-traffic --emulate-shard 3/8 --steps 1 --warmup 0 > $OUT/bench_cfg5_shard3of8.json 2> $OUT/bench_cfg5_shard3of8.log; cat $OUT/bench_cfg5_shard3of8.json
		;;
	cfg5_prof)
		stats cfg5_shard3of8 --config cfg5 --emulate-shard 3/8 --steps 1 --warmup 0
		pmc cfg5_sq "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" --config cfg5 --emulate-shard 3/8 --steps 1 --warmup 0
		;;
	load_prof)  # the CLI's input path: kernel + copy trace of `tomahawk calc` on a 1 M x 20 k cohort-shaped .twk
		python3 - <<PY
import sys, os, time
sys.path.insert(0, "$R")
from tomahawk_amd import hostlib as H
if not os.path.exists("/tmp/c20k.twk"):
    H.write_cohort_twk("/tmp/c20k.twk", 1_000_000, 20000, seed=11, n_threads=64, block_size=128)
PY
		rm -rf /tmp/prof_load
		timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_load -o load -- \
			$R/tomahawk_amd/bin/tomahawk calc -i /tmp/c20k.twk -o /tmp/o.two -t 64 > /dev/null 2> $OUT/load_under_rocprof.log
		f=$(find /tmp/prof_load -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/load_kernel_stats.csv && head -8 "$f" | cut -c1-160
		f=$(find /tmp/prof_load -name "*memory_copy_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/load_memory_copy_stats.csv && cat "$f" | cut -c1-160
		grep "UNPACK\|Unpacked" $OUT/load_under_rocprof.log | cut -c1-400
		;;
	math_prof)  # a survivor-heavy run: 2,504 samples x 200,000 cohort-shaped variants, -p -w 1000000 (33 M surviving pairs)
		python3 - <<PY
import sys, os
sys.path.insert(0, "$R")
from tomahawk_amd import hostlib as H
if not os.path.exists("/tmp/kg_2504_200k.twk"):
    H.write_cohort_twk("/tmp/kg_2504_200k.twk", 2504, 200_000, seed=12, n_threads=64, block_size=500, spacing=100)
PY
		rm -rf /tmp/prof_math
		timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_math -o m -- \
			$R/tomahawk_amd/bin/tomahawk calc -i /tmp/kg_2504_200k.twk -o /tmp/o.two -t 64 -p -w 1000000 > /dev/null 2> $OUT/kg_under_rocprof.log
		f=$(find /tmp/prof_math -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kg_kernel_stats.csv && head -8 "$f" | cut -c1-160
		grep "Finished\|HIP\]\|WRITER\]" $OUT/kg_under_rocprof.log | cut -c1-300
		;;
	t2_list)  # T2 evidence: a rare pair as a sorted carrier-list intersection against the dense contraction (csrc/tools/list_vs_dense.hip)
		: > $OUT/t2_list_vs_dense.txt
		for ac in 2 10 100 1000 10000; do timeout 600 $R/build/list_vs_dense 2000000 $ac 4096 3 >> $OUT/t2_list_vs_dense.txt 2>&1; done
		for ac in 2 10 100 1000; do timeout 600 $R/build/list_vs_dense 5008 $ac 16384 3 >> $OUT/t2_list_vs_dense.txt 2>&1; done
		cat $OUT/t2_list_vs_dense.txt
		;;
	small_n_pmc)   # where do short rows lose their time?  The count kernel alone (csrc/tools/count_microbench, list kernel in patch
		# order, rectangle of 16,384 x 16,384 plane rows = 16,384 tiles) at 5, 10 and 40 K-chunks per tile - the same word pairs
		# per tile-chunk, only the per-tile events (ticket, descriptor loads, epilogue, first LDS wait of a unit) differ in
		# frequency - under SQ counter passes of <= 8 counters each (those this chip's rocprofv3 lists).
		rocprofv3 -L > $OUT/rocprofv3_avail.txt 2>&1
		want="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INST_CYCLES_SALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_INSTS_WAVE32_LDS SQ_INSTS_GDS SQ_ACTIVE_INST_EXP_GDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_IFETCH_LEVEL SQ_CYCLES SQ_WAVES_EQ_64 SQ_ITEMS SQ_INSTS_EXP_GDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC"
		have=""; for c in $want; do grep -qw "$c" $OUT/rocprofv3_avail.txt && have="$have $c"; done
		echo "counters available: $have" | tee $OUT/small_n_pmc_counters.txt
		for words in 160 320 1280; do
			echo "== count_microbench 16384 x $words (${words}/32 chunks per tile), kernel alone"
			ONLYMODE=2 timeout 300 $R/build/count_microbench 16384 $words 3 | tee $OUT/small_n_w${words}_time.txt
			set -- $have; pass=0
			while [ $# -gt 0 ]; do
				grp=""; n=0; while [ $# -gt 0 ] && [ $n -lt 7 ]; do grp="$grp $1"; shift; n=$((n+1)); done
				rm -rf /tmp/pmc_sn
				ONLYMODE=2 timeout 600 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_sn -o sn -- $R/build/count_microbench 16384 $words 1 > /dev/null 2> $OUT/small_n_w${words}_p${pass}.log
				f=$(find /tmp/pmc_sn -name "*counter_collection.csv" | head -1)
				[ -n "$f" ] && python3 $R/profiles/sum_counters.py "$f" > $OUT/small_n_w${words}_p${pass}_pmc_sums.json || echo "pass $pass ($grp) gave no counters: $(tail -2 $OUT/small_n_w${words}_p${pass}.log)"
				pass=$((pass+1))
			done
		done
		python3 - <<PY
import glob, json, re
out = {}
for words in (160, 320, 1280):
    row = {}
    for f in sorted(glob.glob("$OUT/small_n_w%d_p*_pmc_sums.json" % words)):
        d = json.load(open(f))
        for k, v in d.items():
            if "k_count_list_t" in k:
                n = v.get("launches", 1)
                for c, x in v.items():
                    if c != "launches": row[c] = x / n
    t = open("$OUT/small_n_w%d_time.txt" % words).read()
    m = re.search(r"list/patch .* best ([0-9.]+) ms .*\(([0-9.]+)% of the and\+bcnt", t)
    if m: row["best_ms"] = float(m.group(1)); row["pct_of_and_bcnt_ceiling"] = float(m.group(2))
    out["%d chunks per tile" % (words // 32)] = row
json.dump(out, open("$OUT/small_n_pmc.json", "w"), indent=1, sort_keys=True)
ks = sorted({k for r in out.values() for k in r})
print("%-28s" % "per launch" + "".join("%18s" % k for k in out))
for k in ks: print("%-28s" % k + "".join("%18.6g" % out[c].get(k, float("nan")) for c in out))
PY
		;;
	clock_probe)   # the shader clock the count kernel's blocks really run at (s_memtime against the constant 100 MHz counter, over each
		# block's life; csrc/tools/count_microbench FINISH=1): one launch from idle, the 2nd and the 20th of launches back to back,
		# at 5 / 10 / 40 chunks per tile (16,384 tiles) and at N = 1 M rows (977 chunks, 1,024 tiles).  The and+bcnt ceiling is
		# quoted at 2.4 GHz; what a launch gets is in this file.
		: > $OUT/clock_probe.txt
		for reps in 1 2 20; do
			for shape in "16384 160" "16384 320" "16384 1280" "4096 31264"; do
				echo "-- FINISH_REPS=$reps count_microbench $shape" >> $OUT/clock_probe.txt
				FINISH=1 FINISH_REPS=$reps timeout 300 $R/build/count_microbench $shape 1 2>&1 | head -2 >> $OUT/clock_probe.txt
			done
		done
		cat $OUT/clock_probe.txt
		;;
	probe)   # K1's rare x common path, measured before built: a rare variant's carriers probing a common variant's bitvector row against the
		# dense contraction of the same pair (csrc/tools/probe_vs_dense.hip), 2N = 2,000,000 and 2N = 131,072
		: > $OUT/probe_vs_dense.txt
		for ac in 2 10 30 100 300 1000; do timeout 900 $R/build/probe_vs_dense 2000000 $ac 2048 2048 3 >> $OUT/probe_vs_dense.txt 2>&1; done
		for ac in 2 10 30 100; do timeout 900 $R/build/probe_vs_dense 131072 $ac 4096 4096 3 >> $OUT/probe_vs_dense.txt 2>&1; done
		cat $OUT/probe_vs_dense.txt
		;;
	base_sweep)   # does the count kernel's speed depend on where its rows lie in memory?  (a whole `calc` run is now and then 1.7-1.8x
		# slower than the same run a second earlier: r04_small_n_ab.txt, r04_band_trace.txt, r04_kg_prof.txt)
		: > $OUT/base_sweep.txt
		for shape in "16384 160" "16384 96" "8192 31264"; do
			BASE_SWEEP=64 timeout 600 $R/build/count_microbench $shape 2 >> $OUT/base_sweep.txt 2>&1
			BASE_SWEEP=64 STEP=4096 timeout 600 $R/build/count_microbench $shape 2 >> $OUT/base_sweep.txt 2>&1
		done
		cat $OUT/base_sweep.txt
		;;
	kg_prof)  # the small-N regime (the reference's published shape, 2,504 samples x 200,000 cohort-shaped variants): kernel traces of
		# `calc -p -w 1000000` (33 M surviving pairs) and of all-vs-all `-r 0.8` without the allele-count band, each with the
		# fused count -> r2 screen kernel (default) and without it (--engine-option fused=0)
		python3 - <<PY
import sys, os
sys.path.insert(0, "$R")
from tomahawk_amd import hostlib as H
if not os.path.exists("/tmp/kg_2504_200k.twk"):
    H.write_cohort_twk("/tmp/kg_2504_200k.twk", 2504, 200_000, seed=12, n_threads=64, block_size=500, spacing=100)
PY
		$R/tomahawk_amd/bin/tomahawk calc -i /tmp/kg_2504_200k.twk -o /tmp/o.two -t 64 -r 0.8 > /dev/null 2>&1     # warm the page cache
		for fused in 1 0; do
			for run in w1m all w1m_u all_u; do
				case $run in
				w1m)   args="-p -w 1000000"; export TWK_HIP_NO_SCREEN=0 ;;
				all)   args="-r 0.8"; export TWK_HIP_NO_SCREEN=1 ;;
				w1m_u) args="-u -w 1000000"; export TWK_HIP_NO_SCREEN=0 ;;
				all_u) args="-u -r 0.8"; export TWK_HIP_NO_SCREEN=1 ;;
				esac
				[ $TWK_HIP_NO_SCREEN = 0 ] && unset TWK_HIP_NO_SCREEN
				name=kg_${run}_fused$fused
				rm -rf /tmp/prof_$name
				timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o m -- \
					$R/tomahawk_amd/bin/tomahawk calc -i /tmp/kg_2504_200k.twk -o /tmp/o.two -t 64 --engine-option fused=$fused $args > /dev/null 2> $OUT/${name}_under_rocprof.log
				f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${name}_kernel_stats.csv
				echo "== $name"; grep "Finished\|HIP\]" $OUT/${name}_under_rocprof.log | cut -c1-330
				python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/${name}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:9]:
    print("   %-60s calls %5s  total %9.3f ms  avg %9.3f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
print("   all kernels: %.3f ms" % (tot / 1e6))
PY
			done
		done
		unset TWK_HIP_NO_SCREEN
		;;
	shards)  timeout 1200 python3 $R/tests/sweeps/shard_timings.py > $OUT/shard_timings.txt 2>&1; tail -20 $OUT/shard_timings.txt ;;
	*) echo "unknown stage $stage" ;;
	esac
done
