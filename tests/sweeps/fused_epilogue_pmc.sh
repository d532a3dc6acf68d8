cd /tmp; export TMPDIR=/tmp
for v in r05 r06; do
  bin=$GRAFT_REPO_ROOT/build/count_microbench; [ $v = r05 ] && bin=$GRAFT_REPO_ROOT/build/count_microbench_r05
  for shape in "32768 160 160" "32768 96 79"; do
    set -- $shape
    rm -rf /tmp/pmc_$v; mkdir -p /tmp/pmc_$v
    FUSED=1 LIVE=$3 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/pmc_$v -o pmc -- $bin $1 $2 1 512 > /dev/null 2>&1
    f=$(find /tmp/pmc_$v -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$v" "$1 $2 live $3" <<'PY'
import csv, sys, collections
f, v, shape = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f)):
    name = row["Kernel_Name"]
    if "k_count" not in name: continue
    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, c in sorted(acc.items()):
    short = name.split("(")[0].replace("twk::", "")
    # per launch (the tool launches each form several times), per lane-tile: 65536 tiles x 8 waves (counters count wave instructions)
    tiles = 65536 * 8
    print(v, shape, short, {k: round(sum(x) / len(x) / tiles, 1) for k, x in c.items()}, "launches", len(next(iter(c.values()))))
PY
  done
done
