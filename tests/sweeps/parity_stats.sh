#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TAG=${1:-r04}
mkdir -p gpurun_out/$TAG
rm -f /tmp/parity_stats.jsonl
export TWK_PARITY_STATS=/tmp/parity_stats.jsonl
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for s in haplotype_block_sweep hostile_genotype_sweep haplotype_block_sweep_large_n records_at_1m_samples; do timeout 900 python3 tests/sweeps/$s.py 2>&1 | tail -1; done
python3 - <<PY
import json
rows=[json.loads(l) for l in open("/tmp/parity_stats.jsonl")]
keys=[k for k in rows[0] if k not in ("test","n_samples","records","ties","n")]
agg={}
for r in rows:
    for k,v in r.items():
        if isinstance(v,(int,float)) and k not in ("n_samples","records","ties","n"): agg[k]=max(agg.get(k,0.0),v)
print("calls", len(rows), "cubic records", sum(r.get("n",0) for r in rows))
for k in sorted(agg): print(f"  {k}: {agg[k]:.3g}")
json.dump(agg, open("gpurun_out/$TAG/parity_stats_max.json","w"), indent=1)
# the cubic-path records that differ beyond the relative bar, each with what its own floor allowed (tests/util.py cubic_floors)
beyond = [dict(it, test=r.get("test", "")[:80], n_samples=r.get("n_samples")) for r in rows for it in r.get("beyond", [])]
beyond.sort(key=lambda i: -i["dD"])
json.dump({"records_beyond_rtol": len(beyond), "largest_dD": beyond[:20],
           "dD_decades": {str(d): sum(1 for i in beyond if i["dD"] > 0 and int(__import__("math").floor(__import__("math").log10(i["dD"]))) == d) for d in range(-20, -8)}},
          open("gpurun_out/$TAG/parity_stats_beyond.json", "w"), indent=1)
print("records beyond the relative bar:", len(beyond), "largest dD", beyond[0]["dD"] if beyond else None)
PY
