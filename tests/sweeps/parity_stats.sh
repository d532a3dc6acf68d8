#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r03
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
json.dump(agg, open("gpurun_out/r03/parity_stats_max.json","w"), indent=1)
PY
