#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 --pmc counter_collection.csv -> JSON on stdout.
{kernel: {"launches": n, counter: sum, ...}}  (counter values summed over the launches of the kernel)."""
import csv
import json
import sys
from collections import defaultdict

out = defaultdict(lambda: defaultdict(float))
seen = defaultdict(set)
with open(sys.argv[1], newline="") as f:
    for row in csv.DictReader(f):
        k = row.get("Kernel_Name") or row.get("kernel_name")
        k = k.split("(")[0].replace("void ", "").strip()
        out[k][row["Counter_Name"]] += float(row["Counter_Value"])
        seen[k].add(row.get("Dispatch_Id") or row.get("dispatch_id"))
print(json.dumps({k: dict(launches=len(seen[k]), **v) for k, v in out.items()}, indent=1))
