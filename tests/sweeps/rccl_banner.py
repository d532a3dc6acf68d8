import os, subprocess, sys
code = r'''
import sys
sys.path.insert(0, ".")
import numpy as np, tomahawk_amd as T
from tests import util
e = T.HipLd(0)
e.set_problem(500, 300); e.generate_synthetic(1)
e.set_device_sink(True)
e.ld_all(T.MODE_PHASED, T.Filters(minR2=0.0))
print("GATHER", T.gather_records([e], self_loop=True)[0], file=sys.stderr)
'''
for env in ({}, {"RCCL_LOG_LEVEL": "0"}, {"NCCL_DEBUG": "NONE"}, {"NCCL_DEBUG": "WARN"}, {"RCCL_LOG_LEVEL": "1"}):
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env))
    print(env, "-> stdout lines:", len(r.stdout.splitlines()), repr(r.stdout[:80]), "| stderr tail:", r.stderr.strip().splitlines()[-1][:60])
