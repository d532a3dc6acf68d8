"""Which setting keeps RCCL from printing its start-up banner to stdout?  One native self-loop gather (twk_hip_gather_records) per setting in a
child process, stdout lines counted: the environment the child starts with, a putenv from inside the child before the library is opened, and
the engine's own setenv (twk_hip.hip Rccl::open)."""
import os, subprocess, sys
code = r'''
import os, sys
sys.path.insert(0, ".")
if os.environ.get("SET_INSIDE"):
    os.environ["NCCL_DEBUG"] = "NONE"
import numpy as np, tomahawk_amd as T
e = T.HipLd(0)
e.set_problem(500, 300); e.generate_synthetic(1)
e.set_device_sink(True)
e.ld_all(T.MODE_PHASED, T.Filters(minR2=0.0))
print("GATHER", T.gather_records([e], self_loop=True)[0], "NCCL_DEBUG now:", os.environ.get("NCCL_DEBUG"), file=sys.stderr)
import ctypes
libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p
print("C getenv:", libc.getenv(b"NCCL_DEBUG"), file=sys.stderr)
'''
for env in ({}, {"SET_INSIDE": "1"}, {"NCCL_DEBUG": "NONE"}, {"RCCL_LOG_LEVEL": "0"}):
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env))
    print(env, "-> stdout lines:", len(r.stdout.splitlines()), repr(r.stdout[:60]), "| stderr tail:", " / ".join(x[:70] for x in r.stderr.strip().splitlines()[-2:]))
