import os, subprocess, sys, random, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_import as TI
ASAN = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tomahawk_amd", "lib_asan", "tomahawk")
env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="halt_on_error=1:exitcode=98:print_stacktrace=1")
tmp = pathlib.Path("/tmp/fuzz_bcf"); tmp.mkdir(exist_ok=True)
TI.test_import_bcf_equals_import_vcf(tmp, False)
TI.test_import_bcf_equals_import_vcf(tmp, True)
seeds = [open(tmp / "in.raw.bcf", "rb").read(), open(tmp / "in.bcf", "rb").read()]
rng = random.Random(11)
def mutate(data):
    b = bytearray(data)
    k = rng.randrange(4)
    if k == 0:
        for _ in range(rng.randrange(1, 8)): b[rng.randrange(len(b))] = rng.randrange(256)
    elif k == 1:
        i = rng.randrange(len(b)); b[i] ^= 1 << rng.randrange(8)
    elif k == 2:
        i = rng.randrange(len(b)); del b[i:i + rng.randrange(1, 64)]
    else:
        i = rng.randrange(len(b)); b[i:i] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 32)))
    return bytes(b)
bad = 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for it in range(n):
    for k, s in enumerate(seeds):
        p = str(tmp / f"m{k}.bcf"); open(p, "wb").write(mutate(s))
        try:
            r = subprocess.run([ASAN, "import", "-i", p, "-o", str(tmp / "o"), "-t", "2"], capture_output=True, env=env, timeout=60)
        except subprocess.TimeoutExpired:
            print("TIMEOUT", it, k); bad += 1; continue
        if r.returncode not in (0, 1):
            bad += 1; print("BAD rc", r.returncode, it, k); print(r.stderr.decode(errors="replace")[-1500:])
print("iterations", n, "bad", bad)
