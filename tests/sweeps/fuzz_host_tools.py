import os, subprocess, sys, random, numpy as np, gzip
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util
from tomahawk_amd import hostlib
ASAN = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tomahawk_amd", "lib_asan", "tomahawk")
env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="halt_on_error=1:exitcode=98:print_stacktrace=1")
tmp = "/tmp/fuzz"; os.makedirs(tmp, exist_ok=True)
# seeds
two = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "ref_n64_small_p.two")
al = util.random_alleles(60, 40, 3, miss_rate=0.1, miss_variants=0.3)
pos = (1000 + 10 * np.arange(60)).astype(np.uint32); rid = (np.arange(60) // 30).astype(np.uint32)
twk = f"{tmp}/seed.twk"; hostlib.write_twk(twk, al, pos, rid, phased=np.ones(60, np.uint8), n_contigs=2, block_size=16)
vcf = f"{tmp}/seed.vcf"
with open(vcf, "w") as f:
    f.write("##fileformat=VCFv4.2\n##contig=<ID=1,length=100000>\n##contig=<ID=2,length=100000>\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"GT\">\n")
    f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"S{i}" for i in range(40)) + "\n")
    for v in range(60):
        gts = ["%s|%s" % tuple("." if a == 2 else str(a) for a in al[v, s]) for s in range(40)]
        f.write(f"{rid[v]+1}\t{pos[v]}\t.\tA\tC\t.\tPASS\t.\tGT\t" + "\t".join(gts) + "\n")
rng = random.Random(7)
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
def run(cmd, tag):
    global bad
    try:
        r = subprocess.run(cmd, capture_output=True, env=env, timeout=60)
    except subprocess.TimeoutExpired:
        print("TIMEOUT", tag, cmd); bad += 1; return
    if r.returncode not in (0, 1):
        bad += 1; print("BAD rc", r.returncode, tag, " ".join(cmd)); print(r.stderr.decode(errors="replace")[-1500:])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed_two = open(two, "rb").read(); seed_twk = open(twk, "rb").read(); seed_vcf = open(vcf, "rb").read()
for it in range(n):
    p = f"{tmp}/m.two"; open(p, "wb").write(mutate(seed_two))
    run([ASAN, "view", "-i", p, "-H"], f"view {it}")
    run([ASAN, "sort", "-i", p, "-o", f"{tmp}/s.two", "-t", "2"], f"sort {it}")
    run([ASAN, "concat", "-i", p, "-i", two, "-o", f"{tmp}/c.two"], f"concat {it}")
    p = f"{tmp}/m.twk"; open(p, "wb").write(mutate(seed_twk))
    run([ASAN, "calc", "-i", p, "-o", f"{tmp}/x.two"], f"calc {it}")            # no GPU here: fails after the reader, exercising the header / index parsing
    p = f"{tmp}/m.vcf"; open(p, "wb").write(mutate(seed_vcf))
    run([ASAN, "import", "-i", p, "-o", f"{tmp}/i", "-t", "2"], f"import {it}")
    if it % 3 == 0:
        pz = f"{tmp}/m.vcf.gz"; open(pz, "wb").write(mutate(gzip.compress(seed_vcf)))
        run([ASAN, "import", "-i", pz, "-o", f"{tmp}/iz", "-t", "2"], f"import gz {it}")
print("iterations", n, "bad", bad)
