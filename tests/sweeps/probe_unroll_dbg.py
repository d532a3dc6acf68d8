"""The open question of ld_list.hip.h: the unphased probe kernel taking 2 or 4 list entries at a time (option probe_unroll)
against one at a time, on the data of test_probe_pass_equals_dense_and_merges_only[2-1500-True]: records against the dense
run, four repeats each; for the pairs that should not be there: the rows' and the column's list lengths and allele counts.
  python tests/sweeps/probe_unroll_dbg.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tomahawk_amd as T
from oracle import oracle as O
from tests import util
from tests.test_gpu_lists import _cohort_alleles, _with_flips, ORDER
mode, N, M = 2, 1500, 2200
al = _with_flips(_cohort_alleles(M, N, 1300 + N + mode), 7)
data, mask = O.bitvectors_from_alleles(al)
variants = O.variants_from_alleles(al)
hip = T.HipLd(0)


def run(lists, probe, unroll=1):
    hip.set_option("lists", lists); hip.set_option("probe", probe); hip.set_option("probe_unroll", unroll)
    hip.set_problem(N, M); hip.upload(data, util.to_hip_meta(variants), None)
    return np.sort(hip.ld_all(mode, T.Filters(minR2=0.1), window=T.OPT_R2_SCREEN)[0], order=ORDER)


d = run(0, 0)
kd = set(zip(d["idxA"].tolist(), d["idxB"].tolist()))
g = al.astype(np.int16)
het = (g[:, :, 0] != g[:, :, 1]).sum(1); homalt = ((g[:, :, 0] == 1) & (g[:, :, 1] == 1)).sum(1)
llen = het + np.minimum(homalt, N - het - homalt)
for unroll in (4, 2, 1, 4):
    for rep in range(4):
        p = run(2, 1, unroll)
        kp = list(zip(p["idxA"].tolist(), p["idxB"].tolist()))
        extra = [k for k in kp if k not in kd]
        cols = sorted(set(b for _, b in extra))
        print(f"probe_unroll={unroll} run {rep}: dense {len(d)} probes {len(p)} extra {len(extra)} missing {len(kd - set(kp))}"
              + (f"; columns {cols[:4]} list lengths {[int(llen[c]) for c in cols[:4]]} ac {[int(variants['ac'][c]) for c in cols[:4]]}; rows' list lengths {sorted(set(int(llen[a]) for a, _ in extra))[:6]}" if extra else ""), flush=True)
