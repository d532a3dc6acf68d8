#!/bin/bash
# How far up in sample count the fused count -> r2 screen kernel pays (FUSED_MAX_CHUNKS, twk_hip.hip): `tomahawk calc` on
# N x 40,000 cohort-shaped variants, the fused form off (--engine-option fused=0), by the default policy (1) and forced (2), with and
# without the allele-count band (TWK_HIP_NO_SCREEN).  Prints the engine's own kernel times.  Run from the repo root on a GPU box:
#   tests/sweeps/fused_mid_n.sh > gpurun_out/r03/fused_mid_n.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
for N in 12000 20000 32000 60000 100000 140000; do
python3 - <<PY
import sys, os
sys.path.insert(0, "$R")
from tomahawk_amd import hostlib as H
p = "/tmp/mid_${N}.twk"
if not os.path.exists(p):
    H.write_cohort_twk(p, $N, 40000, seed=5, n_threads=64, block_size=500, spacing=100)
PY
  for args in "-p" "-u" "-p -w 400000"; do
    for ns in 0 1; do
      for fused in 0 1 2; do
        echo -n "N=$N calc $args  fused=$fused TWK_HIP_NO_SCREEN=$ns: "
        TWK_HIP_NO_SCREEN=$ns $R/tomahawk_amd/bin/tomahawk calc -i /tmp/mid_${N}.twk -o /tmp/o.two -t 64 --engine-option fused=$fused $args 2>&1 | grep -E "HIP\] count|Finished" | sed -e 's/.*Finished in \([0-9.]*s\).*output: \([0-9,]*\).*/wall \1 records \2;/' -e 's/.*count kernel \([0-9.]*\) ms in \([0-9]*\) launches.*math kernels \([0-9.]*\) ms\(.*\)/ count \1 ms (\2 launches) math \3 ms\4/' | cut -c1-200 | tr '\n' ' '
        echo
      done
    done
  done
  rm -f /tmp/mid_${N}.twk
done
