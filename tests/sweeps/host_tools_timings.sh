#!/bin/bash
# Timings behind profiles/r01_two_tools.txt: dense calc output, sort, view, and a 1000-Genomes-shaped
# windowed run, each against the compiled reference (oracle/_ref/tomahawk_ref).  Run from the repo root
# on a GPU box:  bash tests/sweeps/host_tools_timings.sh
T=tomahawk_amd/bin/tomahawk; R=oracle/_ref/tomahawk_ref
python - <<'PY'
from tomahawk_amd import hostlib as H
H.write_synthetic_twk("/tmp/d.twk", 2504, 8000, seed=3, phased=True, block_size=500, n_threads=32)
H.write_synthetic_twk("/tmp/w.twk", 2504, 200000, seed=9, phased=True, block_size=500, n_threads=64)
PY
echo "== dense calc (-r 0)";     ( time timeout 300 $T calc -p -i /tmp/d.twk -o /tmp/d.two -r 0 -t 32 2> /tmp/my.log < /dev/null ) 2>&1 | grep real; grep -E "HIP" /tmp/my.log
echo "== sort";                  ( time timeout 300 $T sort -i /tmp/d.two -o /tmp/s_my.two -t 32 2> /dev/null < /dev/null ) 2>&1 | grep real
echo "== view to text";          ( time timeout 300 $T view -i /tmp/s_my.two -H -t 32 > /dev/null 2>&1 < /dev/null ) 2>&1 | grep real
echo "== view -I";               ( time timeout 300 $T view -i /tmp/s_my.two -H -I 1:100000-200000 < /dev/null | wc -l ) 2>&1 | grep -E "real|^[0-9]"
echo "== windowed 200k variants";( time timeout 300 $T calc -p -i /tmp/w.twk -o /tmp/w_my.two -w 1000000 2> /tmp/my.log < /dev/null ) 2>&1 | grep real; grep -E "Finished in" /tmp/my.log
if [ -x $R ] && [ "$1" = "--with-reference" ]; then
  echo "== reference: windowed"; ( time timeout 900 $R calc -p -i /tmp/w.twk -o /tmp/w_ref.two -w 1000000 2> /tmp/ref.log < /dev/null ) 2>&1 | grep real; grep "Finished in" /tmp/ref.log
  echo "== reference: sort";     ( time timeout 900 $R sort -i /tmp/d.two -o /tmp/s_ref.two > /dev/null 2>&1 < /dev/null ) 2>&1 | grep real
  echo "== reference: view";     ( time timeout 900 $R view -i /tmp/s_ref.two -H > /dev/null 2>&1 < /dev/null ) 2>&1 | grep real
  echo "== reference: dense calc (about 8 minutes)"; ( time timeout 1200 $R calc -p -i /tmp/d.twk -o /tmp/d_ref.two -r 0 > /dev/null 2>&1 < /dev/null ) 2>&1 | grep real
fi
