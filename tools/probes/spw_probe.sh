#!/bin/bash
# Developer probe (GPU box): what several scenes per solver wave would buy if the solver's LDS were sized by need.
# The contact cap is lowered at build time (breaks parity: timing proxy only) so that 2 / 4 scenes fit a wave.
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/spw_probe.txt; : > $out
for cap in 255 96 64; do
  SLHIP_EXTRA_FLAGS="-DSLHIP_MAX_ACTIVE_CONTACTS=$cap" python -c "import __graft_entry__ as g; import os; os.remove(g.LIB); g.build()" >/dev/null 2>&1
  for spw in 1 2 4; do
    echo "cap=$cap spw=$spw" >> $out
    SLHIP_SOLVE_SPW=$spw timeout 300 python tools/settle_throughput.py 16384 1 3 >> $out 2>&1
  done
done
cat $out
