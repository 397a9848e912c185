#!/bin/bash
# Developer probe (GPU box): the solver's contacts in LDS (round 3) against contacts read from global memory with the next one
# prefetched (LDS per scene: bodies + lists only).
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/lds_probe.txt; : > $out
for v in 0 1; do
  SLHIP_EXTRA_FLAGS="-DSLHIP_SOLVE_CONTACTS_IN_LDS=$v" python -c "import __graft_entry__ as g; import os; os.remove(g.LIB); g.build()" >/dev/null 2>&1
  for spw in 1 2; do
    echo "contacts_in_lds=$v spw=$spw" >> $out
    SLHIP_SOLVE_SPW=$spw timeout 300 python tools/settle_throughput.py 16384 1 3 2>&1 | grep -v amdgpu.ids >> $out
  done
  if [ $v = 0 ]; then timeout 900 python -m pytest tests/test_gpu_settle.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3 >> $out; fi
done
cat $out
