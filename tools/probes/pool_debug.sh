#!/bin/bash
# developer probe: pooled solver variants at the bench batch size
export SLHIP_LIB=/root/repo/build/libslhip_eu2.so
run() { /root/repo/tools/kstats.sh python /root/repo/tools/settle_throughput.py 16384 1 1 | grep "k_w_solve"; }
echo "spw4 pool8 160KB"; SLHIP_SOLVE_POOL_SPW=4 SLHIP_SOLVE_POOL=8 SLHIP_SOLVE_POOL_KB=320 run
echo "spw4 pool4 80KB"; SLHIP_SOLVE_POOL_SPW=4 SLHIP_SOLVE_POOL=4 SLHIP_SOLVE_POOL_KB=320 run
echo "spw2 pool8 160KB"; SLHIP_SOLVE_POOL=8 SLHIP_SOLVE_POOL_KB=320 run
SLHIP_SOLVE_POOL_SPW=4 SLHIP_SOLVE_POOL=8 SLHIP_SOLVE_POOL_KB=320 timeout 300 python -m pytest tests/test_gpu_settle.py -x -q -m gpu 2>&1 | tail -2
