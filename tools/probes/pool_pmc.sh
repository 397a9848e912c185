#!/bin/bash
# developer probe: SQ counters of the fixed-layout and the pooled solver
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS"; do
  echo "## $c"
  /root/repo/tools/kpmc.sh "$c" python /root/repo/tools/settle_throughput.py 4096 1 1 | grep "k_w_solve"
  SLHIP_LIB=/root/repo/build/libslhip_eu2.so SLHIP_SOLVE_POOL=8 SLHIP_SOLVE_POOL_KB=320 /root/repo/tools/kpmc.sh "$c" python /root/repo/tools/settle_throughput.py 4096 1 1 | grep "k_w_solve"
done
