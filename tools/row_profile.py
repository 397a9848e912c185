#!/usr/bin/env python3
"""Developer tool: where a contact row of k_w_solve spends its cycles (round-4/5 verdicts: "nobody has profiled a row yet").
Needs the row-profile build of the library (s_memtime stamps inside solve_group_lds, -DSLHIP_ROW_PROFILE):

    python tools/row_profile.py --build                 # here (cross-compile): stillleben_amd/lib/libslhip_rowprof.so
    SLHIP_LIB=stillleben_amd/lib/libslhip_rowprof.so python tools/row_profile.py [B=4096]     # on the GPU box

One settle of B C2 scenes; prints per row / per patch / per group visit the shader-clock cycles of its parts (a stamp costs an
s_memtime + s_waitcnt: upper bounds of the unstamped code)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF_LIB = os.path.join(ROOT, "stillleben_amd", "lib", "libslhip_rowprof.so")

if "--build" in sys.argv:
    import __graft_entry__ as g

    cmd = [g.HIPCC] + g.HIP_FLAGS + ["-DSLHIP_ROW_PROFILE"] + g._sources() + ["-o", PROF_LIB]
    subprocess.run(cmd, check=True)
    print(PROF_LIB)
    sys.exit(0)

import torch  # noqa: E402

import stillleben_amd as sl  # noqa: E402
from stillleben_amd import _abi, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sl.init_cuda(0)
L = _abi.lib()
if not hasattr(L, "slhip_settle_row_profile"):
    sys.exit("row_profile.py: the loaded library is not the row-profile build (SLHIP_LIB=.../libslhip_rowprof.so)")
table = sl.AssetTable(synthetic.ycb_like_meshes(seed=0, tex_size=64))
batch = sl.SceneBatch(table, B, 20, seed=20260929)
os.environ["SLHIP_SETTLE_PERSISTENT"] = "0"
batch.stage()
batch.settle(frames=2)
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
L.slhip_settle_row_profile(out)        # clear
batch.stage(scene_id_base=B)
batch.settle()
torch.cuda.synchronize()
L.slhip_settle_row_profile(out)
rows, load, alu, patches, fric, visits, other = [float(out[i]) for i in range(7)]
print("B=%d: %d normal rows, %d patches, %d group visits (lane pair 0 of every solver wave, LDS-resident groups)" % (B, rows, patches, visits))
print("  per normal row : %.0f cycles until the contact is in registers (72 B from LDS) + %.0f cycles of arithmetic = %.0f" %
      (load / rows, alu / rows, (load + alu) / rows))
print("  per patch      : %.0f cycles of friction rows (%.2f rows per patch)" % (fric / patches, rows / patches))
print("  per group visit: %.0f cycles outside the rows (body registers in / out), %.1f rows per visit" % (other / visits, rows / visits))
print("  a contact row with its share of the friction: %.0f cycles = %.2f us at 2.4 GHz" %
      ((load + alu + fric) / rows, (load + alu + fric) / rows / 2400.0))
