#!/usr/bin/env python3
"""Known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this box
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half the bytes of a 16 B/lane streaming read).
Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`; tools/collect_counters.py reads the
counters of the `vectorized_elementwise_kernel` dispatches and compares them with N_BYTES."""
import sys

import torch

N_BYTES = 1 << 31            # 2 GiB read + 2 GiB written per dispatch: far beyond the 256 MiB Infinity Cache
a = torch.zeros(N_BYTES // 4, dtype=torch.float32, device="cuda")
b = torch.empty_like(a)
if len(sys.argv) > 1 and sys.argv[1] == "4":
    # 4 B per lane: a view whose storage offset breaks the 16-byte alignment makes torch take its scalar (dword) loop
    a2 = torch.zeros(N_BYTES // 4 + 1, dtype=torch.float32, device="cuda")[1:]
    a = a2
torch.cuda.synchronize()
for _ in range(4):
    torch.add(a, 1.0, out=b)     # one (vectorized) elementwise kernel: reads a (16 B per lane; 4 B with the argument "4"), writes b
torch.cuda.synchronize()
print("calibration: 4 dispatches of %d bytes read + %d bytes written" % (N_BYTES, N_BYTES))
