#!/usr/bin/env python3
"""Developer tool: wall time of a LightMap build (IBL precompute) at the reference's texture sizes."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import stillleben_amd as sl
from test_gpu_ibl import sky
sl.init_cuda(0)
eq = sky(1024, 2048)
sl.LightMap(eq, sizes=dict(env_size=64, env_levels=7, irr_size=8, pre_size=32, pre_levels=5, lut_size=32))
torch.cuda.synchronize()
t = time.perf_counter()
lm = sl.LightMap(eq)
torch.cuda.synchronize()
print("LightMap build at reference sizes (2048x1024 equirect): %.1f ms" % ((time.perf_counter() - t) * 1e3))
print("irradiance mean", float(lm.irradiance.view(-1, 4)[:, :3].mean()), "lut[0]", lm.brdf_lut[:2].tolist())
