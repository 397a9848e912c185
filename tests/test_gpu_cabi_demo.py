"""The C-ABI is self-sufficient: examples/slhip_batch_demo.cpp (plain C++, no Python, no torch -- hipMalloc + the entry
points of include/slhip.h) stages, settles, frames and renders a batch from an asset blob, and produces bit for bit what the
Python host layer (sl.SceneBatch) produces for the same batch.  This is the call sequence the reference's host C++
(Scene::simulateTableTopScene + RenderPass::render, src/scene.cpp:612-759, src/render_pass.cpp:303-796) maps to."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from stillleben_amd import _abi
from stillleben_amd import _settle_batch as SB

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "stillleben_amd", "lib", "slhip_batch_demo")


def test_cpp_demo_matches_the_python_host_path(sl, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import export_assets

    import __graft_entry__ as g

    g.build_demo()
    assert os.path.exists(DEMO)
    n_scenes, n_obj, (W, H), seed, frames = 6, 4, (320, 240), 11, 40
    meshes = export_assets.demo_meshes(sl)
    blob, out = str(tmp_path / "assets.bin"), str(tmp_path / "out.bin")
    _, p, sp = export_assets.export(blob, meshes, n_scenes, n_obj, (W, H), seed, frames=frames)
    r = subprocess.run([DEMO, blob, out, str(W), str(H)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "0 scenes refused" in r.stdout
    raw = open(out, "rb").read()
    nb, P = n_scenes * n_obj, W * H
    off = 0
    bodies = np.frombuffer(raw, SB.BODY_DTYPE, nb, off); off += nb * SB.BODY_DTYPE.itemsize
    scenes = np.frombuffer(raw, _abi.SYNTH_SCENE_DTYPE, n_scenes, off); off += n_scenes * _abi.SYNTH_SCENE_DTYPE.itemsize
    inst = np.frombuffer(raw, np.uint16, n_scenes * P, off).reshape(n_scenes, H, W); off += 2 * n_scenes * P
    coord = np.frombuffer(raw, np.float32, n_scenes * P * 4, off).reshape(n_scenes, H, W, 4); off += 16 * n_scenes * P
    rgb = np.frombuffer(raw, np.uint8, n_scenes * P * 4, off).reshape(n_scenes, H, W, 4); off += 4 * n_scenes * P
    assert off == len(raw)
    # the same batch through the Python host layer
    table = sl.AssetTable(meshes)
    batch = sl.SceneBatch(table, n_scenes, n_obj, resolution=(W, H), seed=seed, render_chunk=n_scenes, manual_exposure=1.0)
    batch.set_camera_intrinsics(533.4, 533.7, W / 2 - 3.5, H / 2 + 0.6)
    assert np.array_equal(np.asarray(batch.params["proj"]), np.asarray(p["proj"]))
    batch.stage()
    batch.settle(frames=frames)
    batch.check_settled()
    batch.place()
    buf = batch.render(0, mask=_abi.OUT_GT6 | _abi.OUT_CAM_COORD)
    torch.cuda.synchronize()
    ref_b = batch.host_bodies()
    for f in ("pose", "lin_vel", "ang_vel", "separation", "flags", "wake_counter", "stuck_counter"):
        assert np.array_equal(np.ascontiguousarray(bodies[f]).view(np.uint8), np.ascontiguousarray(ref_b[f]).view(np.uint8)), f
    assert np.array_equal(scenes["camera_pose"], batch.host_scenes()["camera_pose"])
    assert np.array_equal(inst, buf.instance.cpu().numpy().view(np.uint16)[..., 0])
    assert np.array_equal(coord.view(np.uint32), buf.coord.cpu().numpy().view(np.uint32))
    assert np.array_equal(rgb, buf.rgb.cpu().numpy())
    assert (inst != 0).sum() > 2000 and len(np.unique(inst)) >= 4
