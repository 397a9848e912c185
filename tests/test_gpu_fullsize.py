"""BASELINE config C2 at the benchmark's full step size -- 32768 scenes of 20 YCB-like objects in one batch -- checked
through properties that do not need the CPU oracle to run 32768 settles (it would take minutes):

  * placement independence: a scene's random stream is keyed by its scene id, so scenes 7000..7063 of the big batch must
    come out bit for bit like a separate 64-scene batch staged at scene_id_base = 7000 -- other launch sizes, other
    neighbours AND the other launch form: the big batch runs the lockstep kernels (cross-scene work lists, cost-ordered solver
    launch, partners in the solver waves), the small one the persistent kernel;
  * sanity of every body of every scene: finite, above the table, rotation orthonormal;
  * the oracle on a few of the scenes, bit for bit;
  * the rendered ground truth of a scene is the same bits whether it is rendered as slot 7000 - 6144 of a 1024-scene chunk
    or as slot 0 of the small batch."""
import numpy as np
import pytest
import torch

from stillleben_amd import _abi
from stillleben_amd import _settle_batch as SB

pytestmark = pytest.mark.gpu

N_BIG, N_OBJ, BASE_SMALL, N_SMALL = 32768, 20, 7000, 64


@pytest.fixture(scope="module")
def table(sl):
    from stillleben_amd import synthetic

    return sl.AssetTable(synthetic.ycb_like_meshes(seed=0, tex_size=256))


def _fields_equal(a, b, what):
    for f in ("pose", "lin_vel", "ang_vel", "separation", "flags", "stuck_counter", "wake_counter"):
        x, y = np.ascontiguousarray(a[f]), np.ascontiguousarray(b[f])
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), "%s: %s differs" % (what, f)


def test_full_step_is_placement_independent_and_sane(sl, oracle, table):
    big = sl.SceneBatch(table, N_BIG, N_OBJ, seed=2026, render_chunk=1024)
    big.set_camera_intrinsics(1066.778, 1067.487, 312.9869, 241.3109)
    big.stage()
    torch.cuda.synchronize()
    staged = big.host_bodies()[BASE_SMALL * N_OBJ:(BASE_SMALL + 4) * N_OBJ].copy()
    big.settle()
    big.check_settled()
    big.place()
    small = sl.SceneBatch(table, N_SMALL, N_OBJ, seed=2026, render_chunk=N_SMALL, scene_id_base=BASE_SMALL)
    small.set_camera_intrinsics(1066.778, 1067.487, 312.9869, 241.3109)
    small.stage()
    small.settle()
    small.check_settled()
    small.place()
    torch.cuda.synchronize()
    bb, bs = big.host_bodies(), small.host_bodies()
    _fields_equal(bb[BASE_SMALL * N_OBJ:(BASE_SMALL + N_SMALL) * N_OBJ], bs, "scenes %d.. of the big batch vs the small batch" % BASE_SMALL)
    # every body of every scene
    pose = bb["pose"].reshape(-1, 4, 4)
    assert np.isfinite(pose).all() and np.isfinite(bb["lin_vel"]).all() and np.isfinite(bb["ang_vel"]).all()
    assert (pose[:, 2, 3] > 0.0).all()                                       # nothing below the table
    R = pose[:, :3, :3].astype(np.float64)
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-4
    speed = np.linalg.norm(bb["lin_vel"][:, :3], axis=1)
    assert np.mean(speed < 0.05) >= 0.95                                     # SURVEY 8c k6 on the full batch: the piles are at rest
    assert np.mean((bb["flags"] & SB.BODY_ASLEEP) != 0) >= 0.85
    # the oracle on four of the scenes (full 400 steps)
    hulls, verts = big.se.pool.arrays()
    prm = np.array(big.settle_params)
    for k in range(4):
        ref = staged[k * N_OBJ:(k + 1) * N_OBJ].copy()
        srec1 = np.zeros(1, dtype=SB.SETTLE_SCENE_DTYPE)
        srec1["body_end"], srec1["has_plane"], srec1["plane_z"] = N_OBJ, 1, 0.04
        oracle.settle(srec1, ref, hulls, verts, prm)
        s = BASE_SMALL + k
        _fields_equal(bb[s * N_OBJ:(s + 1) * N_OBJ], ref, "scene %d vs the oracle" % s)
    # the ground truth of scene 7000: chunk 6 (scenes 6144..7167) of the big batch, slot 856, against slot 0 of the small batch
    chunk, slot = BASE_SMALL // 1024, BASE_SMALL % 1024
    gb = big.render(chunk)
    gs = small.render(0)
    torch.cuda.synchronize()
    for name in ("instance", "cls", "coord", "normals", "rgb"):
        a, b = getattr(gb, name)[slot], getattr(gs, name)[0]
        assert torch.equal(a, b), name
    assert int((gs.instance[0] != 0).sum()) > 2000                           # and there is something to see
