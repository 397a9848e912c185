"""The C++ host layer of the per-object API (csrc/slhip_records.cpp through stillleben_amd/_host_records.py): one call per batch
assembles the records that `_batch.build_batch_per_scene` builds scene by scene in numpy -- the same bits -- and its shadow
matrices agree with the float32 numpy statement of render_pass.cpp:69-211 (`_shadow.shadow_matrices_numpy`)."""
import numpy as np
import torch

import scenes as S
from stillleben_amd import _batch, _host_records, _shadow
from stillleben_amd._batch import HostPool


def _scenes(sl):
    scs = [S.clutter_scene(sl, 60 + i, n_objects=2 + 2 * i, size=(160, 120), with_bunny=(i % 2 == 1)) for i in range(4)]
    scs[1].objects[0].metallic = 0.7
    scs[1].objects[1].roughness = 0.2
    scs[2].objects[0].casts_shadows = False
    ld = scs[3].light_directions.clone()
    ld[1] = torch.tensor([0.2, 0.4, -1.0])
    scs[3].light_directions = ld
    lc = scs[3].light_colors.clone()
    lc[1] = torch.tensor([50.0, 60.0, 70.0])
    scs[3].light_colors = lc
    scs[0].background_plane_size = torch.tensor([0.0, 0.0])          # a scene without the plane
    scs.append(sl.Scene((160, 120)))                                  # ... and one without objects
    return scs


def test_batch_records_equal_the_per_scene_path(sl):
    scs = _scenes(sl)
    assert _host_records.eligible(scs, None)
    p1, p2 = HostPool(), HostPool()
    s1, d1, c1 = _host_records.build(scs, p1, with_shadows=True)
    s2, d2, c2 = _batch.build_batch_per_scene(scs, p2, None, True)
    assert len(d1) == len(d2) and len(c1) == len(c2)
    for name in s2.dtype.names:
        assert np.array_equal(np.ascontiguousarray(s1[name]).view(np.uint8), np.ascontiguousarray(s2[name]).view(np.uint8)), name
    for name in d2.dtype.names:
        assert np.array_equal(np.ascontiguousarray(d1[name]).view(np.uint8), np.ascontiguousarray(d2[name]).view(np.uint8)), name
    assert np.array_equal(c1, c2)
    # what the dispatcher picks
    s3, d3, c3 = _batch.build_batch(scs, HostPool(), with_shadows=True)
    assert s3.tobytes() == s1.tobytes() and d3.tobytes() == d1.tobytes() and c3.tobytes() == c1.tobytes()


def test_shadow_matrices_agree_with_the_numpy_statement(sl):
    for sc in _scenes(sl):
        a = _shadow.shadow_matrices(sc)
        b = _shadow.shadow_matrices_numpy(sc)
        for x, y in zip(a, b):
            assert np.allclose(x, y, rtol=2e-4, atol=2e-4)


def test_ineligible_batches_take_the_per_scene_path(sl):
    scs = _scenes(sl)[:2]
    assert not _host_records.eligible(scs, lambda o: True)
    s1, d1, c1 = _batch.build_batch(scs, HostPool(), predicate=lambda o: o.instance_index != 1)
    s2, d2, c2 = _batch.build_batch(scs, HostPool())
    assert len(d1) == len(d2) - 2                                     # instance 1 of both scenes left out
