"""CPU tests pinning the render ORACLE against the known-answer properties the reference's
own tests hold for this path (reference tests/basic.cpp:108-261, :375-453)."""
import math

import numpy as np
import torch

import scenes as S
from stillleben_amd import _abi
from stillleben_amd._batch import HostPool, build_batch


def oracle_render(oracle, scene_list, flags=_abi.OUT_ALL, **kw):
    pool = HostPool()
    srec, drec, crec = build_batch(scene_list, pool, with_shadows=bool(flags & _abi.RENDER_SHADOWS))
    W, H = scene_list[0].viewport
    return oracle.render(pool.arrays(), srec, drec, W, H, flags, **kw)


def test_cube_vertex_indices_and_barycentrics(sl, oracle):
    # g1 (basic.cpp:395-452): exactly 5 distinct vertex-id values incl. 0; ids pairwise distinct;
    # barycentrics sum to 1
    scene = S.cube_lookat_scene(sl)
    r = oracle_render(oracle, [scene])
    vi = r.vertex_idx[0, :, :, :3]
    n_points = 24
    assert vi.max() <= n_points
    assert len(np.unique(vi)) == 5
    covered = vi[..., 0] != 0
    assert covered.sum() > 1000
    v = vi[covered]
    assert np.all(v[:, 0] != v[:, 1]) and np.all(v[:, 1] != v[:, 2]) and np.all(v[:, 0] != v[:, 2])
    b = r.bary[0, :, :, :3][covered]
    assert np.allclose(b.sum(axis=1), 1.0, atol=1e-5)
    assert np.all(r.bary[0][~covered] == 0)


def test_cube_analytic_depth_and_silhouette(sl, oracle):
    # g2: the visible face x=+1 is fronto-parallel at camera z = 3 => depth 3.0 on every covered
    # pixel; the silhouette is the axis-aligned square of half-width fx/3 centred on (cx,cy)
    scene = S.cube_lookat_scene(sl)
    r = oracle_render(oracle, [scene])
    depth = r.coord[0, :, :, 3]
    inst = r.instance[0, :, :, 0]
    covered = inst != 0
    assert np.allclose(depth[covered], 3.0, atol=2e-6)
    fx = 640.0 / (2.0 * math.tan(math.radians(58.0) / 2.0))
    half = fx / 3.0
    ys, xs = np.nonzero(covered)
    # pixel i is covered iff its centre i+0.5 lies inside (cx - half, cx + half)
    exp_x0 = math.ceil(320.0 - half - 0.5)
    exp_x1 = math.floor(320.0 + half - 0.5)
    assert xs.min() in (exp_x0, exp_x0 + 1) and xs.max() in (exp_x1 - 1, exp_x1)
    assert (xs.max() - xs.min()) == (ys.max() - ys.min())
    assert covered[ys.min():ys.max() + 1, xs.min():xs.max() + 1].all()
    # g5: background values (render_pass.cpp:316,525-532)
    assert np.all(r.coord[0][~covered] == 3000.0)
    assert np.all(r.cam_coord[0][~covered] == 3000.0)
    assert np.all(r.normals[0][~covered] == 0.0)
    assert np.all(r.cls[0][~covered] == 0)
    assert np.all(r.vertex_idx[0][~covered] == 0)


def test_projection_identity(sl, oracle):
    # g6: u + 0.5 = fx * x / z + cx for the interpolated camera coordinates of every pixel
    scene = S.clutter_scene(sl, 3, n_objects=5, plane=False)
    fx, fy, cx, cy = 300.0, 310.0, 150.0, 125.0
    scene.set_camera_intrinsics(fx, fy, cx, cy)
    r = oracle_render(oracle, [scene], flags=_abi.OUT_ALL & ~_abi.OUT_RGB)
    cam = r.cam_coord[0]
    covered = r.instance[0, :, :, 0] != 0
    assert covered.sum() > 500
    ys, xs = np.nonzero(covered)
    c = cam[covered]
    u = fx * c[:, 0] / c[:, 2] + cx
    v = fy * c[:, 1] / c[:, 2] + cy
    assert np.abs(u - (xs + 0.5)).max() < 2e-2
    assert np.abs(v - (ys + 0.5)).max() < 2e-2
    # depth channel == camera z
    assert np.array_equal(r.coord[0][covered][:, 3], c[:, 2])


def test_bunny_render_invariants(sl, oracle):
    # g3/g4 (basic.cpp:108-261): instance index 0xFFFF reads back as int16 -1; class map >10 px and
    # < 50 %; vertexIndex pixel 0 == 0 and max > 10
    m = sl.Mesh(S.BUNNY, physics=False)
    m.center_bbox()
    m.scale_to_bbox_diagonal(0.5)
    scene = sl.Scene((640, 480))
    obj = sl.Object(m)
    scene.add_object(obj)
    assert obj.instance_index == 1
    obj.instance_index = 0xFFFF
    pose = torch.eye(4)
    pose[2, 3] = scene.min_dist_for_object_diameter(0.5)
    obj.set_pose(pose)
    r = oracle_render(oracle, [scene], flags=_abi.OUT_ALL & ~_abi.OUT_RGB)
    inst = r.instance[0, :, :, 0]
    n = inst.size
    cnt = int((inst == 65535).sum())
    assert 10 < cnt < n // 2
    assert set(np.unique(inst).tolist()) == {0, 65535}
    assert int(inst.view(np.int16).min()) == -1
    cls = r.cls[0, :, :, 0]
    assert 10 < int((cls != 0).sum()) < n // 2
    assert tuple(r.vertex_idx[0, 0, 0, :3]) == (0, 0, 0)
    assert r.vertex_idx.max() > 10
    # projected silhouette fills most of the image height at min distance
    ys, xs = np.nonzero(inst)
    assert ys.max() - ys.min() > 200


def test_rgb_alpha_and_tonemap(sl, oracle):
    # basic.cpp:173-187: >10 non-transparent pixels; background alpha 0
    scene = S.clutter_scene(sl, 1, n_objects=4, plane=False)
    scene.manual_exposure = 1.0
    r = oracle_render(oracle, [scene], flags=_abi.OUT_ALL | _abi.RENDER_SHADOWS | _abi.RENDER_SSAO, shadow_res=512)
    a = r.rgb[0, :, :, 3]
    covered = r.instance[0, :, :, 0] != 0
    assert np.all(a[covered] == 255) and np.all(a[~covered] == 0)
    assert np.all(r.rgb[0][~covered] == 0)
    assert r.rgb[0][covered][:, :3].max() > 20  # lit


def test_draw_order_tie_break_and_plane(sl, oracle):
    # two identical coincident cubes: the earlier object must win every pixel (GL_LESS)
    scene = S.cube_lookat_scene(sl, (320, 240))
    m = scene.objects[0].mesh
    o2 = sl.Object(m)
    scene.add_object(o2)
    r = oracle_render(oracle, [scene], flags=_abi.OUT_INSTANCE)
    assert set(np.unique(r.instance).tolist()) == {0, 1}


def test_depth_peel(sl, oracle):
    scene = S.clutter_scene(sl, 5, n_objects=6, plane=False)
    flags = _abi.OUT_COORD | _abi.OUT_INSTANCE
    r0 = oracle_render(oracle, [scene], flags=flags)
    r1 = oracle_render(oracle, [scene], flags=flags, depth_peel=r0.coord)
    d0, d1 = r0.coord[0, :, :, 3], r1.coord[0, :, :, 3]
    both = (r0.instance[0, :, :, 0] != 0) & (r1.instance[0, :, :, 0] != 0)
    assert both.sum() > 100
    assert np.all(d1[both] > d0[both])
    # nothing appears in the second layer where the first was empty
    assert not np.any((r0.instance[0] == 0) & (r1.instance[0] != 0))


def test_ssao_tables_match_generator(oracle):
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location(
        "gen", os.path.join(os.path.dirname(__file__), "..", "tools", "gen_ssao_tables.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    noise, kern = gen.tables()
    n2, k2 = oracle.ssao_tables()
    assert np.array_equal(noise.reshape(-1), n2)
    assert np.array_equal(kern.reshape(-1), k2)
    # kernel samples live in the +z hemisphere with length <= 1
    k = k2.reshape(64, 3)
    assert np.all(k[:, 2] >= 0) and np.all(np.linalg.norm(k, axis=1) <= 1.0)


def test_unorm8_shortcut_is_exact():
    """The kernels convert texture bytes with q = x * (1/255); q' = fma(fma(-q, 255, x), 1/255, q) instead of
    the IEEE division the oracle writes ((float)x / 255.0f, render_ref.c / diff_ref.c).  Checked here for all
    256 bytes with exactly rounded fma emulation (rational arithmetic, round to nearest even)."""
    from fractions import Fraction

    def rn32(fr):
        c = np.float32(float(fr))
        best = None
        for cand in (np.nextafter(c, np.float32(-np.inf)), c, np.nextafter(c, np.float32(np.inf))):
            err = abs(Fraction(float(cand)) - fr)
            even = (int(np.float32(cand).view(np.uint32)) & 1) == 0
            if best is None or err < best[0] or (err == best[0] and even and not best[2]):
                best = (err, cand, even)
        return np.float32(best[1])

    def fma(a, b, c):
        return rn32(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))

    c = np.float32(1.0) / np.float32(255.0)
    for x in range(256):
        xf = np.float32(x)
        q = xf * c
        fast = fma(fma(-q, np.float32(255.0), xf), c, q)
        assert fast == xf / np.float32(255.0), x


def _owners(r):
    """Per pixel the (sorted) vertex-id triple of the owning triangle; 0,0,0 = nobody."""
    return np.sort(r.vertex_idx[0, :, :, :3].astype(np.int64), axis=-1)


def _interior(covered):
    """Pixels whose 3x3 neighbourhood is inside the silhouette's filled outline: the sheet is convex, so every pixel
    between the first and last covered pixel of its row AND of its column belongs to it."""
    H, W = covered.shape
    rows = np.zeros_like(covered)
    for y in range(H):
        xs = np.nonzero(covered[y])[0]
        if len(xs):
            rows[y, xs[0]:xs[-1] + 1] = True
    cols = np.zeros_like(covered)
    for x in range(W):
        ys = np.nonzero(covered[:, x])[0]
        if len(ys):
            cols[ys[0]:ys[-1] + 1, x] = True
    return rows & cols


def check_watertight(render, sl, n_points, seed, tilt):
    """VERDICT r01 item 8 / SURVEY H1: along shared edges no pixel is owned twice and none never.  `render(scene)` returns
    the result object of one scene.  Never: the convex sheet's silhouette has no hole.  Twice (fronto-parallel sheet, every
    triangle at the same depth): submitting the triangles in reverse order leaves every pixel with the same owner."""
    fwd = render(S.sheet_scene(sl, S.delaunay_sheet(sl, n_points, seed), tilt_deg=tilt))
    inst = fwd.instance[0, :, :, 0]
    covered = inst != 0
    assert covered.sum() > 10000 and not covered[0].any() and not covered[:, 0].any()       # the whole sheet is in view
    holes = _interior(covered) & ~covered
    assert not holes.any(), "pixels inside the sheet owned by no triangle: %s" % np.argwhere(holes)[:5]
    own = _owners(fwd)
    assert (own[covered][:, 0] > 0).all() and (own[covered][:, 0] != own[covered][:, 1]).all()
    if tilt == 0.0:
        depth = fwd.coord[0, :, :, 3][covered]
        assert depth.min() == depth.max()                                                   # one depth: ties everywhere
        rev = render(S.sheet_scene(sl, S.delaunay_sheet(sl, n_points, seed, reverse=True), tilt_deg=tilt))
        assert np.array_equal(rev.instance[0, :, :, 0], inst)
        moved = np.any(_owners(rev) != own, axis=-1)
        assert not moved.any(), "%d pixels claimed by two triangles, first %s" % (moved.sum(), np.argwhere(moved)[:5])
    return int(covered.sum())


def test_watertight_sheet_no_pixel_owned_twice_or_never(sl, oracle):
    # triangles of ~50 px, of ~2 px and (mostly) smaller than a pixel; fronto-parallel and tilted
    for n_points, seed, tilt in ((600, 1, 0.0), (15000, 2, 0.0), (90000, 3, 0.0), (4000, 4, 35.0), (60000, 5, 50.0)):
        check_watertight(lambda scene: oracle_render(oracle, [scene], flags=_abi.OUT_INSTANCE | _abi.OUT_VERTEX_IDX | _abi.OUT_COORD),
                         sl, n_points, seed, tilt)


def test_gl_comparison_harness_separates_silhouette_from_interior(sl, oracle, tmp_path):
    """tools/compare_gl.py: a frame against itself is clean; against a one-pixel-shifted silhouette only
    silhouette pixels differ; a moved object or a depth offset is an interior disagreement."""
    import importlib.util
    import os
    import subprocess
    import sys

    spec = importlib.util.spec_from_file_location("compare_gl", os.path.join(os.path.dirname(__file__), "..", "tools", "compare_gl.py"))
    cg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cg)
    r = oracle_render(oracle, [S.clutter_scene(sl, 1, n_objects=6)], flags=_abi.OUT_ALL | _abi.RENDER_SSAO)
    ours = {"instance": r.instance[0], "cls": r.cls[0], "coord": r.coord[0], "normals": r.normals[0], "rgb": r.rgb[0]}
    rep = cg.compare(ours, ours)
    assert rep["ok"] and rep["silhouette_mismatch"] == 0 and rep["interior_mismatch"] == 0 and rep["max_depth_diff"] == 0.0
    # a different fill rule: every silhouette grown by one pixel to the right
    grown = dict(ours)
    inst = ours["instance"].copy()
    inst[:, 1:][(inst[:, 1:] == 0) & (inst[:, :-1] != 0)] = inst[:, :-1][(inst[:, 1:] == 0) & (inst[:, :-1] != 0)]
    grown["instance"] = inst
    rep = cg.compare(grown, ours)
    assert rep["ok"] and rep["silhouette_mismatch"] > 50 and rep["interior_mismatch"] == 0
    # a real disagreement: the frame shifted by 6 pixels, and a depth offset of 5 mm
    moved = {k: np.roll(v, 6, axis=1) for k, v in ours.items()}
    rep = cg.compare(moved, ours)
    assert not rep["ok"] and rep["interior_mismatch"] > 100
    off = dict(ours)
    off["coord"] = ours["coord"] + np.array([0, 0, 0, 0.005], np.float32)
    rep = cg.compare(off, ours)
    assert not rep["ok"] and rep["interior_mismatch"] == 0 and abs(rep["max_depth_diff"] - 0.005) < 1e-6
    # command line
    np.savez(tmp_path / "a.npz", **ours)
    np.savez(tmp_path / "b.npz", **moved)
    tool = os.path.join(os.path.dirname(__file__), "..", "tools", "compare_gl.py")
    assert subprocess.run([sys.executable, tool, str(tmp_path / "a.npz"), str(tmp_path / "a.npz")], capture_output=True).returncode == 0
    assert subprocess.run([sys.executable, tool, str(tmp_path / "a.npz"), str(tmp_path / "b.npz")], capture_output=True).returncode == 1


def lambert_kat_scene(sl, size=(640, 480)):
    """The cube of reference tests/basic.cpp:375-453 seen from (4, 0, 0): its face x = +1 is fronto-parallel at camera z = 3.
    ONE directional light shining straight along the view axis (colour 3, no shadows), the object forced to roughness 1 /
    metallic 0, ambient 0.25."""
    scene = S.cube_lookat_scene(sl, size)
    obj = scene.objects[0]
    obj.metallic, obj.roughness = 0.0, 1.0
    scene.light_directions = torch.tensor([[-1.0, 0.0, 0.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]])
    scene.light_colors = torch.tensor([[3.0, 3.0, 3.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]])
    scene.ambient_light = torch.tensor([0.25, 0.25, 0.25])
    scene.manual_exposure = 1.0
    return scene


def lambert_kat_expected(albedo=0.8, light=3.0, ambient=0.25):
    """render_shader.frag:272-373 by hand for N = V = L (the centre of the face): with roughness 1 the GGX lobe is flat
    (alpha^2 = 1: D = 1 / pi), Smith's G = 1 at N.V = N.L = 1 (k = (1 + 1)^2 / 8 = 1/2: x / (x / 2 + 1/2)), Schlick's term
    vanishes at normal incidence (F = F0 = 0.04), the specular denominator is 4; kD = 1 - F for a dielectric."""
    F = 0.04
    specular = (1.0 / math.pi) * 1.0 * F / 4.0
    return ((1.0 - F) * albedo / math.pi + specular) * light * 1.0 + ambient * albedo


def test_fronto_parallel_lambert_known_answer(sl, oracle):
    """A radiance that can be computed by hand (no part of it is this repository's choice): the HDR value the fragment
    shader writes at the centre of a fronto-parallel matte face lit along the view axis."""
    scene = lambert_kat_scene(sl)
    r = oracle_render(oracle, [scene], flags=_abi.OUT_ALL, want_hdr=True)
    W, H = scene.viewport
    px = r.hdr[0, H // 2, W // 2]
    want = lambert_kat_expected()
    assert r.instance[0, H // 2, W // 2, 0] == 1
    assert np.allclose(px[:3], want, rtol=2e-5), (px, want)          # (the pixel centre is half a pixel off the axis: 1e-6)
    assert px[3] == 1.0
    # across the face N.L = 1 stays, N.V falls off with the view angle: the corner of the silhouette is darker by Fresnel / G only
    cov = r.instance[0, :, :, 0] == 1
    face = r.hdr[0][cov][:, 0]
    assert face.max() <= want * (1 + 1e-3) and face.min() > 0.97 * want


def test_ssao_is_exactly_one_on_the_open_plane(sl, oracle):
    """What the HIP path's SSAO pass relies on to skip its 64 taps (slhip_render.hip k_ssao_mask), held against the full loop of the
    restatement: on the background plane, where every texel a tap can reach -- within fx radius sqrt(1 + (X/Z)^2) / (Z - radius) + 2
    pixels -- belongs to the plane or to the cleared background and no tap leaves the image, no sample is occluded, the occlusion
    is 1 - 0 / 64 = 1 and the blurred factor leaves the colour bit for bit what it is without SSAO.  The rule below is the
    kernel's, tile by tile (8 x 8), in numpy."""
    from scipy import ndimage

    scene = S.clutter_scene(sl, 5, n_objects=6, size=(640, 480))
    scene.set_camera_look_at(torch.tensor([2.2, -1.4, 1.9]), torch.tensor([0.0, 0.0, 0.1]))
    W, H = scene.viewport
    f0 = _abi.OUT_ALL | _abi.RENDER_SHADOWS
    r1 = oracle_render(oracle, [scene], flags=f0 | _abi.RENDER_SSAO, want_hdr=True)
    r0 = oracle_render(oracle, [scene], flags=f0, want_hdr=True)
    P = scene.projection_matrix().numpy()
    fx, fy = P[0, 0] * W / 2, P[1, 1] * H / 2
    tmax = max((1 + abs(P[0, 2])) / P[0, 0], (1 + abs(P[1, 2])) / P[1, 1])
    reach = max(fx, fy) * 0.1 * math.sqrt(1 + tmax * tmax) * 1.001
    inst, z = r1.instance[0, :, :, 0], r1.cam_coord[0, :, :, 2]
    geo = z < 2999.0
    other = (geo & (inst != 0)).reshape(H // 8, 8, W // 8, 8).any(axis=(1, 3))
    zmin = np.where(geo, z, np.inf).reshape(H // 8, 8, W // 8, 8).min(axis=(1, 3))
    skip = np.zeros((H // 8, W // 8), bool)
    for ty in range(H // 8):
        for tx in range(W // 8):
            if other[ty, tx] or not np.isfinite(zmin[ty, tx]) or zmin[ty, tx] <= 0.2:
                continue
            R = int(math.ceil(reach / (zmin[ty, tx] - 0.1))) + 2
            x0, x1, y0, y1 = tx * 8 - R, tx * 8 + 7 + R, ty * 8 - R, ty * 8 + 7 + R
            if x0 < 0 or y0 < 0 or x1 >= W or y1 >= H:
                continue
            skip[ty, tx] = not other[y0 // 8:y1 // 8 + 1, x0 // 8:x1 // 8 + 1].any()
    assert skip.mean() > 0.15                                         # the case is exercised
    px = np.kron(skip, np.ones((8, 8), bool))
    safe = ndimage.binary_erosion(px, structure=np.ones((5, 5), bool), border_value=0)   # the 4 x 4 blur reads skipped tiles only
    differs = (r1.hdr[0].view(np.uint32) != r0.hdr[0].view(np.uint32)).any(axis=2)
    assert differs.sum() > 1000 and not (differs & safe).any()


def ssao_tables_from_the_generator():
    """The 4 x 4 noise tile and the 64 kernel samples from tools/gen_ssao_tables.py (std::mt19937 restated, pinned bit for bit to
    libstdc++'s stream: test_oracle_constants.py::test_ssao_random_stream_is_libstdcxx) -- not from the oracle, not from the kernels."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("gen", os.path.join(os.path.dirname(__file__), "..", "tools", "gen_ssao_tables.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    noise, kern = gen.tables()
    return np.asarray(noise, np.float64).reshape(16, 3), np.asarray(kern, np.float64).reshape(64, 3)


def inner_corner_kat_scene(sl):
    """A right-angle inner corner: floor z = 0 and wall x = 0, two faces of 4 m cubes, seen from (0.9, 0.35, 0.8)."""
    a = 2.0
    scene = sl.Scene((640, 480))
    m = sl.Mesh(S.CUBE, physics=False)
    m.center_bbox()
    m.scale_to_bbox_diagonal(2 * a * math.sqrt(3.0))
    for c in ((0.0, 0.0, -a), (-a, 0.0, a)):
        o = sl.Object(m)
        pose = np.eye(4, dtype=np.float32)
        pose[:3, 3] = c
        o.set_pose(torch.from_numpy(pose))
        scene.add_object(o)
    scene.set_camera_look_at(torch.tensor([0.9, 0.35, 0.8]), torch.tensor([0.0, 0.0, 0.02]))
    return scene


def inner_corner_kat_check(scene, ao, cam_coord, instance):
    """ao [H, W]: the unblurred occlusion of the SSAO pass; cam_coord [H, W, >= 3], instance [H, W] of the same picture.  The
    occlusion is recomputed in float64 numpy from the geometry: the depth "texture" is known in closed form (the nearer of two
    ray-plane intersections), the hemisphere basis comes from the 4 x 4 noise tile (Gram-Schmidt against the plane's normal), then
    the 64 kernel samples at radius 0.1, their projection, the analytic depth along the ray through the sample's window position,
    the 2.5 mm bias and the smoothstep range check (ssao_shader.frag:20-56).  The pass samples a rasterised depth plane bilinearly
    instead: a sample within 0.4 mm of the threshold may fall on either side (`near`), everything else must agree to the sample."""
    W, H = scene.viewport
    P = scene.projection_matrix().numpy().astype(np.float64)
    T = scene.camera_pose().numpy().astype(np.float64)                    # camera -> world
    Rcw, tcw = T[:3, :3], T[:3, 3]
    noise, kern = ssao_tables_from_the_generator()
    # the two planes in camera space, n . X = c, each valid on its side of the crease (floor: world x >= 0, wall: world z >= 0)
    planes = [(Rcw.T @ np.array(nw), -float(np.array(nw) @ tcw), ax) for nw, ax in (((0.0, 0.0, 1.0), 0), ((1.0, 0.0, 0.0), 2))]

    def first_hit(u, v):
        d = np.array([(u - P[0, 2]) / P[0, 0], (v - P[1, 2]) / P[1, 1], 1.0])      # w = camera z: the ray X = z d
        best, which = np.inf, -1
        for k, (nc, c, ax) in enumerate(planes):
            t = c / (nc @ d)
            if t > 0 and (Rcw @ (t * d) + tcw)[ax] >= -1e-9 and t < best:
                best, which = t, k
        return best, which, d

    def analytic(i, j):
        z, which, d = first_hit((i + 0.5) / W * 2 - 1, (j + 0.5) / H * 2 - 1)
        frag, n = z * d, planes[which][0]
        rv = noise[(j & 3) * 4 + (i & 3)]
        rv = rv / np.linalg.norm(rv)
        tg = rv - n * (rv @ n)
        tg /= np.linalg.norm(tg)
        bt = np.cross(n, tg)
        occ, near = 0.0, 0
        for s in kern:
            sp = frag + 0.1 * (tg * s[0] + bt * s[1] + n * s[2])
            clip = P @ np.array([sp[0], sp[1], sp[2], 1.0])
            sd = first_hit(clip[0] / clip[3], clip[1] / clip[3])[0]
            near += abs(sd - (sp[2] - 0.0025)) < 4e-4
            if sd <= sp[2] - 0.0025:
                t = min(max(0.1 / abs(frag[2] - sd), 0.0), 1.0)
                occ += t * t * (3.0 - 2.0 * t)
        return 1.0 - occ / 64.0, frag, near

    pix = []
    for j in range(192, 300, 4):                                          # the crease: where a row changes from floor to wall
        row = instance[j]
        x = int(np.argmax(row != row[0]))
        assert 60 < x < W - 60
        pix += [(x + dx, j) for dx in (-41, -17, -6, -2, 1, 5, 14, 37)]
    got, want, slack = [], [], []
    for i, j in pix:
        A, frag, near = analytic(i, j)
        assert np.abs(frag - cam_coord[j, i, :3]).max() < 2e-5            # the geometry the pass starts from is the analytic one
        got.append(float(ao[j, i])); want.append(A); slack.append(near)
    got, want, slack = np.array(got), np.array(want), np.array(slack)
    assert (want < 0.9).mean() > 0.3 and want.min() < 0.7 and want.max() == 1.0      # the corner darkens, the open faces do not
    assert np.all(np.abs(got - want) <= slack / 64.0 + 1e-6), np.abs(got - want).max()
    assert np.abs(got - want).mean() < 2e-3


def test_ssao_inner_corner_known_answer(sl, oracle):
    """The SSAO pass against a computation that shares nothing with it but the pinned sample tables (inner_corner_kat_check)."""
    scene = inner_corner_kat_scene(sl)
    r = oracle_render(oracle, [scene], flags=_abi.OUT_ALL)
    ao = oracle.ssao_pass(scene.projection_matrix().numpy().astype(np.float32), r.cam_coord[0], r.normals[0])
    inner_corner_kat_check(scene, ao, r.cam_coord[0], r.instance[0, :, :, 0])


def cast_shadow_kat_scene(sl, light_colour):
    """A cube (half edge 0.08) floating at z = 0.3 over a matte floor (the top face z = 0 of a 1 m cube) under ONE directional light."""
    scene = sl.Scene((640, 480))
    big = sl.Mesh(S.CUBE, physics=False)
    big.center_bbox()
    big.scale_to_bbox_diagonal(2 * 0.5 * math.sqrt(3.0))
    small = sl.Mesh(S.CUBE, physics=False)
    small.center_bbox()
    small.scale_to_bbox_diagonal(2 * 0.08 * math.sqrt(3.0))
    for m, c in ((big, (0.0, 0.0, -0.5)), (small, (0.0, 0.0, 0.3))):
        o = sl.Object(m)
        pose = np.eye(4, dtype=np.float32)
        pose[:3, 3] = c
        o.set_pose(torch.from_numpy(pose))
        o.metallic, o.roughness = 0.0, 1.0
        scene.add_object(o)
    scene.set_camera_look_at(torch.tensor([1.0, 0.8, 0.9]), torch.tensor([0.0, 0.0, 0.05]))
    ld = np.array([-0.3, 0.2, -1.0], np.float32)
    ld /= np.linalg.norm(ld)
    scene.light_directions = torch.tensor([ld.tolist(), [0.0] * 3, [0.0] * 3])
    scene.light_colors = torch.tensor([[light_colour] * 3, [0.0] * 3, [0.0] * 3])
    scene.ambient_light = torch.tensor([0.1, 0.1, 0.1])
    scene.manual_exposure = 1.0
    return scene, ld


def cast_shadow_kat_check(scene, ld, cam_coord, instance, hs, hn, ha):
    """hs / hn / ha: the float pictures [H, W, 3] with shadows, without, and with the light switched off; cam_coord [H, W, >=3],
    instance [H, W] of any of them.  A floor point is in shadow iff the ray from it against the light's direction meets the cube
    (slab test) -- nothing of the shadow matrices, the map or the bias enters that."""
    from scipy import ndimage

    T = scene.camera_pose().numpy().astype(np.float64)
    Xw = cam_coord[:, :, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    top = (instance == 1) & (np.abs(Xw[:, :, 2]) < 1e-4)                                      # the floor's upper face
    d = -ld.astype(np.float64)
    lo, hi = np.array([-0.08, -0.08, 0.22]), np.array([0.08, 0.08, 0.38])
    t1, t2 = (lo - Xw) / d, (hi - Xw) / d
    tn, tf = np.minimum(t1, t2).max(axis=2), np.maximum(t1, t2).min(axis=2)
    shadow = top & (tn <= tf) & (tf > 0)
    k = np.ones((13, 13), bool)
    core_shadow, core_lit = ndimage.binary_erosion(shadow, structure=k), ndimage.binary_erosion(top & ~shadow, structure=k)
    assert core_shadow.sum() > 1000 and core_lit.sum() > 50000
    assert np.array_equal(hs[core_lit], hn[core_lit])
    assert np.array_equal(hs[core_shadow], ha[core_shadow])
    assert np.allclose(ha[core_shadow], 0.1 * 0.8, rtol=1e-6) and hn[core_lit].min() > 5 * 0.08     # ambient * albedo against the lit floor
    # the outline, away from the floor's own rim (its side faces are in the map too -- front faces are culled in the shadow pass --
    # and a handful of rim pixels compare against them)
    inner = ndimage.binary_erosion(top, structure=np.ones((5, 5), bool))
    rendered = inner & (np.abs(hs - hn).max(axis=2) > 1e-6)
    band = ndimage.binary_dilation(shadow, structure=np.ones((7, 7), bool)) & ~ndimage.binary_erosion(shadow, structure=np.ones((7, 7), bool))
    assert not ((rendered ^ (shadow & inner)) & ~band).any()
    assert abs(int(rendered.sum()) - int((shadow & inner).sum())) < 0.1 * shadow.sum()


def test_cast_shadow_known_answer(sl, oracle):
    """The shadow pass (shadow_shader.vert, render_pass.cpp:131-211, the PCF of render_shader.frag) against plain geometry: a cube
    floating over a matte floor under ONE directional light.  Well inside the analytic shadow the picture must be the ambient-only
    picture, well outside it the picture without shadows, bit for bit (every PCF tap agrees there); the rendered outline may differ
    from the analytic one by the map's texel and the 4 x 4 taps only: a band of a few pixels."""
    lit_scene, ld = cast_shadow_kat_scene(sl, 3.0)
    with_shadow = oracle_render(oracle, [lit_scene], flags=_abi.OUT_ALL | _abi.RENDER_SHADOWS, want_hdr=True)
    without = oracle_render(oracle, [lit_scene], flags=_abi.OUT_ALL, want_hdr=True)
    ambient = oracle_render(oracle, [cast_shadow_kat_scene(sl, 0.0)[0]], flags=_abi.OUT_ALL, want_hdr=True)
    cast_shadow_kat_check(lit_scene, ld, with_shadow.cam_coord[0], with_shadow.instance[0, :, :, 0],
                          with_shadow.hdr[0][:, :, :3], without.hdr[0][:, :, :3], ambient.hdr[0][:, :, :3])


def auto_exposure_kat_scene(sl, light):
    """lambert_kat_scene with the auto exposure of the reference (manualExposure < 0) and a light of the given colour."""
    scene = lambert_kat_scene(sl, (320, 240))
    scene.light_colors = torch.tensor([[light, light, light], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]])
    scene.manual_exposure = -1.0
    return scene


def auto_exposure_kat_expected():
    """tone_map_shader.frag:102-131 by hand for a two-tone picture -- a uniformly lit grey face of radiance L over the cleared
    background (0, 0, 0, 0): the 1 x 1 mip level is (f L, f L, f L, f) for a coverage f, so avg.rgb / avg.a = L whatever f is;
    lum = 0.1 L (the luma weights add up to 1); the face's Y = L is divided by 9.6 x 0.1 L + 1e-4: 1 / 0.96 whatever L is; grey
    stays grey through Yxy and back; ACES(1 / 0.96) = 0.8123; no gamma (the shader's last line overwrites it): 207."""
    x = 1.0 / 0.96
    v = (x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14)
    return int(math.floor(v * 255.0 + 0.5))


def test_auto_exposure_of_a_two_tone_picture_known_answer(sl, oracle):
    want = auto_exposure_kat_expected()
    assert want == 207
    pics = []
    for light in (3.0, 30.0):
        scene = auto_exposure_kat_scene(sl, light)
        r = oracle_render(oracle, [scene], flags=_abi.OUT_ALL)
        W, H = scene.viewport
        face = r.instance[0, :, :, 0] == 1
        assert 0.02 < face.mean() < 0.5
        px = r.rgb[0][face][:, :3].astype(int)
        # the face is uniform to 3 % (Fresnel / G towards its corners): every pixel within 2 steps of the hand-computed value,
        # the centre -- the brightest spot, 1.5 % above the mean -- within 1
        assert abs(int(r.rgb[0, H // 2, W // 2, 0]) - want) <= 1 and np.abs(px - want).max() <= 2, (px.min(), px.max())
        assert (r.rgb[0][~face] == 0).all()
        pics.append(r.rgb[0].copy())
    # ten times the light: the same picture (one step of rounding at most)
    assert np.abs(pics[0].astype(int) - pics[1].astype(int)).max() <= 1


def depth_peel_kat_scene(sl):
    """Two fronto-parallel square sheets on the optical axis: a small one 3 m from the camera, a large one 5 m from it."""
    near = S.delaunay_sheet(sl, 40, seed=1, half=0.4)
    far = S.delaunay_sheet(sl, 40, seed=2, half=1.2)
    scene = S.sheet_scene(sl, near, roll_deg=0.0, distance=3.0)
    o = sl.Object(far)
    cam = scene.camera_pose().numpy().astype(np.float64)
    P = np.eye(4)
    P[:3, :3] = np.diag([1.0, -1.0, -1.0])
    P[:3, 3] = [0.0, 0.0, 5.0]
    o.set_pose(torch.from_numpy((cam @ P).astype(np.float32)))
    scene.add_object(o)
    return scene


def depth_peel_kat_check(scene, first, second):
    """first / second: (instance [H, W], depth [H, W]) of the plain render and of the render peeled by it.  Where the near sheet
    covers the far one the second layer IS the far sheet (instance 2 at depth 5), where the first layer already shows the far
    sheet or nothing the second layer is empty -- and the near sheet's outline is the analytic square |x|, |y| <= 0.4 at z = 3."""
    (i0, d0), (i1, d1) = first, second
    H, W = i0.shape
    P = scene.projection_matrix().numpy()
    fx, fy, cx, cy = P[0, 0] * W / 2, P[1, 1] * H / 2, W / 2, H / 2
    xs, ys = np.meshgrid(np.arange(W) + 0.5, np.arange(H) + 0.5)
    inside_near = (np.abs((xs - cx) / fx * 3.0) < 0.4 - 0.02) & (np.abs((ys - cy) / fy * 3.0) < 0.4 - 0.02)
    outside_near = (np.abs((xs - cx) / fx * 3.0) > 0.4 + 0.02) | (np.abs((ys - cy) / fy * 3.0) > 0.4 + 0.02)
    assert inside_near.sum() > 1000
    assert (i0[inside_near] == 1).all() and np.allclose(d0[inside_near], 3.0, atol=1e-5)
    assert (i1[inside_near] == 2).all() and np.allclose(d1[inside_near], 5.0, atol=1e-5)
    assert (i1[outside_near] == 0).all()
    far_only = outside_near & (i0 == 2)
    assert far_only.sum() > 1000 and np.allclose(d0[far_only], 5.0, atol=1e-5)


def test_depth_peel_second_layer_known_answer(sl, oracle):
    scene = depth_peel_kat_scene(sl)
    flags = _abi.OUT_COORD | _abi.OUT_INSTANCE
    r0 = oracle_render(oracle, [scene], flags=flags)
    r1 = oracle_render(oracle, [scene], flags=flags, depth_peel=r0.coord)
    depth_peel_kat_check(scene, (r0.instance[0, :, :, 0], r0.coord[0, :, :, 3]), (r1.instance[0, :, :, 0], r1.coord[0, :, :, 3]))
