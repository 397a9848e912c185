"""Material inputs of the fragment stage beyond the base colour (R7 of SURVEY.md 8a): normal map with
per-vertex tangents (R1 / R4), metallic-roughness, occlusion and emissive textures
(RenderShader::setMaterial, render_shader.cpp:395-415; render_shader.frag:259-298) and the projected
sticker decal (object.cpp:494-513, vert:89-94, frag:248-256) -- HIP path against the oracle."""
import numpy as np
import pytest
import torch

from stillleben_amd import _abi, _engine, _loaders
from stillleben_amd._batch import HostPool, build_batch

pytestmark = pytest.mark.gpu


def textured_sphere(seed, n_lat=24, n_lon=48, with_tex=("normal", "mr", "occlusion", "emissive", "base")):
    """UV sphere with smooth normals, UVs, computed tangents and procedural 64x64 textures."""
    rng = np.random.default_rng(seed)
    th = np.linspace(0.02, np.pi - 0.02, n_lat, dtype=np.float32)
    ph = np.linspace(0.0, 2 * np.pi, n_lon, dtype=np.float32)
    T, P = np.meshgrid(th, ph, indexing="ij")
    pos = np.stack([np.sin(T) * np.cos(P), np.sin(T) * np.sin(P), np.cos(T)], axis=-1).reshape(-1, 3).astype(np.float32) * 0.1
    nrm = (pos / np.linalg.norm(pos, axis=1, keepdims=True)).astype(np.float32)
    uv = np.stack([P / (2 * np.pi), T / np.pi], axis=-1).reshape(-1, 2).astype(np.float32)
    idx = []
    for i in range(n_lat - 1):
        for j in range(n_lon - 1):
            a, b, c, d = i * n_lon + j, i * n_lon + j + 1, (i + 1) * n_lon + j, (i + 1) * n_lon + j + 1
            idx += [a, c, b, b, c, d]
    idx = np.array(idx, np.uint32)
    m = _loaders.ConsolidatedMesh()
    m.positions, m.normals, m.uvs, m.indices = pos, nrm, uv, idx
    m.colors = np.ones((len(pos), 4), np.float32)
    tan = _loaders.compute_tangents(pos, nrm, uv, idx)
    tan[:, 3] = 1.0                                    # consolidate.cpp:275-279 (quirk kept by the loader)
    m.tangents = np.nan_to_num(tan).astype(np.float32)

    def tex(kind):
        y, x = np.mgrid[0:64, 0:64].astype(np.float32) / 64.0
        t = np.zeros((64, 64, 4), np.float32)
        if kind == "normal":
            nx, ny = 0.35 * np.sin(14 * x), 0.35 * np.cos(10 * y)
            nz = np.sqrt(np.maximum(1 - nx * nx - ny * ny, 0.0))
            t[..., 0], t[..., 1], t[..., 2] = nx * 0.5 + 0.5, ny * 0.5 + 0.5, nz * 0.5 + 0.5
        elif kind == "mr":
            t[..., 1], t[..., 2] = 0.2 + 0.7 * x, 0.1 + 0.8 * y
        elif kind == "occlusion":
            t[..., 0] = 0.3 + 0.7 * (np.sin(9 * x) * 0.5 + 0.5)
        elif kind == "emissive":
            t[..., :3] = np.stack([(x > 0.8), (y > 0.8), (x < 0.1)], axis=-1) * 0.9
        else:
            t[..., :3] = 0.3 + 0.6 * rng.random((64, 64, 3), dtype=np.float32)
        t[..., 3] = 1.0
        return np.round(t * 255).astype(np.uint8)

    names = ["base", "normal", "mr", "occlusion", "emissive"]
    index = {}
    for k in names:
        if k in with_tex:
            index[k] = len(m.textures)
            m.textures.append(tex(k))
    m._tex_alpha = [False] * len(m.textures)
    m.materials = [_loaders.Material(base_color=(0.9, 0.8, 0.7, 1.0), metallic=1.0, roughness=1.0, emissive=(1.0, 0.5, 2.0),
                                     base_texture=index.get("base"), normal_texture=index.get("normal"),
                                     mr_texture=index.get("mr"), occlusion_texture=index.get("occlusion"),
                                     emissive_texture=index.get("emissive"))]
    m.submeshes = [_loaders.SubMesh(0, len(idx), 0)]
    return m


def make_scene(sl, meshes, seed, size=(240, 180)):
    import scenes as S

    rng = np.random.default_rng(seed)
    scene = sl.Scene(size, seed=seed)
    for k, m in enumerate(meshes):
        o = sl.Object(m)
        pose = np.eye(4, dtype=np.float32)
        pose[:3, :3] = S.random_rotation(rng)
        pose[:3, 3] = [0.22 * (k - (len(meshes) - 1) / 2.0), 0.02 * k, 0.12]
        o.set_pose(torch.from_numpy(pose))
        scene.add_object(o)
    scene.set_camera_look_at(torch.tensor([0.1, -0.55, 0.45]), torch.tensor([0.0, 0.0, 0.1]))
    scene.background_plane_size = torch.tensor([2.0, 2.0])
    scene.light_directions = torch.tensor([[0.3, 0.4, -0.85]])
    scene.manual_exposure = 1.0
    return scene


def render_both(sl, oracle, scenes_, light_maps=None):
    from stillleben_amd._context import engine
    from test_gpu_render import assert_geometry_equal, assert_rgb_close

    eng = engine()
    W, H = scenes_[0].viewport
    bufs = eng.render(scenes_, _abi.OUT_ALL, ssao=True, shadows=True)
    torch.cuda.synchronize()
    pool = HostPool()
    srec, drec, _ = build_batch(scenes_, pool, with_shadows=True)
    ref = oracle.render(pool.arrays(), srec, drec, W, H, _abi.OUT_ALL | _abi.RENDER_SSAO | _abi.RENDER_SHADOWS,
                        shadow_res=_engine.SHADOW_RES, light_maps=light_maps)
    assert_geometry_equal(bufs, ref)
    assert_rgb_close(bufs, ref)
    return bufs, ref, drec


def test_all_material_textures_match_oracle(sl, oracle):
    full = sl.Mesh.from_data(textured_sphere(1))
    plain = sl.Mesh.from_data(textured_sphere(2, with_tex=("base",)))
    nrm_only = sl.Mesh.from_data(textured_sphere(3, with_tex=("normal",)))
    scene = make_scene(sl, [full, plain, nrm_only], 11)
    bufs, ref, drec = render_both(sl, oracle, [scene])
    f = drec["flags"]
    want = _abi.DRAW_HAS_NORMAL_TEX | _abi.DRAW_HAS_MR_TEX | _abi.DRAW_HAS_OCCLUSION_TEX | _abi.DRAW_HAS_EMISSIVE_TEX | _abi.DRAW_HAS_BASE_TEX
    assert (f[1] & want) == want and (f[2] & want) == _abi.DRAW_HAS_BASE_TEX and (f[3] & want) == _abi.DRAW_HAS_NORMAL_TEX
    # the normal map shows up in the normals target of the mapped objects
    inst = bufs.instance.cpu().numpy()[0, :, :, 0]
    nrm = bufs.normals.cpu().numpy()[0]
    assert (inst == 1).sum() > 500 and (inst == 3).sum() > 500
    flat = sl.Mesh.from_data(textured_sphere(3, with_tex=()))
    scene2 = make_scene(sl, [full, plain, flat], 11)
    from stillleben_amd._context import engine

    n2 = engine().render([scene2], _abi.OUT_ALL, ssao=True, shadows=True).normals.cpu().numpy()[0]
    assert np.abs(nrm[inst == 3] - n2[inst == 3]).max() > 0.05        # bumps
    assert np.array_equal(nrm[inst == 2], n2[inst == 2])              # untouched object: identical


def test_flat_normal_map_is_a_no_op(sl, oracle):
    """A texture of (128,128,255) decodes to (0.004, 0.004, 1): the shading normal stays within 1 % of the
    geometric one -- and the oracle agrees bit for bit either way."""
    data = textured_sphere(4, with_tex=("normal",))
    data.textures[0][...] = (128, 128, 255, 255)
    m = sl.Mesh.from_data(data)
    ref_m = sl.Mesh.from_data(textured_sphere(4, with_tex=()))
    a, _, _ = render_both(sl, oracle, [make_scene(sl, [m], 5)])
    from stillleben_amd._context import engine

    b = engine().render([make_scene(sl, [ref_m], 5)], _abi.OUT_ALL, ssao=True, shadows=True)
    na, nb = a.normals.cpu().numpy()[0], b.normals.cpu().numpy()[0]
    inst = a.instance.cpu().numpy()[0, :, :, 0]
    assert np.abs(na[inst == 1] - nb[inst == 1]).max() < 0.02


def test_sticker_decal_and_ibl_occlusion(sl, oracle):
    from test_gpu_ibl import sky

    data = textured_sphere(6, with_tex=("base", "occlusion"))
    data.materials[0].emissive[:] = 0.0                        # keep the image out of saturation
    data.materials[0].metallic, data.materials[0].roughness = 0.1, 0.6
    m = sl.Mesh.from_data(data)
    scene = make_scene(sl, [m, m], 21)
    scene.manual_exposure = 0.3
    st = np.zeros((32, 48, 4), np.uint8)
    st[..., 0] = 255                                            # red sticker ...
    st[8:24, 12:36, 1] = 255                                    # ... with a yellow centre
    st[..., 3] = 255
    st[:4, :, 3] = 0                                            # transparent top rows: the base shows through
    o = scene.objects[0]
    o.sticker_texture = sl.Texture2D(torch.from_numpy(st))
    o.sticker_range = [-0.5, -0.4, 0.5, 0.4]
    o.sticker_rotation = [0.0, 0.3826834, 0.0, 0.9238795]       # 45 degrees about y
    sizes = dict(env_size=32, env_levels=6, irr_size=4, pre_size=16, pre_levels=5, lut_size=16)
    lm = sl.LightMap(sky(32, 64, seed=9), sizes=sizes)
    lm.light_directions = [np.array([0.2, 0.3, -0.9], np.float32)]
    lm.light_colors = [np.array([2.0, 2.0, 2.0], np.float32)]
    scene.light_map = lm
    host_maps = [({"env": lm.env.cpu().numpy(), "irradiance": lm.irradiance.cpu().numpy(), "prefilter": lm.prefilter.cpu().numpy(),
                   "brdf_lut": lm.brdf_lut.cpu().numpy()}, sizes)] * (lm._slot + 1)
    bufs, ref, drec = render_both(sl, oracle, [scene], light_maps=host_maps)
    assert drec["flags"][1] & _abi.DRAW_HAS_STICKER and not (drec["flags"][2] & _abi.DRAW_HAS_STICKER)
    # object.cpp:494-513: Magnum's Matrix4{...} takes COLUMNS, so the matrix scales x, y by 2 / diagonal, shifts z by
    # one (twice: its own last column and the explicit translation) and keeps w = 1
    P0 = scene.objects[1].sticker_view_projection()
    d = float(m.bbox.diagonal)
    assert np.allclose(P0, [[2 / d, 0, 0, 0], [0, 2 / d, 0, 0], [0, 0, 1, 2], [0, 0, 0, 1]], atol=1e-5)
    P = o.sticker_view_projection()
    assert np.allclose(P[3], [0, 0, 0, 1]) and abs(P[0, 0] - (2 / d) * np.cos(np.pi / 4)) < 1e-4
    rgb = bufs.rgb.cpu().numpy()[0].astype(np.int32)
    inst = bufs.instance.cpu().numpy()[0, :, :, 0]
    red = (inst == 1) & (rgb[..., 0] > rgb[..., 2] + 40)
    assert red.sum() > 50                                        # the decal is visible on object 1 ...
    assert ((inst == 2) & (rgb[..., 0] > rgb[..., 2] + 80)).sum() < red.sum() // 4     # ... and only there


def quad_mesh(tex, sampler, uv_scale=1.0, uv_shift=0.0):
    """A 0.4 m square in the xy plane with UVs [shift, shift + scale]^2 and one base-colour texture."""
    m = _loaders.ConsolidatedMesh()
    m.positions = np.array([[-0.2, -0.2, 0], [0.2, -0.2, 0], [0.2, 0.2, 0], [-0.2, 0.2, 0]], np.float32)
    m.normals = np.array([[0, 0, 1]] * 4, np.float32)
    m.uvs = (np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32) * uv_scale + uv_shift).astype(np.float32)
    m.colors = np.ones((4, 4), np.float32)
    m.indices = np.array([0, 1, 2, 0, 2, 3], np.uint32)
    m.textures = [tex]
    m.tex_samplers = [sampler]
    m._tex_alpha = [False]
    m.materials = [_loaders.Material(base_color=(1, 1, 1, 1), metallic=0.0, roughness=0.9, base_texture=0)]
    m.submeshes = [_loaders.SubMesh(0, 6, 0)]
    return m


def checker(n=256, cell=1):
    y, x = np.mgrid[0:n, 0:n]
    c = (((x // cell) + (y // cell)) & 1).astype(np.uint8) * 255
    t = np.stack([c, c, c, np.full_like(c, 255)], axis=-1)
    t[: n // 2, :, 2] = 255           # blue tint on the top half so that orientation errors show
    return t.astype(np.uint8)


@pytest.mark.parametrize("sampler,uv_scale", [
    (_abi.SAMPLER_DEFAULT, 1.0),                       # repeat, trilinear
    (0x10 | 0x20 | (1 << 6), 1.0),                     # linear, nearest mip level
    (0x10 | 0x20, 3.0),                                # no mipmaps, repeat x3
    (1 | (1 << 2) | 0x10 | 0x20 | (2 << 6), 2.0),      # clamp to edge, UVs beyond [0,1]
    (2 | (2 << 2) | (2 << 6), 2.5),                    # mirrored repeat, nearest texel + linear between levels
    (0, 1.0),                                          # nearest, base level
])
def test_sampler_modes_and_mipmaps_match_oracle(sl, oracle, sampler, uv_scale):
    import scenes as S

    mesh = sl.Mesh.from_data(quad_mesh(checker(128, 2), sampler, uv_scale, -0.25 if uv_scale > 1 else 0.0))
    scene = sl.Scene((200, 150), seed=1)
    rng = np.random.default_rng(3)
    for k, (z, tilt) in enumerate(((0.6, 0.2), (1.6, 1.1), (3.0, 1.35))):   # near (magnified) .. far and grazing (minified)
        o = sl.Object(mesh)
        pose = np.eye(4, dtype=np.float32)
        c, s_ = np.cos(tilt), np.sin(tilt)
        pose[:3, :3] = np.array([[1, 0, 0], [0, c, -s_], [0, s_, c]], np.float32)
        pose[:3, 3] = [0.35 * (k - 1), 0.0, z]
        o.set_pose(torch.from_numpy(pose))
        scene.add_object(o)
    scene.light_directions = torch.tensor([[0.0, 0.2, 1.0]])
    scene.manual_exposure = 1.0
    del rng, S
    render_both(sl, oracle, [scene])


def test_minified_checkerboard_is_filtered_not_aliased(sl):
    """1-texel checkerboard seen from far away: with mipmaps the quad is uniform mid-grey, with the base level
    only it is aliased noise -- the property mip-mapping exists for."""
    from stillleben_amd._context import engine

    out = {}
    for name, sampler in (("mip", _abi.SAMPLER_DEFAULT), ("nomip", 0x10 | 0x20)):
        mesh = sl.Mesh.from_data(quad_mesh(checker(256, 1), sampler))
        scene = sl.Scene((160, 120), seed=1)
        o = sl.Object(mesh)
        pose = np.eye(4, dtype=np.float32)
        pose[:3, 3] = [0.0, 0.0, 2.5]
        o.set_pose(torch.from_numpy(pose))
        scene.add_object(o)
        scene.ambient_light = torch.tensor([1.0, 1.0, 1.0])
        scene.light_colors = torch.zeros(3, 3)
        scene.manual_exposure = 0.5
        b = engine().render([scene], _abi.OUT_ALL, ssao=False, shadows=False)
        rgb = b.rgb.cpu().numpy()[0].astype(np.float32)
        inst = b.instance.cpu().numpy()[0, :, :, 0]
        inner = np.zeros_like(inst, bool)
        ys, xs = np.nonzero(inst == 1)
        inner[ys.min() + 2: ys.max() - 1, xs.min() + 2: xs.max() - 1] = True
        out[name] = rgb[inner & (inst == 1)][:, 0]
    assert out["mip"].std() < 2.0 and out["nomip"].std() > 10.0 * max(out["mip"].std(), 0.5)


def test_mip_chain_and_gltf_sampler_parsing():
    from stillleben_amd._batch import HostPool, mip_down

    img = np.arange(5 * 6 * 4, dtype=np.uint8).reshape(5, 6, 4)
    nxt = mip_down(img)
    assert nxt.shape == (2, 3, 4)
    assert nxt[0, 0, 0] == (int(img[0, 0, 0]) + img[0, 1, 0] + img[1, 0, 0] + img[1, 1, 0] + 2) // 4
    one = mip_down(np.full((1, 4, 4), 200, np.uint8))
    assert one.shape == (1, 2, 4) and (one == 200).all()
    pool = HostPool()
    off, w, h = pool.add_texture(np.zeros((8, 4, 4), np.uint8))
    assert (off, w, h) == (0, 4, 8) and pool.n_tex_bytes == 4 * (32 + 8 + 2 + 1)     # 4x8, 2x4, 1x2, 1x1
    assert pool.add_texture(np.zeros((3, 3, 4), np.uint8), mips=False)[0] == pool.n_tex_bytes - 36
    doc = {"samplers": [{"magFilter": 9728, "minFilter": 9985, "wrapS": 33071, "wrapT": 33648}, {}]}
    assert _loaders._gltf_sampler(doc, 0) == (1 | (2 << 2) | 0x20 | (1 << 6))
    assert _loaders._gltf_sampler(doc, 1) == _abi.SAMPLER_DEFAULT and _loaders._gltf_sampler(doc, None) == _abi.SAMPLER_DEFAULT
