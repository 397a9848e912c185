"""Image-based lighting ('next' row f1) on the GPU: the precompute kernels against the oracle
(oracle/ibl_ref.c, same sampling rules and operation order; libm vs device transcendentals differ in
the last bits, so float tolerances apply), the `.ibl` / Radiance-HDR loaders, and the lights an sIBL
file names."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SMALL = dict(env_size=32, env_levels=6, irr_size=4, pre_size=16, pre_levels=3, lut_size=16)


def sky(H=64, W=128, seed=0):
    """Smooth sky + ground + a bright sun blob + mild texture, f32 [H,W,3], row 0 = top."""
    rng = np.random.default_rng(seed)
    v = (np.arange(H, dtype=np.float32)[:, None] + 0.5) / H          # 0 top .. 1 bottom
    u = (np.arange(W, dtype=np.float32)[None, :] + 0.5) / W
    img = np.zeros((H, W, 3), np.float32)
    img[..., 0] = 0.3 + 0.5 * (1 - v) + 0.05 * np.sin(12 * u)
    img[..., 1] = 0.4 + 0.4 * (1 - v)
    img[..., 2] = 0.6 + 0.3 * (1 - v) + 0.05 * np.cos(9 * u + 3 * v)
    img[v[:, 0] > 0.55] *= np.array([0.35, 0.3, 0.25], np.float32)
    sun = np.exp(-(((u - 0.3) * 2 * W / H) ** 2 + (v - 0.25) ** 2) * 400.0)
    img += 30.0 * sun[..., None] * np.array([1.0, 0.9, 0.7], np.float32)
    img += 0.02 * rng.random((H, W, 3), dtype=np.float32)
    return img.astype(np.float32)


def rel_close(a, b, rtol, atol):
    return np.abs(a - b) <= atol + rtol * np.abs(b)


def test_precompute_matches_oracle(sl, oracle):
    eq = sky()
    lm = sl.LightMap(eq, sizes=SMALL)
    ref = oracle.light_map_build(eq, SMALL)
    env = lm.env.cpu().numpy()
    # equirect -> cube: atan2 / asin differ in the last bits -> sub-texel shifts of ~1e-6 px
    assert rel_close(env, ref["env"], 1e-4, 1e-4).all(), np.abs(env - ref["env"]).max()
    # the mip chain is exact arithmetic on level 0: compare the GPU's chain with a numpy box filter of ITS level 0
    n = SMALL["env_size"]
    lvl0 = env[:24 * n * n].reshape(6, n, n, 4)
    off = 24 * n * n
    cur = lvl0
    for l in range(1, SMALL["env_levels"]):
        m = n >> l
        nxt = ((cur[:, 0::2, 0::2] + cur[:, 0::2, 1::2]) + (cur[:, 1::2, 0::2] + cur[:, 1::2, 1::2])) * np.float32(0.25)
        got = env[off:off + 24 * m * m].reshape(6, m, m, 4)
        assert np.array_equal(got, nxt), "mip level %d" % l
        off += 24 * m * m
        cur = nxt
    assert off == env.size and cur.shape[1] == 1
    irr = lm.irradiance.cpu().numpy()
    assert rel_close(irr, ref["irradiance"], 2e-3, 1e-4).all(), np.abs(irr - ref["irradiance"]).max()
    pre = lm.prefilter.cpu().numpy()
    assert rel_close(pre, ref["prefilter"], 5e-3, 1e-3).mean() > 0.999, np.abs(pre - ref["prefilter"]).max()
    lut = lm.brdf_lut.cpu().numpy()
    assert rel_close(lut, ref["brdf_lut"], 1e-4, 1e-5).all(), np.abs(lut - ref["brdf_lut"]).max()


def test_physical_sanity_of_the_maps(sl):
    """A constant white environment: irradiance = pi * L (cosine-weighted hemisphere), every prefilter level
    = L, and the BRDF table's scale + bias stays in (0, 1]."""
    eq = np.ones((16, 32, 3), np.float32)
    lm = sl.LightMap(eq, sizes=SMALL)
    irr = lm.irradiance.cpu().numpy().reshape(-1, 4)[:, :3]
    # the reference's Riemann sum (delta = 0.02) of cos * sin over the hemisphere, times pi / nrSamples
    assert np.abs(irr - irr.mean()).max() < 1e-4 and abs(irr.mean() - 1.0) < 0.02
    pre = lm.prefilter.cpu().numpy().reshape(-1, 4)[:, :3]
    assert np.abs(pre - 1.0).max() < 1e-4
    lut = lm.brdf_lut.cpu().numpy().reshape(SMALL["lut_size"], SMALL["lut_size"], 2)
    s = lut.sum(axis=2)
    assert (s > 0).all() and (s <= 1.0 + 1e-3).all()
    assert lut[0, -1, 0] > 0.9          # smooth surface seen head-on: almost all of F0 comes back


def write_rgbe(path, img, rle):
    H, W, _ = img.shape
    m = img.max(axis=2)
    e = np.where(m > 1e-32, np.ceil(np.log2(np.maximum(m, 1e-32)) + 1e-9), 0).astype(np.int32)
    scale = np.where(m > 1e-32, np.ldexp(1.0, 8 - e), 0.0)
    rgbe = np.zeros((H, W, 4), np.uint8)
    rgbe[..., :3] = np.clip(img * scale[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.where(m > 1e-32, e + 128, 0).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (H, W))
        for y in range(H):
            if not rle:
                f.write(rgbe[y].tobytes())
                continue
            f.write(bytes([2, 2, W >> 8, W & 255]))
            for c in range(4):
                row = rgbe[y, :, c]
                x = 0
                while x < W:
                    run = 1
                    while x + run < W and run < 127 and row[x + run] == row[x]:
                        run += 1
                    if run >= 3:
                        f.write(bytes([128 + run, int(row[x])]))
                        x += run
                    else:
                        n = min(64, W - x)
                        f.write(bytes([n]) + row[x:x + n].tobytes())
                        x += n
    return rgbe


def test_ibl_file_hdr_loader_and_lights(sl, tmp_path):
    from stillleben_amd import light_map

    img = sky(24, 48, seed=3)
    img[:, :8] = 0.5            # constant runs -> RLE runs
    for rle in (False, True):
        p = str(tmp_path / ("sky_%d.hdr" % rle))
        rgbe = write_rgbe(p, img, rle)
        got = light_map.read_radiance_hdr(p)
        e = rgbe[..., 3].astype(np.int32)
        expect = rgbe[..., :3].astype(np.float32) * np.where(e > 0, np.ldexp(1.0, e - 136), 0.0)[..., None].astype(np.float32)
        assert np.array_equal(got, expect.astype(np.float32))
        assert np.abs(got - img).max() <= img.max() / 128.0     # 8-bit mantissa
    ibl = tmp_path / "sky.ibl"
    ibl.write_text("\n".join([
        "[Header]", 'Name = "test sky"',
        "[Reflection]", 'REFfile = "sky_1.hdr"', "REFmap = 1", "REFgamma = 1.0",
        "[Sun]", "SUNcolor = 255,128,0", "SUNmulti = 2.0", "SUNu = 0.25", "SUNv = 0.3",
        "[Light1]", "LIGHTcolor = 0,255,255", "LIGHTu = -0.25", "LIGHTv = 0.5",
    ]) + "\n")
    lm = sl.LightMap(str(ibl), sizes=SMALL)
    assert lm.path.endswith("sky.ibl") and len(lm.light_directions) == 2
    theta, phi = (0.25 + 0.5) * 2 * math.pi, 0.3 * math.pi     # light_map.cpp:310-318
    pos = np.array([math.cos(phi) * math.sin(theta), math.sin(phi) * math.sin(theta), math.cos(theta)], np.float32)
    assert np.allclose(lm.light_directions[0], -pos, atol=1e-6)
    assert np.allclose(lm.light_colors[0], [2.0, 2.0 * 128 / 255, 0.0], atol=1e-6)
    assert np.allclose(lm.light_colors[1], [0.0, 1.0, 1.0], atol=1e-6)
    # the same image given directly
    lm2 = sl.LightMap(str(tmp_path / "sky_1.hdr"), sizes=SMALL)
    assert torch.equal(lm.env, lm2.env) and lm2.light_directions == []
    with pytest.raises(RuntimeError):
        sl.LightMap(str(tmp_path / "missing.ibl"))
    bad = tmp_path / "bad.ibl"
    bad.write_text("[Reflection]\nREFfile = sky_1.hdr\nREFmap = 2\n")
    with pytest.raises(RuntimeError):
        sl.LightMap(str(bad))


def test_render_with_light_map_matches_oracle(sl, oracle):
    """IBL shading term + sky background + the map's sun light, end to end: every geometric output bit
    for bit, rgb to the 8-bit tolerance.  The oracle renders with the maps the GPU built (downloaded), so
    this isolates the fragment stage from the precompute's transcendental differences."""
    import scenes as S
    from stillleben_amd import _abi, _engine
    from stillleben_amd._batch import HostPool, build_batch
    from stillleben_amd._context import engine
    from test_gpu_render import assert_geometry_equal, assert_rgb_close

    eq = sky(48, 96, seed=2)
    sizes = dict(env_size=64, env_levels=7, irr_size=8, pre_size=32, pre_levels=5, lut_size=32)
    lm = sl.LightMap(eq, sizes=sizes)
    lm.light_directions = [np.array([0.3, -0.2, -0.93], np.float32)]       # as a [Sun] group would give
    lm.light_colors = [np.array([3.0, 2.8, 2.5], np.float32)]
    eng = engine()
    scs = []
    for seed, metal, rough in ((5, 0.9, 0.15), (6, 0.1, 0.7)):
        sc = S.clutter_scene(sl, seed, n_objects=5, size=(200, 150), with_bunny=(seed == 6))
        for o in sc.objects:
            o.metallic, o.roughness = metal, rough
        sc.background_plane_size = torch.tensor([0.0, 0.0])      # let the sky show
        sc.light_map = lm
        sc.ambient_light = torch.tensor([0.7, 0.7, 0.7])         # ignored while a light map is bound
        scs.append(sc)
    plain = S.clutter_scene(sl, 7, n_objects=3, size=(200, 150))     # a scene without a light map in the same batch
    scs.append(plain)
    mask = _abi.OUT_ALL
    bufs = eng.render(scs, mask, ssao=True, shadows=True)
    torch.cuda.synchronize()
    pool = HostPool()
    srec, drec, _ = build_batch(scs, pool, with_shadows=True)
    assert list(srec["light_map"]) == [lm._slot + 1, lm._slot + 1, 0]
    assert np.allclose(srec["light_dir"][0, 0, :3], lm.light_directions[0]) and (srec["ambient"][0] == 0).all()
    host_maps = [({"env": lm.env.cpu().numpy(), "irradiance": lm.irradiance.cpu().numpy(), "prefilter": lm.prefilter.cpu().numpy(),
                   "brdf_lut": lm.brdf_lut.cpu().numpy()}, sizes)] * (lm._slot + 1)
    ref = oracle.render(pool.arrays(), srec, drec, 200, 150, mask | _abi.RENDER_SSAO | _abi.RENDER_SHADOWS,
                        shadow_res=_engine.SHADOW_RES, light_maps=host_maps)
    assert_geometry_equal(bufs, ref)
    assert_rgb_close(bufs, ref)
    rgb = bufs.rgb.cpu().numpy()
    inst = bufs.instance.cpu().numpy()[..., 0]
    empty = bufs.coord.cpu().numpy()[..., 0] == 3000.0                    # pixels without geometry (clear value)
    sky_px = rgb[0][empty[0]]
    assert len(sky_px) > 100 and (sky_px[:, 3] == 0).all() and sky_px[:, :3].max() > 0   # environment colour, alpha 0
    assert (rgb[2][empty[2]] == 0).all()                                   # no light map: background stays cleared
    # the IBL term matters: the same scene without the map is darker on the objects
    scs[0].light_map = None
    dark = eng.render([scs[0]], mask, ssao=True, shadows=True).rgb.cpu().numpy()[0]
    assert rgb[0][inst[0] != 0][:, :3].astype(np.int32).sum() != dark[inst[0] != 0][:, :3].astype(np.int32).sum()


def test_background_image_and_exposure_average(sl, oracle):
    """Scene.background_image fills the pixels without geometry (alpha 0) and -- like the sky -- does not
    enter the auto-exposure average, which the reference takes before any background is drawn
    (render_pass.cpp:632-646)."""
    import scenes as S
    from stillleben_amd import _abi, _engine
    from stillleben_amd._batch import HostPool, build_batch
    from stillleben_amd._context import engine
    from test_gpu_render import assert_geometry_equal, assert_rgb_close

    rng = np.random.default_rng(0)
    img = (rng.random((30, 40, 4)) * 255).astype(np.uint8)
    img[:15, :, 0] = 255                                       # red-ish top half
    sc = S.clutter_scene(sl, 8, n_objects=4, size=(160, 120))
    sc.background_plane_size = torch.tensor([0.0, 0.0])
    sc.background_image = sl.Texture(torch.from_numpy(img))
    auto = S.clutter_scene(sl, 8, n_objects=4, size=(160, 120))    # same scene, no background: same exposure expected
    auto.background_plane_size = torch.tensor([0.0, 0.0])
    eng = engine()
    scs = [sc, auto]
    bufs = eng.render(scs, _abi.OUT_ALL, ssao=True, shadows=True)
    torch.cuda.synchronize()
    pool = HostPool()
    srec, drec, _ = build_batch(scs, pool, with_shadows=True)
    assert srec["bg_tex"][0][1] == 40 and srec["bg_tex"][0][2] == 30 and srec["bg_tex"][1][1] == 0
    ref = oracle.render(pool.arrays(), srec, drec, 160, 120, _abi.OUT_ALL | _abi.RENDER_SSAO | _abi.RENDER_SHADOWS,
                        shadow_res=_engine.SHADOW_RES)
    assert_geometry_equal(bufs, ref)
    assert_rgb_close(bufs, ref)
    rgb = bufs.rgb.cpu().numpy()
    empty = bufs.coord.cpu().numpy()[..., 0] == 3000.0
    assert (rgb[0][empty[0]][:, 3] == 0).all() and rgb[0][empty[0]][:, :3].max() > 0 and (rgb[1][empty[1]] == 0).all()
    sc.manual_exposure = 1.0
    fixed = eng.render([sc], _abi.OUT_ALL, ssao=True, shadows=True).rgb.cpu().numpy()[0]
    top, bottom = fixed[:40][empty[0][:40]], fixed[-40:][empty[0][-40:]]
    assert top[:, 0].mean() > bottom[:, 0].mean() + 20         # upright: the red half of the image is at the top
    covered = ~empty[0]
    assert np.array_equal(rgb[0][covered][:, :3], rgb[1][covered][:, :3])   # objects: identical exposure with / without background
