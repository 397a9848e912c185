"""The constants the render oracle and the HIP kernels restate from the reference's shaders, pinned: tests/golden/
shader_constants.json holds the values as oracle/ref_build/gen_shader_constants.py read them out of the reference's own source
files (SSAO parameters, PCF offsets and bias, BRDF floors, tone-map matrices and ACES coefficients, light defaults, the clear
value) plus the first 288 draws of libstdc++'s mt19937{0xdeadbeef} / uniform_real_distribution<float>(0,1) pair that the SSAO
tables are made of.  Here: the same quantities as oracle/render_ref.c and csrc/slhip_render.hip spell them, value by value."""
import importlib.util
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "shader_constants.json")) as f:
        return json.load(f)


def _src(rel):
    with open(os.path.join(ROOT, rel)) as f:
        return f.read()


def _f(pattern, text, n=1):
    m = re.search(pattern, text)
    assert m, "pattern not found: " + pattern
    vals = [float(x.rstrip("f")) for x in m.groups()]
    return vals[0] if n == 1 else vals


NUM = r"(-?[0-9.]+(?:e-?[0-9]+)?f?)"


@pytest.mark.parametrize("rel", ["oracle/render_ref.c", "stillleben_amd/csrc/slhip_render.hip"])
def test_shader_constants_match_the_reference_sources(golden, rel):
    s = _src(rel)
    g = golden
    # SSAO
    assert _f(r"const float radius = %s, bias = %s;" % (NUM, NUM), s, 2) == [g["ssao"]["radius"], g["ssao"]["bias"]]
    assert _f(r"\* %s;\s*\n?[^\n]*" % NUM, s[s.index("(d - cd) *" if "(d - cd) *" in s else "(dz - cd) *"):]) == g["ssao"]["blur_sharpness"]
    # PCF: 4 x 4 taps at -1.5 + k texels, bias, mean of 16
    lo, hi, step = g["shadow_pcf"]["loop"]
    assert _f(r"\(%s \+ \(float\)(?:xx|k)\) \* scale" % NUM, s) == lo and (hi - lo) / step + 1 == 4
    assert _f(r"pz - %s\)" % NUM, s) == g["shadow_pcf"]["depth_bias"]
    assert _f(r"acc / %s;" % NUM, s) == g["shadow_pcf"]["divisor"] == 16
    # BRDF
    assert _f(r"clampf\(dot3\(normal, V\), %s, 1\.0f\)" % NUM, s) == g["brdf"]["nov_floor"]
    assert _f(r"fmaxf\(roughness_in, %s\)" % NUM, s) == g["brdf"]["min_roughness"]
    assert _f(r"F0\[c\] = %s \* \(1\.0f - metallic\)" % NUM, s) == g["brdf"]["dielectric_specular"]
    assert _f(r"fmaxf\(4\.0f \* NoV \* NdotL, %s\)" % NUM, s) == g["brdf"]["specular_denominator_floor"]
    assert _f(r"(?:camz|cz) - %s <= " % NUM, s) == g["brdf"]["depth_peel_epsilon"]
    # tone map
    assert _f(r"%s \* \(%s \* \(a(?:vg\[)?0\]? / a(?:vg\[)?3\]?\) \+ %s \* \(a(?:vg\[)?1\]? / a(?:vg\[)?3\]?\) \+ %s \*" % (NUM, NUM, NUM, NUM), s, 4) == \
        [g["tone_map"]["lum_twiddle"]] + g["tone_map"]["rgb_to_lum"]
    m = g["tone_map"]["rgb_to_xyz"]
    assert _f(r"X = %s \* c\[0\] \+ %s \* c\[1\] \+ %s \* c\[2\]" % (NUM, NUM, NUM), s, 3) == m[0]
    assert _f(r"Y(?:xy\[0\])? = %s \* c\[0\] \+ %s \* c\[1\] \+ %s \* c\[2\]" % (NUM, NUM, NUM), s, 3) == m[1]
    assert _f(r"Z = %s \* c\[0\] \+ %s \* c\[1\] \+ %s \* c\[2\]" % (NUM, NUM, NUM), s, 3) == m[2]
    for r in range(3):
        assert _f(r"o\[%d\] = %s \* x2 \+ %s \* y2 \+ %s \* z2" % (r, NUM, NUM, NUM), s, 3) == g["tone_map"]["xyz_to_rgb"][r]
    # (the HIP kernel multiplies by the hardware reciprocal of the denominator, the oracle divides)
    assert _f(r"\(x \* \(%s \* x \+ %s\)\) (?:/ |\* frcp)\(x \* \(%s \* x \+ %s\) \+ %s\)" % (NUM, NUM, NUM, NUM, NUM), s, 5) == g["tone_map"]["aces"]
    assert _f(r"/= \(%s \* lum \+ %s\)" % (NUM, NUM), s, 2) == g["tone_map"]["exposure"]
    assert g["tone_map"]["gamma_line_overwritten"]                 # quirk q1: the output is linear
    # clear value, lights
    assert _f(r"(?:INVALID_VALUE|kInvalid =) %s" % NUM, s) == g["render_pass"]["invalid_value"][0]
    assert _f(r"#define SLHIP_NUM_LIGHTS %s" % NUM, _src("include/slhip.h")) == g["lights"]["num_lights"]


def test_ssao_random_stream_is_libstdcxx(golden, oracle):
    """tools/gen_ssao_tables.py restates std::mt19937 + uniform_real_distribution<float>: its draws equal, bit for bit, the
    ones libstdc++ itself produced in the build container; the tables the oracle and the kernels carry follow from them."""
    spec = importlib.util.spec_from_file_location("gen", os.path.join(ROOT, "tools", "gen_ssao_tables.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    g = golden["ssao"]
    rng = gen.MT19937(g["seed"])
    bits = np.array(g["libstdcxx_stream_f32_bits"], dtype=np.uint32)
    assert len(bits) == g["noise_count"] * g["draws_per_noise_texel"] + g["kernel_size"] * g["draws_per_kernel_sample"]
    mine = np.array([np.float32(rng.uniform01()) for _ in range(len(bits))], dtype=np.float32)
    assert np.array_equal(mine.view(np.uint32), bits)
    noise, kernel = gen.tables()
    n2, k2 = oracle.ssao_tables()
    assert np.array_equal(noise.reshape(-1), n2) and np.array_equal(kernel.reshape(-1), k2)
    # the noise tile IS the stream: 2 draws per texel mapped to [-1, 1)
    st = bits.view(np.float32)
    assert np.array_equal(noise[:, 0], np.float32(2.0) * st[0:32:2] - np.float32(1.0))
    assert g["kernel_size_cpp"] == g["kernel_size"] == 64 and g["lerp"] == [0.1, 1.0] and g["scale_divisor"] == 64.0


@pytest.mark.parametrize("rel", ["oracle/synth_ref.c", "stillleben_amd/csrc/slhip_synth.hip", "stillleben_amd/csrc/slhip_records.cpp",
                                 "stillleben_amd/_shadow.py"])
def test_shadow_matrix_literals_match_the_reference(golden, rel):
    """computeFrustumCorners / computeShadowMapMatrix (reference src/render_pass.cpp:69-211) in its four restatements -- the oracle,
    the placement kernel, the host C++ record builder, the per-scene Python mirror: the frustum's NDC z range and clamp, the
    light-space z margin (x 5 to both sides of the mean) and the shadow map's size are the reference's."""
    s = _src(rel)
    g = golden["shadow_matrix"]
    assert g["ndc_near_far"] == [-1.0, 1.0] and g["mean_divisor"] == 2.0
    far_m = _f(r"far\w* = mean_z \+ (?:f32\()?%s\)? \* spread" % NUM, s)
    near_m = _f(r"near\w* = mean_z - (?:f32\()?%s\)? \* spread" % NUM, s)
    assert [far_m, near_m] == g["z_margin"]
    assert _f(r"mean_z = \(near\w* \+ far\w*\) / (?:f32\()?%s" % NUM, s) == g["mean_divisor"]
    assert _f(r"max\w*\((?:std::)?f?max\w*\((?:f32\()?%s\)?, near_obj\)" % NUM, s) == g["ndc_near_clamp"]
    from stillleben_amd import _engine

    assert [_engine.SHADOW_RES] * 2 == g["map_size"]
