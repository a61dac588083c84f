"""Pins the oracle's particle-light / light-probe restatement (oracle/ilm_oracle_lights.c) on the hand-derived
closed forms of tests/golden/lights_ext.json.  No GPU."""
import numpy as np
import pytest

from illuminant_amd import abi, scenes
from tests import lights_common as lc


@pytest.mark.parametrize("index", range(len(lc.load_cases())))
def test_closed_form_case(oracle, index):
    lc.check_case(lc.load_cases()[index], lc.OracleBackend(oracle))


def test_fp16_per_light_blend_model_closed_form(oracle):
    """orc_set_lightmap_blend(1): clear to half(Ambient), then dst = half(float(dst) + float(half(src))) per light (the reference's
    HalfVector4 lightmap under additive ROP blending, LightingRenderer.cs:476-479,:206).  Lights without shadows on an open ground plane:
    the per-light contribution is the fp32 model's difference between consecutive prefixes, so the model can be replayed in numpy."""
    w, h = 24, 16
    lights = scenes.random_lights(5, 9, w, h, z=(4.0, 20.0), radius=3.0, ramp=(20.0, 40.0), have_distance_field=False)
    env = scenes.environment()
    dfu = abi.DistanceFieldUniforms()
    amb = (0.0213, 0.0377, 0.0591, 1.0)
    # contribution of light k alone on a zero clear colour (fp32): exactly what the shader outputs for that light
    singles = []
    for k in range(len(lights)):
        one = (abi.LightVertex * 1)(lights[k])
        f, _ = oracle.render_sphere_lights(one, env, dfu, None, None, (0.0, 0.0, 0.0, 0.0), w, h)
        singles.append(f)
    want = np.broadcast_to(np.asarray(amb, np.float32), (h, w, 4)).astype(np.float16)
    for f in singles:
        drawn = f[..., 3:4] > 0                               # the light's quad covered the pixel and the shader did not discard
        nxt = (want.astype(np.float32) + f.astype(np.float16).astype(np.float32)).astype(np.float16)
        want = np.where(drawn, nxt, want)
    got, _ = oracle.render_sphere_lights(lights, env, dfu, None, None, amb, w, h, blend_fp16=True)
    assert np.array_equal(got, want.astype(np.float32))
    plain, _ = oracle.render_sphere_lights(lights, env, dfu, None, None, amb, w, h)
    assert not np.array_equal(plain, got) and np.allclose(plain, got, rtol=9 * 2.0 ** -11, atol=1e-6)
