"""The closed-form cases of tests/golden/lights_ext.json run through a backend (oracle or HIP path)."""
import json
import os

import numpy as np

from illuminant_amd import abi, scenes
from tests.util import assert_close

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CS = 8          # 64-slot chunk
W, H = 32, 16   # lightmap of the fixture scenes


def load_cases():
    with open(os.path.join(GOLDEN, "lights_ext.json")) as f:
        return json.load(f)["cases"]


def particle_light_params(radius, ramp, color=(1, 1, 1, 1), casts_shadows=False, ao_radius=0.0, ao_opacity=0.0, falloff_y=1.0,
                          spec=(0, 0, 0), spec_power=1.0, ramp_mode=0, shadow_distance_falloff=None):
    """_ParticleLightBatchSetup, LightingRenderer.cs:769-790."""
    p = abi.ParticleLightParams()
    p.LightProperties = abi.f4(radius, ramp, ramp_mode, 1.0 if casts_shadows else 0.0)
    p.MoreLightProperties = abi.f4(ao_radius if ao_opacity > 0.001 else 0.0, -99999.0 if shadow_distance_falloff is None else shadow_distance_falloff,
                                   falloff_y, min(max(ao_opacity, 0.0), 1.0))
    p.LightColor = abi.f4(*color)
    p.LightSpecularColor = abi.f4(spec[0], spec[1], spec[2], spec_power)
    p.StippleFactor = 1.0
    return p


def no_field_uniforms():
    """SetDistanceFieldParameters without a field, LightingRenderer.cs:1905-1914."""
    u = abi.DistanceFieldUniforms()
    u.ConeAndMisc = abi.f4(0, 0, 0, 1)
    u.StepAndMisc2 = abi.f4(64, 3, 0, 1)
    u.Extent = abi.f4(0, 0, 128, 0)
    return u


class OracleBackend:
    def __init__(self, oracle):
        self.orc = oracle

    def particle_lights(self, chunk, quad_count, params, lightmap):
        lm = lightmap.copy()
        self.orc.render_particle_lights([chunk], [quad_count], params, scenes.environment(), no_field_uniforms(), None, None, lm)
        return lm

    def probes(self, lights, pp, pn, ramp=None):
        self.orc.set_light_ramp(ramp)
        try:
            return self.orc.render_light_probes(lights, pp, pn, scenes.environment(), no_field_uniforms(), None)
        finally:
            self.orc.set_light_ramp(None)


class GpuBackend:
    def __init__(self, ctx):
        from illuminant_amd import native
        self.native, self.ctx = native, ctx

    def particle_lights(self, chunk, quad_count, params, lightmap):
        native = self.native
        eng = native.Engine(self.ctx, CS, scenes.randomness_table(7))
        sysm = native.System(eng)
        sysm.add_chunk()
        sysm.upload(0, abi.PLANE_POSITION, chunk[0])
        sysm.upload(0, abi.PLANE_RENDER_COLOR, chunk[3])
        lm = native.Lightmap(self.ctx, W, H, abi.LIGHTMAP_FLOAT4)
        # seed the lightmap through a sphere-light pass with zero lights: clear to `ambient`
        native.render_sphere_lights(self.ctx, None, scenes.environment(), no_field_uniforms(), None, None, tuple(lightmap[0, 0]), lm)
        native.render_particle_lights(self.ctx, sysm, params, scenes.environment(), no_field_uniforms(), None, None, lm, quad_counts=[quad_count])
        out = lm.download()
        for x in (lm, sysm, eng):
            x.close()
        return out

    def probes(self, lights, pp, pn, ramp=None):
        self.ctx.set_light_ramp(ramp)
        try:
            return self.native.render_light_probes(self.ctx, lights, pp, pn, scenes.environment(), no_field_uniforms(), None)
        finally:
            self.ctx.set_light_ramp(None)


def check_case(case, backend):
    if case["kind"] == "particle_light":
        n = CS * CS
        chunk = [np.zeros((n, 4), np.float32) for _ in range(5)]
        slot = case.get("slot", 5)
        chunk[0][slot] = case["particle"]["position"]
        chunk[3][slot] = case["particle"]["render_color"]
        params = particle_light_params(case["radius"], case["ramp"], case["light_color"])
        before = np.zeros((H, W, 4), np.float32)
        before[:] = np.asarray(case["lightmap_before"], np.float32)
        out = backend.particle_lights(chunk, case.get("quad_count", n), params, before)
        x, y = case["pixel"]
        assert_close(out[y, x], case["expected"], "particle light %s" % case.get("why", ""), rtol=1e-5)
    elif case["kind"] == "light_probe":
        lights = (abi.LightVertex * len(case["lights"]))()
        for i, l in enumerate(case["lights"]):
            lights[i] = scenes.sphere_light(l["position"], l["radius"], l["ramp"], color=l["color"], casts_shadows=False,
                                            ramp_offset=l.get("ramp_offset", 0.0), ramp_rate=l.get("ramp_rate", 1.0))
        pr = case["probe"]
        pp = np.asarray([pr["position"] + [1.0]], np.float32)
        nrm = pr["normal"] if pr["normal"] is not None else [0.0, 0.0, 0.0]
        pn = np.asarray([nrm + [1.0 if pr["enable_shadows"] else 0.0]], np.float32)
        ramp = np.asarray(case["ramp_texture"], np.float32) if "ramp_texture" in case else None
        out = backend.probes(lights, pp, pn, ramp=ramp)
        assert_close(out[0], case["expected"], "light probe", rtol=1e-5, atol=2e-6)
    else:
        raise AssertionError(case["kind"])
