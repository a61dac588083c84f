"""scenes.py's host-side builders -- what every GPU test, the oracle's inputs and bench.py feed the kernels -- against
tests/golden/reference_uniforms.py, a second transcription of the same C# lines (DistanceField's constructor, Uniforms.DistanceField +
DistanceFieldPacked1, Uniforms.ParticleSystem, RenderSphereLightSource's LightVertex, the Environment block) that shares no code with
them.  Byte equality of the structs: a misreading of LightingRenderer.cs:1894-1940 / Uniforms.cs:90-236 would have to be made twice."""
import ctypes
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import reference_uniforms as ref  # noqa: E402
from illuminant_amd import scenes  # noqa: E402


def raw(s):
    return bytes(ctypes.string_at(ctypes.addressof(s), ctypes.sizeof(s)))


FIELDS = [  # (virtual w, h, depth, requested slices, requested resolution, max encoded distance): the configs' fields, the demos', odd ones
    (2048, 2048, 128.0, 32, 0.25, 128), (4096, 4096, 128.0, 32, 0.125, 128), (256, 256, 64.0, 9, 1.0, 128), (1920, 1080, 64.0, 9, 0.25, 320),
    (128, 96, 64.0, 9, 0.5, 128), (256, 192, 96.0, 12, 0.5, 128), (77, 53, 40.0, 7, 1.0, 64), (1000, 300, 33.0, 200, 0.33, 128),
    (8192, 64, 10.0, 2, 1.0, 128), (640, 480, 200.0, 64, 0.05, 128), (333, 777, 12.5, 40, 0.77, 256),
]


@pytest.mark.parametrize("args", FIELDS)
def test_distance_field_layout_and_uniform_block(args):
    theirs = scenes.DistanceFieldLayout(*args)
    mine = ref.ReferenceDistanceField(*args)
    assert (theirs.slice_width, theirs.slice_height, theirs.slice_count, theirs.physical_slice_count, theirs.column_count, theirs.row_count,
            theirs.atlas_width, theirs.atlas_height) == (mine.SliceWidth, mine.SliceHeight, mine.SliceCount, mine.PhysicalSliceCount,
                                                          mine.ColumnCount, mine.RowCount, mine.TextureWidth, mine.TextureHeight)
    assert theirs.resolution == mine.Resolution
    for q in (dict(), dict(max_cone_radius=24.0, power=0.7, step_limit=64, min_step_size=1.0, long_step_factor=0.5),
              dict(max_cone_radius=8.0, power=1.3, step_limit=17, min_step_size=2.5, long_step_factor=0.8)):
        a = theirs.uniforms(**q)
        b = ref.distance_field_uniforms(mine, MaxConeRadius=q.get("max_cone_radius", 24.0), OcclusionToOpacityPower=q.get("power", 1.0),
                                        MaxStepCount=q.get("step_limit", 64), MinStepSize=q.get("min_step_size", 3.0), LongStepFactor=q.get("long_step_factor", 1.0))
        assert raw(a) == raw(b), (args, q)
    # the particle path: the same block, DistanceFieldPacked1 never set
    assert raw(theirs.uniforms(packed1=False)) == raw(ref.distance_field_uniforms(mine, set_packed1=False))
    # a field that is only partly generated (TextureSliceCount.z = valid slices * slice depth)
    mine.ValidSliceCount = 3
    assert raw(theirs.uniforms(valid_slice_count=3)) == raw(ref.distance_field_uniforms(mine))


def test_particle_system_block():
    for cs in (16, 48, 64, 128, 256, 1024, 4096):
        for kw in (dict(), dict(friction=0.1, max_velocity=2048.0, life_decay=1.2), dict(dt_seconds=1.0 / 144, friction=0.02, max_velocity=90.0, life_decay=4.0,
                                                                                        rotation_from_velocity=True, z_to_y=0.5, size=(3.0, 2.0)),
                   dict(collision=(128.0, 0.5, 0.33, 0.05))):
            a = scenes.system_uniforms(cs, **kw)
            b = ref.particle_system_uniforms(cs, kw.get("dt_seconds", 1.0 / 60), Size=kw.get("size", (1.0, 1.0)), Friction=kw.get("friction", 0.0),
                                             MaximumVelocity=kw.get("max_velocity", 9999.0), LifeDecayPerSecond=kw.get("life_decay", 1.0),
                                             Collision=kw.get("collision", (128.0, 0.0, 0.33, 0.0)), RotationFromVelocity=kw.get("rotation_from_velocity", False),
                                             ZToY=kw.get("z_to_y", 0.0))
            assert raw(a) == raw(b), (cs, kw)


def test_sphere_light_vertex():
    rng = np.random.default_rng(5)
    for i in range(200):
        pos = rng.uniform(-100, 4000, 3)
        kw = dict(color=tuple(rng.uniform(0, 1, 4)), opacity=float(rng.uniform(0.1, 1)), intensity_scale=float(rng.uniform(0.5, 2)), ramp_mode=int(rng.integers(0, 3)),
                  casts_shadows=bool(rng.integers(0, 2)), have_distance_field=bool(rng.integers(0, 2)), ao_radius=float(rng.uniform(0, 20)),
                  ao_opacity=float(rng.uniform(0, 1)), falloff_y=float(rng.uniform(0.5, 2)),
                  shadow_distance_falloff=None if i % 2 else float(rng.uniform(1, 50)), shadow_filter=int(rng.integers(-1, 3)),
                  specular=tuple(rng.uniform(0, 1, 3)), specular_power=float(rng.uniform(1, 16)), ramp_offset=float(rng.uniform(-1, 1)), ramp_rate=float(rng.uniform(0.2, 3)))
        radius, ramp = float(rng.uniform(1, 50)), float(rng.uniform(10, 1200))
        a = scenes.sphere_light(tuple(pos), radius, ramp, **kw)
        b = ref.sphere_light_vertex(tuple(pos), radius, ramp, Color=kw["color"], Opacity=kw["opacity"], intensityScale=kw["intensity_scale"], RampMode=kw["ramp_mode"],
                                    CastsShadows=kw["casts_shadows"], have_distance_field=kw["have_distance_field"], AmbientOcclusionRadius=kw["ao_radius"],
                                    AmbientOcclusionOpacity=kw["ao_opacity"], FalloffYFactor=kw["falloff_y"], ShadowDistanceFalloff=kw["shadow_distance_falloff"],
                                    ShadowFilter=kw["shadow_filter"], SpecularColor=kw["specular"], SpecularPower=kw["specular_power"],
                                    RampOffsetAndRate=(kw["ramp_offset"], kw["ramp_rate"]))
        assert raw(a) == raw(b), (i, kw)


def test_environment_block():
    for kw in (dict(), dict(maximum_z=64.0), dict(ground_z=3.0, maximum_z=200.0, z_to_y=0.5, light_occlusion=40.0, render_scale=(0.5, 0.75)),
               dict(gbuffer_size=(1920, 1080)), dict(gbuffer_size=(72, 56), viewport_scale=(2.0, 1.5), viewport_position=(3.0, 2.0), viewport_relative=True),
               dict(z_to_y=0.00005)):
        a = scenes.environment(**kw)
        b = ref.environment_uniforms(GroundZ=kw.get("ground_z", 0.0), MaximumZ=kw.get("maximum_z", 128.0), ZToYMultiplier=kw.get("z_to_y", 0.0), TwoPointFiveD=True,
                                     LightOcclusion=kw.get("light_occlusion", 0.0), RenderScale=kw.get("render_scale", (1.0, 1.0)), gbuffer_size=kw.get("gbuffer_size"),
                                     ViewportScale=kw.get("viewport_scale", (1.0, 1.0)), ViewportPosition=kw.get("viewport_position", (0.0, 0.0)),
                                     GBufferViewportRelative=kw.get("viewport_relative", False))
        assert raw(a) == raw(b), kw
