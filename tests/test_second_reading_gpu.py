"""The kernels against the second, independent reading of the HLSL directly (tests/golden/second_reading.npz), through the C ABI: the
frame's SDF sample / pair / trace counts and every particle's liveness exactly, floats within the north star's 1e-4."""
import numpy as np
import pytest

from illuminant_amd import abi, native
from tests.test_second_reading import FIX, second
from tests.util import assert_close

pytestmark = pytest.mark.gpu


def test_lit_frame_against_the_second_reading(ctx):
    L = second.lighting_inputs()
    lights = (abi.LightVertex * len(L["lights"]))(*L["lights"])
    sdf = native.DistanceFieldTexture(ctx, L["atlas"], abi.SDF_UNORM16)
    lm = native.Lightmap(ctx, L["width"], L["height"])
    stats = native.render_sphere_lights(ctx, lights, L["env"], L["dfu"], None, sdf, L["ambient"], lm, want_stats=True)
    assert [int(stats.SdfSamples), int(stats.PixelLightPairs), int(stats.TracedPairs)] == [int(v) for v in FIX["light_counts"]]
    assert_close(lm.download(), FIX["lightmap"], "GPU lightmap vs the second reading")
    lm.close(); sdf.close()


def test_particle_step_against_the_second_reading(ctx):
    P = second.particle_inputs()
    cs = P["chunk_size"]
    eng = native.Engine(ctx, cs, P["rnd"])
    sysm = native.System(eng)
    sysm.add_chunk()
    for plane, data in ((abi.PLANE_POSITION, P["pos"]), (abi.PLANE_VELOCITY, P["vel"]), (abi.PLANE_ATTRIBUTES, P["attr"])):
        sysm.upload(0, plane, data)
    d = abi.StepDesc()
    d.FirstChunk, d.ChunkCount = 0, -1
    d.System, d.Update = P["system"], P["update"]
    d.OpCount = 2
    d.Ops[0].Type = abi.OP_GRAVITY
    d.Ops[0].u.Gravity = P["gravity"]
    d.Ops[1].Type = abi.OP_NOISE
    d.Ops[1].u.Noise = P["noise"]
    d.UpdateMode = abi.UPDATE_POSITIONS
    d.Flags = abi.STEP_COUNT_LIVE
    sysm.step(d)
    got_pos = sysm.download(0, abi.PLANE_POSITION)
    assert np.array_equal(got_pos[:, 3] > 0, FIX["position"][:, 3] > 0)
    assert int(sysm.step_counts()[0]) == int((FIX["position"][:, 3] > 0).sum())
    for plane, key in ((abi.PLANE_POSITION, "position"), (abi.PLANE_VELOCITY, "velocity"), (abi.PLANE_RENDER_COLOR, "render_color"), (abi.PLANE_RENDER_DATA, "render_data")):
        assert_close(sysm.download(0, plane), FIX[key], "GPU %s vs the second reading" % key)
    sysm.close(); eng.close()


def _system_with(ctx, cs, rnd, pos, vel, attr):
    eng = native.Engine(ctx, cs, rnd)
    sysm = native.System(eng)
    sysm.add_chunk()
    for plane, data in ((abi.PLANE_POSITION, pos), (abi.PLANE_VELOCITY, vel), (abi.PLANE_ATTRIBUTES, attr)):
        sysm.upload(0, plane, data)
    return eng, sysm


@pytest.mark.parametrize("case", sorted(second.SPAWN_CASES))
def test_spawn_against_the_second_reading(ctx, case):
    S = second.spawn_inputs(case)
    P = second.particle_inputs()
    eng, sysm = _system_with(ctx, S["chunk_size"], S["rnd"], S["pos"], S["vel"], S["attr"])
    sysm.spawn(0, P["system"], S["spawn"])
    before = dict(position=S["pos"], velocity=S["vel"], attributes=S["attr"])
    planes = dict(position=abi.PLANE_POSITION, velocity=abi.PLANE_VELOCITY, attributes=abi.PLANE_ATTRIBUTES)
    want = {k: FIX["spawn_%s_%s" % (case, k)] for k in planes}
    got = {k: sysm.download(0, planes[k]) for k in planes}
    written = np.zeros(len(S["pos"]), bool)
    got_written = np.zeros(len(S["pos"]), bool)
    for k in planes:
        written |= np.any(want[k] != before[k], axis=1)
        got_written |= np.any(got[k] != before[k], axis=1)
    assert np.array_equal(written, got_written), "the spawn kernel wrote different slots than the second reading"
    for k in planes:
        assert np.array_equal(got[k][~written], want[k][~written])
        assert_close(got[k], want[k], "GPU PS_Spawn %s, %s vs the second reading" % (k, case))
    sysm.close(); eng.close()


@pytest.mark.parametrize("area_type", [0, 1, 2, 3, 4, 5])
def test_fma_against_the_second_reading(ctx, area_type):
    P = second.fma_inputs(area_type)
    key = "after_fma_" if area_type == 0 else "after_fma_area%d_" % area_type
    eng, sysm = _system_with(ctx, P["chunk_size"], P["rnd"], P["pos"], P["vel"], P["attr"])
    sysm.fma(0, P["system"], P["fma"])
    assert_close(sysm.download(0, abi.PLANE_POSITION), FIX[key + "position"], "GPU position after PS_FMA vs the second reading")
    assert_close(sysm.download(0, abi.PLANE_VELOCITY), FIX[key + "velocity"], "GPU velocity after PS_FMA vs the second reading")
    sysm.close(); eng.close()


@pytest.mark.parametrize("case", sorted(second.COLLISION_CASES))
def test_collision_update_against_the_second_reading(ctx, case):
    import ctypes
    Cn = second.collision_inputs(case)
    cs = Cn["chunk_size"]
    eng, sysm = _system_with(ctx, cs, second.particle_inputs()["rnd"], Cn["pos"], Cn["vel"], Cn["attr"])
    sdf = native.DistanceFieldTexture(ctx, Cn["atlas"], abi.SDF_UNORM16)
    sysm.set_distance_field(sdf)
    native.check(native.lib().ilm_debug_step_sdf_samples(ctx.handle, 1, None))
    sysm.update(0, Cn["system"], Cn["update"], df=Cn["dfu"])
    lookups = ctypes.c_uint64(0)
    native.check(native.lib().ilm_debug_step_sdf_samples(ctx.handle, 0, ctypes.byref(lookups)))
    assert int(lookups.value) == int(FIX["collision_%s_samples" % case][0]), "sampleDistanceFieldEx calls differ from the second reading"
    want = {k: FIX["collision_%s_%s" % (case, k)] for k in ("position", "velocity", "render_color", "render_data")}
    got_pos = sysm.download(0, abi.PLANE_POSITION)
    assert np.array_equal(got_pos[:, 3] > 0, want["position"][:, 3] > 0)
    for plane, key in ((abi.PLANE_POSITION, "position"), (abi.PLANE_VELOCITY, "velocity"), (abi.PLANE_RENDER_COLOR, "render_color"), (abi.PLANE_RENDER_DATA, "render_data")):
        assert_close(sysm.download(0, plane), want[key], "GPU collision update %s, %s vs the second reading" % (key, case))
    sdf.close(); sysm.close(); eng.close()


def test_lit_frame_under_a_gbuffer_against_the_second_reading(ctx):
    G = second.lighting_gbuffer_inputs()
    lights = (abi.LightVertex * len(G["lights"]))(*G["lights"])
    sdf = native.DistanceFieldTexture(ctx, G["atlas"], abi.SDF_UNORM16)
    gb = native.GBufferTexture(ctx, G["gbuffer"], abi.GBUFFER_FLOAT4)
    lm = native.Lightmap(ctx, G["width"], G["height"])
    stats = native.render_sphere_lights(ctx, lights, G["env"], G["dfu"], gb, sdf, G["ambient"], lm, want_stats=True)
    assert [int(stats.SdfSamples), int(stats.PixelLightPairs), int(stats.TracedPairs)] == [int(v) for v in FIX["light_counts_gbuffer"]]
    got = lm.download()
    assert np.array_equal(got[..., 3], FIX["lightmap_gbuffer"][..., 3])
    assert_close(got, FIX["lightmap_gbuffer"], "GPU lightmap under a G-buffer vs the second reading")
    lm.close(); gb.close(); sdf.close()


def test_noise_under_an_area_against_the_second_reading(ctx):
    P = second.noise_area_inputs()
    eng, sysm = _system_with(ctx, P["chunk_size"], P["rnd"], P["pos"], P["vel"], P["attr"])
    sysm.noise(0, P["system"], P["noise"])
    assert_close(sysm.download(0, abi.PLANE_POSITION), FIX["after_area_noise_position"], "GPU position after PS_Noise under an area vs the second reading")
    assert_close(sysm.download(0, abi.PLANE_VELOCITY), FIX["after_area_noise_velocity"], "GPU velocity after PS_Noise under an area vs the second reading")
    sysm.close(); eng.close()
