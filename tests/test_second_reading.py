"""Holds oracle/ to the second, independent reading of the HLSL (tests/golden/second_reading.py -> second_reading.npz; CPU only).

The fixture was computed in float32 numpy directly from the shader text (no fused multiply-adds, libm transcendentals); the oracle is
the C restatement the kernels are compared with.  Integer results must agree exactly: the SDF sample / pixel-light pair / traced-pair
counts of the lit frame, the liveness of every particle slot.  Floats agree to 2e-6 of the component's scale (the oracle fuses the
sampler's multiply-adds where the device does; nothing else differs).  It does not pin parity -- neither side executes the reference --
but a misreading shared by oracle and kernels would have to be made a third time, independently, to survive this test."""
import importlib.util
import os

import numpy as np

from illuminant_amd import abi
from tests.util import assert_close

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("second_reading", os.path.join(HERE, "golden", "second_reading.py"))
second = importlib.util.module_from_spec(spec)
spec.loader.exec_module(second)

FIX = np.load(os.path.join(HERE, "golden", "second_reading.npz"))
TOL = dict(rtol=2e-6, atol=2e-6)


def test_lit_frame_of_the_second_reading(oracle):
    L = second.lighting_inputs()
    lights = (abi.LightVertex * len(L["lights"]))(*L["lights"])
    tex = oracle.make_texture(L["atlas"], abi.SDF_UNORM16)
    frame, stats = oracle.render_sphere_lights(lights, L["env"], L["dfu"], None, tex, L["ambient"], L["width"], L["height"], want_stats=True)
    assert [int(stats.SdfSamples), int(stats.PixelLightPairs), int(stats.TracedPairs)] == [int(v) for v in FIX["light_counts"]]
    # 2e-6 everywhere but where the trace ends within rounding of FULLY_SHADOWED_THRESHOLD: saturate(visibility - 0.075) cancels there and
    # pow(., 0.7) amplifies what is left (two pixels of this frame, 2e-5 relative) -- those must still meet the north star's 1e-4
    assert_close(frame, FIX["lightmap"], "oracle lightmap vs the second reading (north-star tolerance)")
    err = np.abs(frame.astype(np.float64) - FIX["lightmap"]) - 2e-6 * np.abs(FIX["lightmap"]) - 2e-6 * np.abs(FIX["lightmap"]).reshape(-1, 4).max(axis=0)
    assert (err > 0).sum() <= 12, "%d elements of the lightmap differ from the second reading by more than 2e-6" % int((err > 0).sum())
    # the frame is a real one: most pixels lit by several lights, shadows present
    assert (FIX["lightmap"][..., 3] >= 3.0).mean() > 0.5 and FIX["light_counts"][2] > 5000 and FIX["light_counts"][0] > 8 * FIX["light_counts"][2] // 2


def test_particle_passes_of_the_second_reading(oracle):
    P = second.particle_inputs()
    cs = P["chunk_size"]
    n = cs * cs

    def desc(ops, mode):
        d = abi.StepDesc()
        d.FirstChunk, d.ChunkCount = 0, -1
        d.System = P["system"]
        d.Update = P["update"]
        d.OpCount = len(ops)
        for i, (typ, params) in enumerate(ops):
            d.Ops[i].Type = typ
            if typ == abi.OP_GRAVITY:
                d.Ops[i].u.Gravity = params
            else:
                d.Ops[i].u.Noise = params
        d.UpdateMode = mode
        d.Flags = abi.STEP_COUNT_LIVE
        return d

    def fresh():
        return [P["pos"].copy(), P["vel"].copy(), P["attr"].copy(), np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)]

    # pass by pass, as the reference draws them (each technique reads the previous one's targets)
    chunk = fresh()
    oracle.step([chunk], cs, P["rnd"], desc([(abi.OP_GRAVITY, P["gravity"])], abi.UPDATE_NONE))
    assert_close(chunk[1], FIX["after_gravity_velocity"], "velocity after PS_Gravity", **TOL)
    oracle.step([chunk], cs, P["rnd"], desc([(abi.OP_NOISE, P["noise"])], abi.UPDATE_NONE))
    assert_close(chunk[0], FIX["after_noise_position"], "position after PS_Noise", **TOL)
    assert_close(chunk[1], FIX["after_noise_velocity"], "velocity after PS_Noise", **TOL)
    # and the whole Update in one step
    chunk = fresh()
    counts = oracle.step([chunk], cs, P["rnd"], desc([(abi.OP_GRAVITY, P["gravity"]), (abi.OP_NOISE, P["noise"])], abi.UPDATE_POSITIONS), want_counts=True)
    assert np.array_equal(chunk[0][:, 3] > 0, FIX["position"][:, 3] > 0), "liveness differs from the second reading"
    assert int(counts[0]) == int((FIX["position"][:, 3] > 0).sum())
    assert_close(chunk[0], FIX["position"], "PositionAndLife after PS_Update", **TOL)
    assert_close(chunk[1], FIX["velocity"], "Velocity after PS_Update", **TOL)
    assert_close(chunk[3], FIX["render_color"], "RenderColor", **TOL)
    assert_close(chunk[4], FIX["render_data"], "RenderData", **TOL)
    live = FIX["position"][:, 3] > 0
    assert 0.5 < live.mean() < 0.9 and (P["pos"][:, 3] > 0).sum() > live.sum()          # some particles die in this step


import pytest


@pytest.mark.parametrize("case", sorted(second.SPAWN_CASES))
def test_spawn_of_the_second_reading(oracle, case):
    """PS_Spawn: the same slots written (range, alpha discard), everything else untouched bit for bit, the written values within 2e-6."""
    S = second.spawn_inputs(case)
    pos, vel, attr = S["pos"].copy(), S["vel"].copy(), S["attr"].copy()
    oracle.spawn(pos, vel, attr, S["chunk_size"], S["rnd"], S["spawn"])
    want = [FIX["spawn_%s_%s" % (case, k)] for k in ("position", "velocity", "attributes")]
    written = np.any(want[0] != S["pos"], axis=1) | np.any(want[1] != S["vel"], axis=1) | np.any(want[2] != S["attr"], axis=1)
    got_written = np.any(pos != S["pos"], axis=1) | np.any(vel != S["vel"], axis=1) | np.any(attr != S["attr"], axis=1)
    assert np.array_equal(written, got_written), "PS_Spawn wrote different slots than the second reading"
    if case == "polygon_discard":
        assert 100 < written.sum() < 1000          # the discard removes a real share of the range
    else:
        assert written.sum() == 1093
    for got, w, name in zip((pos, vel, attr), want, ("position", "velocity", "attributes")):
        assert np.array_equal(got[~written], w[~written])
        assert_close(got, w, "PS_Spawn %s, %s" % (name, case), **TOL)


@pytest.mark.parametrize("area_type", [0, 1, 2, 3, 4, 5])
def test_fma_of_the_second_reading(oracle, area_type):
    """PS_FMA with its weight from evaluateByTypeId: no area, then ellipsoid / box / cylinder / spheroid / octagon, rotated."""
    P = second.fma_inputs(area_type)
    key = "after_fma_" if area_type == 0 else "after_fma_area%d_" % area_type
    pos, vel = P["pos"].copy(), P["vel"].copy()
    oracle.fma(pos, vel, P["chunk_size"], P["system"], P["fma"])
    assert_close(pos, FIX[key + "position"], "position after PS_FMA, area %d" % area_type, **TOL)
    assert_close(vel, FIX[key + "velocity"], "velocity after PS_FMA, area %d" % area_type, **TOL)
    live = P["pos"][:, 3] > 0
    step = np.abs(FIX[key + "velocity"] - P["vel"])[live].max(axis=1)
    assert step.max() > 1e-3 and (area_type == 0 or step.min() < 0.5 * step.max())          # the area weights the particles differently


@pytest.mark.parametrize("case", sorted(second.COLLISION_CASES))
def test_collision_update_of_the_second_reading(oracle, case):
    """UpdateWithDistanceField: the state machine is discontinuous (a lookup on the other side of a threshold changes a slot wholesale),
    so agreement of every slot to 2e-6 means every branch decision of every particle agreed."""
    Cn = second.collision_inputs(case)
    cs = Cn["chunk_size"]
    n = cs * cs
    got = [Cn["pos"].copy(), Cn["vel"].copy(), Cn["attr"].copy(), np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)]
    oracle.update(got[0], got[1], got[2], got[3], got[4], cs, Cn["system"], Cn["update"], df=Cn["dfu"], sdf=oracle.make_texture(Cn["atlas"], abi.SDF_UNORM16))
    want = {k: FIX["collision_%s_%s" % (case, k)] for k in ("position", "velocity", "render_color", "render_data")}
    assert np.array_equal(got[0][:, 3] > 0, want["position"][:, 3] > 0), "liveness differs from the second reading"
    for k, key in ((0, "position"), (1, "velocity"), (3, "render_color"), (4, "render_data")):
        assert_close(got[k], want[key], "collision update %s, %s" % (key, case), **TOL)
    # the scene takes every branch: redirected and bounced particles carry BOUNCE_DELAY, escaping ones a zero velocity.w next to a speed
    assert (want["velocity"][:, 3] == 3.0).sum() > 500


def test_lit_frame_under_a_gbuffer_of_the_second_reading(oracle):
    """sampleGBuffer's G-buffer branch (viewport-relative addressing, the unshadowed / fullbright encodings, relativeY,
    decodeNormalSpherical), the normal factor, light occlusion, both shadow filters, the three falloff modes, AO and specular."""
    G = second.lighting_gbuffer_inputs()
    lights = (abi.LightVertex * len(G["lights"]))(*G["lights"])
    tex = oracle.make_texture(G["atlas"], abi.SDF_UNORM16)
    gb = oracle.make_texture(G["gbuffer"], abi.GBUFFER_FLOAT4)
    frame, stats = oracle.render_sphere_lights(lights, G["env"], G["dfu"], gb, tex, G["ambient"], G["width"], G["height"], want_stats=True)
    assert [int(stats.SdfSamples), int(stats.PixelLightPairs), int(stats.TracedPairs)] == [int(v) for v in FIX["light_counts_gbuffer"]]
    want = FIX["lightmap_gbuffer"]
    assert np.array_equal(frame[..., 3], want[..., 3]), "the lights drawn per pixel (discards: fullbright, filter, distance) differ"
    assert_close(frame, want, "oracle lightmap under a G-buffer vs the second reading (north-star tolerance)")
    err = np.abs(frame.astype(np.float64) - want) - 2e-6 * np.abs(want) - 2e-6 * np.abs(want).reshape(-1, 4).max(axis=0)
    assert (err > 0).sum() <= 12, "%d elements of the lightmap differ from the second reading by more than 2e-6" % int((err > 0).sum())
    # the fullbright band (G-buffer rows 20..22 = frame rows 18..20 under the scrolled viewport) receives no light at all
    assert not want[18:21, :, 3].any() and (want[..., 3] >= 2.0).mean() > 0.4


def test_noise_under_an_area_of_the_second_reading(oracle):
    """PS_Noise weighted by a rotated box (computeWeight through evaluateByTypeId), ReplaceOldVelocity on."""
    P = second.noise_area_inputs()
    pos, vel = P["pos"].copy(), P["vel"].copy()
    oracle.noise(pos, vel, P["chunk_size"], P["rnd"], P["system"], P["noise"])
    assert_close(pos, FIX["after_area_noise_position"], "position after PS_Noise under an area", **TOL)
    assert_close(vel, FIX["after_area_noise_velocity"], "velocity after PS_Noise under an area", **TOL)
    moved = np.abs(FIX["after_area_noise_position"] - P["pos"]).max(axis=1)
    assert (moved > 1e-3).any() and (moved == 0).any()                         # inside the box's reach and beyond its falloff
