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
