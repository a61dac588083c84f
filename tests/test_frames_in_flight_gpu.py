"""Frames in flight (r05): sibling contexts (ilm_ctx_create_sibling) render alternate frames on their own streams into their own lightmaps and
READ one distance field and one G-buffer, owned by the first of them.  The reference keeps a ring of lightmaps for the same reason
(BufferRing, Illuminant/Lighting/LightingRenderer.cs:472-485).  What must hold: a frame through a borrowed field / G-buffer is the owner's
frame bit for bit; when the owner regenerates the field between frames -- with nobody synchronising anything -- every frame still sees
exactly the field it was queued behind (the library orders the siblings' reads against the owner's writes with events); unrelated contexts
keep refusing each other's objects."""
import numpy as np
import pytest

from illuminant_amd import abi, native, scenes
from tests import lights_common as lc

pytestmark = pytest.mark.gpu
AMBIENT = (0.05, 0.06, 0.07, 1.0)


def scene(w=512, h=384):
    layout = scenes.DistanceFieldLayout(512, 384, 96.0, 12, 0.5, 128)
    dfu = layout.uniforms(max_cone_radius=24.0, power=0.7, step_limit=64, min_step_size=1.0, long_step_factor=0.5)
    lights = scenes.random_lights(21, 24, w, h, z=(8.0, 48.0), radius=12.0, ramp=(80.0, 260.0))
    return layout, dfu, lights, w, h


def obstruction_set(k, layout):
    return scenes.obstruction_array(scenes.random_obstructions(100 + k, 10 + k % 5, (512, 384), 8.0, 40.0, 50.0))


def test_a_sibling_renders_the_owners_frame_through_the_owners_field(ctx):
    layout, dfu, lights, w, h = scene()
    a = native.Context(0)
    b = a.sibling()
    stranger = native.Context(0)
    field = native.DistanceFieldTexture(a, None, abi.SDF_UNORM16, size=(layout.atlas_width, layout.atlas_height))
    field.render_slices(scenes.render_desc(layout), list(range(0, layout.slice_count, 3)), obstruction_set(0, layout))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    nx = 0.2 * np.sin(xx / 11.0)
    g = scenes.encode_gbuffer(np.stack([nx, np.zeros_like(nx), np.sqrt(1.0 - nx * nx)], axis=-1), 0.0, 4.0 + 3.0 * np.cos(yy / 19.0))
    gb = native.GBufferTexture(a, g, abi.GBUFFER_FLOAT4)
    env = scenes.environment(gbuffer_size=(w, h))
    lm_a, lm_b, lm_s = native.Lightmap(a, w, h), native.Lightmap(b, w, h), native.Lightmap(stranger, w, h)
    try:
        sa = native.render_sphere_lights(a, lights, env, dfu, gb, field, AMBIENT, lm_a, want_stats=True)
        sb = native.render_sphere_lights(b, lights, env, dfu, gb, field, AMBIENT, lm_b, want_stats=True)
        assert (sa.SdfSamples, sa.PixelLightPairs, sa.TracedPairs) == (sb.SdfSamples, sb.PixelLightPairs, sb.TracedPairs) and sa.SdfSamples > 1_000_000
        assert np.array_equal(lm_a.download().view(np.uint32), lm_b.download().view(np.uint32))
        # the written object must be the context's own; unrelated contexts still refuse each other's objects
        with pytest.raises(native.IlluminantError):
            native.render_sphere_lights(b, lights, env, dfu, gb, field, AMBIENT, lm_a)
        with pytest.raises(native.IlluminantError):
            native.render_sphere_lights(stranger, lights, env, dfu, None, field, AMBIENT, lm_s)
        # particle lights through the borrowed field
        cs = 16
        eng = native.Engine(b, cs, scenes.randomness_table(7))
        sysm = native.System(eng); sysm.add_chunk()
        pos, vel, attr = scenes.make_particles(5, cs * cs, pos_lo=(0, 0, 2), pos_hi=(w, h, 30))
        rc = scenes.uniform(6, (cs * cs, 4), 0.3, 1.0).astype(np.float32); rc[:, :3] *= rc[:, 3:4]
        sysm.upload(0, abi.PLANE_POSITION, pos); sysm.upload(0, abi.PLANE_RENDER_COLOR, rc)
        params = lc.particle_light_params(3.0, 40.0, (1.0, 0.9, 0.8, 1.0), casts_shadows=True)
        native.render_particle_lights(b, sysm, params, env, dfu, gb, field, lm_b)
        eng_a = native.Engine(a, cs, scenes.randomness_table(7))
        sys_a = native.System(eng_a); sys_a.add_chunk()
        sys_a.upload(0, abi.PLANE_POSITION, pos); sys_a.upload(0, abi.PLANE_RENDER_COLOR, rc)
        native.render_particle_lights(a, sys_a, params, env, dfu, gb, field, lm_a)
        assert np.array_equal(lm_a.download().view(np.uint32), lm_b.download().view(np.uint32))
        for x in (sysm, eng, sys_a, eng_a):
            x.close()
    finally:
        for x in (lm_a, lm_b, lm_s, gb, field):
            x.close()
        for c in (stranger, b, a):
            c.close()


def heavy_scene():
    """cfg3's field (1536 x 2048 atlas, 33 slices) under a 1280 x 800 frame of 48 wide lights: generation and frame each keep the device
    busy for longer than the host needs to queue the next call, so that a missing dependency shows"""
    layout = scenes.DistanceFieldLayout(2048, 2048, 128.0, 32, 0.25)
    dfu = layout.uniforms(power=0.7, min_step_size=1.0, long_step_factor=0.5)
    w, h = 1280, 800
    lights = scenes.random_lights(31, 48, w, h, z=(8.0, 64.0), radius=24.0, ramp=(300.0, 700.0))
    return layout, dfu, lights, w, h


def heavy_obstructions(k):
    return scenes.obstruction_array(scenes.random_obstructions(500 + k, 180 + 7 * (k % 4), (2048, 2048), 12.0, 90.0, 80.0))


@pytest.mark.parametrize("sfmt", [abi.SDF_UNORM16, abi.SDF_FP16])
def test_frames_in_flight_over_a_field_that_changes_every_frame(ctx, sfmt):
    """Seven frames, the field regenerated in front of each (on the owner's stream; first with a decoy set, then with the frame's own), the
    frames alternating between the owner and its sibling, no synchronisation anywhere until the end: frame k must be the frame a single
    context renders after generating field k.  (Negative control: a build without the ordering, -DILM_EXP_NO_SHARED_ORDER, fails this.)"""
    layout, dfu, lights, w, h = heavy_scene()
    desc = scenes.render_desc(layout)
    triplets = list(range(0, layout.slice_count, 3))
    frames = 7
    sets = [heavy_obstructions(k) for k in range(frames)]
    decoys = [heavy_obstructions(100 + k) for k in range(frames)]
    env = scenes.environment()
    # references, one context, strictly sequential
    want = []
    ref_field = native.DistanceFieldTexture(ctx, None, sfmt, size=(layout.atlas_width, layout.atlas_height))
    ref_lm = native.Lightmap(ctx, w, h)
    for k in range(frames):
        ref_field.render_slices(desc, triplets, sets[k])
        native.render_sphere_lights(ctx, lights, env, dfu, None, ref_field, AMBIENT, ref_lm)
        want.append(ref_lm.download())
    ref_lm.close(); ref_field.close()
    assert not np.array_equal(want[0], want[1])
    # Twice, with a throw-away context created between the owner and its sibling the second time: the runtime deals streams to a handful of
    # hardware queues in creation order, and two streams that land on ONE queue are serialised by it -- the scenario would then pass
    # without any ordering of ours.  One of the two arrangements has the siblings on different queues.
    for pad in (0, 1):
        a = native.Context(0)
        extra = [native.Context(0) for _ in range(pad)]
        b = a.sibling()
        field = native.DistanceFieldTexture(a, None, sfmt, size=(layout.atlas_width, layout.atlas_height))
        ring = [[native.Lightmap(c, w, h) for _ in range((frames + 1) // 2)] for c in (a, b)]      # one lightmap per frame: every frame is checked
        try:
            for k in range(frames):
                field.render_slices(desc, triplets, decoys[k])                          # owner's stream; must wait for the sibling's frame k - 1
                field.render_slices(desc, triplets, sets[k])
                c = (a, b)[k & 1]
                native.render_sphere_lights(c, lights, env, dfu, None, field, AMBIENT, ring[k & 1][k // 2])   # must see field k: not the decoy, not k + 1
            a.sync(); b.sync()
            for k in range(frames):
                got = ring[k & 1][k // 2].download()
                assert np.array_equal(got.view(np.uint32), want[k].view(np.uint32)), "frame %d (context %s, arrangement %d) saw another field" % (k, "ab"[k & 1], pad)
        finally:
            for lms in ring:
                for lm in lms:
                    lm.close()
            field.close()
            b.close()
            for x in extra:
                x.close()
            a.close()


def test_frames_in_flight_under_a_gbuffer_that_changes_every_frame(ctx):
    """The same for the G-buffer: the owner re-renders it (ilm_gbuffer_render: ground plane + height volumes whose heights change with the
    frame; first a decoy, then the frame's own) in front of every frame, owner and sibling alternate, nothing synchronises until the end."""
    layout, dfu, lights, w, h = heavy_scene()
    frames = 6
    desc = scenes.gbuffer_render_desc(0.0)

    def volumes(k):
        r = scenes.uniform(900 + k, (40, 5))
        vols = []
        for v in range(40):
            cx, cy, rad = 60 + r[v, 0] * (w - 120), 60 + r[v, 1] * (h - 120), 20 + r[v, 2] * 60
            vols.append(([(cx - rad, cy - rad), (cx + rad, cy - rad), (cx + rad, cy + rad), (cx - rad, cy + rad)], 0.0, 4.0 + 40.0 * r[v, 3], True, True))
        return scenes.height_volume_arrays(vols)

    env = scenes.environment(gbuffer_size=(w, h))
    field_set = heavy_obstructions(0)
    want = []
    ref_field = native.DistanceFieldTexture(ctx, None, abi.SDF_UNORM16, size=(layout.atlas_width, layout.atlas_height))
    ref_field.render_slices(scenes.render_desc(layout), list(range(0, layout.slice_count, 3)), field_set)
    ref_gb = native.GBufferTexture(ctx, None, abi.GBUFFER_FLOAT4, size=(w, h))
    ref_lm = native.Lightmap(ctx, w, h)
    for k in range(frames):
        ref_gb.render(desc, *volumes(k))
        native.render_sphere_lights(ctx, lights, env, dfu, ref_gb, ref_field, AMBIENT, ref_lm)
        want.append(ref_lm.download())
    for x in (ref_lm, ref_gb, ref_field):
        x.close()
    assert not np.array_equal(want[0], want[1])
    for pad in (0, 1):
        a = native.Context(0)
        extra = [native.Context(0) for _ in range(pad)]
        b = a.sibling()
        field = native.DistanceFieldTexture(a, None, abi.SDF_UNORM16, size=(layout.atlas_width, layout.atlas_height))
        field.render_slices(scenes.render_desc(layout), list(range(0, layout.slice_count, 3)), field_set)
        gb = native.GBufferTexture(a, None, abi.GBUFFER_FLOAT4, size=(w, h))
        ring = [[native.Lightmap(c, w, h) for _ in range((frames + 1) // 2)] for c in (a, b)]
        try:
            for k in range(frames):
                gb.render(desc, *volumes(50 + k))
                gb.render(desc, *volumes(k))
                native.render_sphere_lights((a, b)[k & 1], lights, env, dfu, gb, field, AMBIENT, ring[k & 1][k // 2])
            a.sync(); b.sync()
            for k in range(frames):
                got = ring[k & 1][k // 2].download()
                assert np.array_equal(got.view(np.uint32), want[k].view(np.uint32)), "frame %d (context %s, arrangement %d) saw another G-buffer" % (k, "ab"[k & 1], pad)
        finally:
            for lms in ring:
                for lm in lms:
                    lm.close()
            gb.close(); field.close()
            b.close()
            for x in extra:
                x.close()
            a.close()
