"""Parity of the HIP sphere-light / SDF cone-trace pass (through the C ABI) against the CPU oracle."""
import numpy as np
import pytest

from illuminant_amd import abi, native, scenes
from tests.util import assert_close

pytestmark = pytest.mark.gpu


def small_scene(fmt=abi.SDF_UNORM16, n_lights=12, width=160, height=112, seed=5):
    layout = scenes.DistanceFieldLayout(256, 192, 96.0, 12, 0.5, 128)
    obstacles = scenes.random_obstacles(seed, 14, (256, 192), size_lo=8.0, size_hi=30.0, z_hi=40.0)
    atlas = scenes.build_sdf_atlas(layout, obstacles, fmt=fmt)
    dfu = layout.uniforms(max_cone_radius=24.0, power=0.7, step_limit=64, min_step_size=1.0, long_step_factor=0.5)
    lights = scenes.random_lights(seed + 1, n_lights, width, height, z=(8.0, 48.0), radius=10.0, ramp=(40.0, 120.0))
    return layout, atlas, dfu, lights, width, height


def render_both(ctx, oracle, lights, env, dfu, gbuf_arr, gfmt, atlas, sfmt, ambient, width, height, lm_fmt=abi.LIGHTMAP_FLOAT4,
                row_begin=0, row_end=None, want_stats=True):
    sdf = native.DistanceFieldTexture(ctx, atlas, sfmt) if atlas is not None else None
    gb = native.GBufferTexture(ctx, gbuf_arr, gfmt) if gbuf_arr is not None else None
    lm = native.Lightmap(ctx, width, height, lm_fmt)
    stats = native.render_sphere_lights(ctx, lights, env, dfu, gb, sdf, ambient, lm, row_begin, row_end, want_stats=want_stats)
    got = lm.download()
    otex = oracle.make_texture(atlas, sfmt) if atlas is not None else None
    ogb = oracle.make_texture(gbuf_arr, gfmt) if gbuf_arr is not None else None
    want, ostats = oracle.render_sphere_lights(lights, env, dfu, ogb, otex, ambient, width, height,
                                               row_begin, height if row_end is None else row_end, want_stats=True)
    lm.close()
    if gb is not None:
        gb.close()
    if sdf is not None:
        sdf.close()
    return got, want, stats, ostats


@pytest.mark.parametrize("sfmt", [abi.SDF_UNORM16, abi.SDF_FP16])
def test_ground_plane_render_matches_oracle(ctx, oracle, sfmt):
    layout, atlas, dfu, lights, w, h = small_scene(sfmt)
    env = scenes.environment()
    got, want, stats, ostats = render_both(ctx, oracle, lights, env, dfu, None, 0, atlas, sfmt, (0.05, 0.06, 0.07, 1.0), w, h)
    # sample counts are integers: the trace visited exactly the same steps
    assert (stats.SdfSamples, stats.PixelLightPairs, stats.TracedPairs) == (ostats.SdfSamples, ostats.PixelLightPairs, ostats.TracedPairs)
    assert stats.SdfSamples > 10 * w * h
    assert_close(got, want, "lightmap")
    # the scene must contain both lit and shadowed pixels
    assert (want[..., 3] > 1.5).mean() > 0.5


def test_gbuffer_render_matches_oracle(ctx, oracle):
    layout, atlas, dfu, lights, w, h = small_scene()
    # a G-buffer with a tilted bump, raised ground, an unshadowed band and a fullbright band
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    nx = 0.3 * np.sin(xx / 9.0); ny = 0.3 * np.cos(yy / 7.0)
    nz = np.sqrt(np.maximum(1.0 - nx * nx - ny * ny, 0.0))
    normal = np.stack([nx, ny, nz], axis=-1)
    z = 6.0 + 5.0 * np.sin(xx / 17.0) * np.cos(yy / 13.0)
    g = scenes.encode_gbuffer(normal, 0.0, z)
    g[10:20] = scenes.encode_gbuffer(normal[10:20], 0.0, z[10:20], enable_shadows=False)
    g[30:34] = scenes.encode_gbuffer(normal[30:34], 0.0, z[30:34], fullbright=True)
    g[40:44, :, :2] = 0.0   # zero normal: directional occlusion disabled
    # lights with AO, specular, both falloff modes and a shadow filter
    for i in range(len(lights)):
        lights[i].MoreLightProperties.x = 12.0 if i % 2 else 0.0
        lights[i].MoreLightProperties.w = 0.6
        lights[i].Color2 = abi.f4(0.3, 0.2, 0.1, 8.0) if i % 3 == 0 else abi.f4(0, 0, 0, 1)
        lights[i].LightProperties.z = float(i % 3)
        lights[i].EvenMoreLightProperties.x = float((i % 4) - 1)
    env = scenes.environment(gbuffer_size=(w, h), light_occlusion=40.0)
    for gfmt, garr in ((abi.GBUFFER_FLOAT4, g), (abi.GBUFFER_HALF4, g.astype(np.float16).view(np.uint16))):
        got, want, stats, ostats = render_both(ctx, oracle, lights, env, dfu, garr, gfmt, atlas, abi.SDF_UNORM16,
                                               (0.0, 0.0, 0.0, 0.0), w, h)
        assert (stats.SdfSamples, stats.PixelLightPairs, stats.TracedPairs) == (ostats.SdfSamples, ostats.PixelLightPairs, ostats.TracedPairs)
        assert_close(got, want, "lightmap gbuffer fmt %d" % gfmt)
        assert not got[30:34].any()            # fullbright pixels are discarded for every light


def test_no_distance_field_and_25d_footprint(ctx, oracle):
    w, h = 96, 80
    lights = scenes.random_lights(9, 5, w, h, z=(4.0, 20.0), radius=6.0, ramp=(20.0, 40.0), falloff_y=0.5, have_distance_field=False)
    env = scenes.environment(z_to_y=1.5)
    dfu = abi.DistanceFieldUniforms()
    dfu.StepAndMisc2 = abi.f4(64, 3, 1, 1)
    got, want, stats, ostats = render_both(ctx, oracle, lights, env, dfu, None, 0, None, 0, (0.1, 0.1, 0.1, 1.0), w, h)
    assert stats.SdfSamples == 0 == ostats.SdfSamples
    assert stats.PixelLightPairs == ostats.PixelLightPairs
    assert_close(got, want, "lightmap without distance field")
    # FalloffYFactor < 1 makes the lit ellipse taller than the raster quad: the footprint clips it
    assert (want[..., 3] == 1.0).any() and (want[..., 3] > 1.0).any()


def test_light_at_pixel_without_field_is_fully_lit(ctx, oracle):
    """KAT 5 of SURVEY 8c: inside the light radius opacity is 1 (LightCommon.fxh:208-209)."""
    w = h = 32
    lights = (abi.LightVertex * 1)(scenes.sphere_light((16.0, 16.0, 0.0), 8.0, 4.0, color=(0.25, 0.5, 1.0, 0.5),
                                                       have_distance_field=False))
    env = scenes.environment()
    dfu = abi.DistanceFieldUniforms()
    got, want, _, _ = render_both(ctx, oracle, lights, env, dfu, None, 0, None, 0, (0, 0, 0, 0), w, h)
    assert np.allclose(got[16, 16], (0.125, 0.25, 0.5, 1.0), rtol=0, atol=1e-7)
    assert_close(got, want, "single light")


def test_strips_and_output_formats(ctx, oracle):
    layout, atlas, dfu, lights, w, h = small_scene(n_lights=6)
    env = scenes.environment()
    amb = (0.02, 0.02, 0.02, 1.0)
    sdf = native.DistanceFieldTexture(ctx, atlas)
    full = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
    native.render_sphere_lights(ctx, lights, env, dfu, None, sdf, amb, full)
    ref = full.download()
    # row strips (the multi-GPU screen split) tile the frame exactly
    strips = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
    for (a, b) in ((0, 37), (37, 64), (64, h)):
        native.render_sphere_lights(ctx, lights, env, dfu, None, sdf, amb, strips, a, b)
    assert np.array_equal(strips.download(), ref)
    half = native.Lightmap(ctx, w, h, abi.LIGHTMAP_HALF4)
    native.render_sphere_lights(ctx, lights, env, dfu, None, sdf, amb, half)
    assert np.array_equal(half.download(), ref.astype(np.float16))
    rgba = native.Lightmap(ctx, w, h, abi.LIGHTMAP_RGBA8)
    native.render_sphere_lights(ctx, lights, env, dfu, None, sdf, amb, rgba)
    assert np.array_equal(rgba.download(), np.rint(np.clip(ref, 0, 1) * 255.0).astype(np.uint8))
    # zero lights: the ambient clear (LightingRenderer.cs:1013-1024)
    native.render_sphere_lights(ctx, None, env, dfu, None, sdf, amb, full)
    assert np.array_equal(full.download(), np.broadcast_to(np.asarray(amb, np.float32), (h, w, 4)))
    for x in (full, strips, half, rgba, sdf):
        x.close()


def test_many_lights_exceed_one_tile_list(ctx, oracle):
    """More lights than the LDS tile list holds (1024) are processed in batches, in light order."""
    w, h = 48, 32
    n = 1300
    lights = scenes.random_lights(21, n, w, h, z=(4.0, 12.0), radius=2.0, ramp=(6.0, 14.0), have_distance_field=False)
    env = scenes.environment()
    dfu = abi.DistanceFieldUniforms()
    got, want, stats, ostats = render_both(ctx, oracle, lights, env, dfu, None, 0, None, 0, (0, 0, 0, 0), w, h)
    assert stats.PixelLightPairs == ostats.PixelLightPairs
    assert_close(got, want, "1300 lights")


def test_fp16_per_light_blend_model_matches_oracle(ctx, oracle):
    """ILM_BLEND_FP16_PER_LIGHT: the reference's HalfVector4 lightmap, rounded by the ROP after every light (LightingRenderer.cs:476-479).
    The kernel's model equals the oracle's bit for bit in every lightmap format; SDF sample / pair / trace counts do not depend on the
    blend; the fp32-accumulate frame (the parity model) differs from it by fp16 rounding only."""
    layout, atlas, dfu, lights, w, h = small_scene(n_lights=12)
    env = scenes.environment()
    amb = (0.0213, 0.0377, 0.0591, 1.0)                       # not representable in fp16: the clear colour is rounded too
    sdf = native.DistanceFieldTexture(ctx, atlas)
    tex = oracle.make_texture(atlas, abi.SDF_UNORM16)
    want, wstats = oracle.render_sphere_lights(lights, env, dfu, None, tex, amb, w, h, want_stats=True, blend_fp16=True)
    want32, _ = oracle.render_sphere_lights(lights, env, dfu, None, tex, amb, w, h)
    assert np.array_equal(want, want.astype(np.float16).astype(np.float32)), "every value of the fp16 model is an fp16 value"
    ctx.set_lightmap_blend(True)
    try:
        for fmt in (abi.LIGHTMAP_FLOAT4, abi.LIGHTMAP_HALF4):
            lm = native.Lightmap(ctx, w, h, fmt)
            stats = native.render_sphere_lights(ctx, lights, env, dfu, None, sdf, amb, lm, want_stats=True)
            got = lm.download().astype(np.float32)
            # the per-light contributions differ from the oracle's by the parity tolerance BEFORE they are rounded: a contribution within
            # that distance of an fp16 rounding boundary may land on the neighbouring fp16 value -- one fp16 ulp (2^-11 relative) at most
            diff = np.abs(got - want)
            assert (diff <= np.abs(want) * 2.0 ** -10 + 1e-7).all()
            assert (diff > 0).mean() < 0.02, "more than 2 %% of the texels differ from the oracle's fp16 model"
            assert (stats.SdfSamples, stats.PixelLightPairs, stats.TracedPairs) == (wstats.SdfSamples, wstats.PixelLightPairs, wstats.TracedPairs)
            lm.close()
    finally:
        ctx.set_lightmap_blend(False)
    # back to the parity model, and how far the two models are apart: fp16's 2^-11 per partial sum, accumulated over <= 12 lights
    lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
    native.render_sphere_lights(ctx, lights, env, dfu, None, sdf, amb, lm)
    assert_close(lm.download(), want32, "fp32-accumulate frame after switching back")
    rel = np.abs(want - want32) / np.maximum(np.abs(want32), 1e-3)
    assert 1e-5 < rel.max() < 13 * 2.0 ** -11
    lm.close(); sdf.close()


@pytest.mark.parametrize("sfmt", [abi.SDF_UNORM16, abi.SDF_FP16])
def test_the_trace_follows_the_atlas_when_it_changes(ctx, oracle, sfmt):
    """The cone trace reads the field through a cell array built from the atlas (hlsl_math.hpp, SdfView::cells).  Whatever writes the atlas
    -- ilm_sdf_upload, ilm_sdf_render_slices, a caller holding the device pointer -- must reach the next frame: the frame after each
    change equals the oracle's frame of the NEW field, with its exact SDF sample counts."""
    layout, atlas_a, dfu, lights, w, h = small_scene(sfmt)
    atlas_b = scenes.build_sdf_atlas(layout, scenes.random_obstacles(77, 9, (256, 192), size_lo=12.0, size_hi=40.0, z_hi=60.0), fmt=sfmt)
    assert not np.array_equal(atlas_a, atlas_b)
    env = scenes.environment()
    amb = (0.03, 0.03, 0.03, 1.0)
    sdf = native.DistanceFieldTexture(ctx, atlas_a, sfmt)
    lm = native.Lightmap(ctx, w, h)

    def check(atlas, what):
        stats = native.render_sphere_lights(ctx, lights, env, dfu, None, sdf, amb, lm, want_stats=True)
        want, ostats = oracle.render_sphere_lights(lights, env, dfu, None, oracle.make_texture(atlas, sfmt), amb, w, h, want_stats=True)
        assert (stats.SdfSamples, stats.TracedPairs) == (ostats.SdfSamples, ostats.TracedPairs), what
        assert_close(lm.download(), want, what)
        return stats.SdfSamples

    s_a = check(atlas_a, "first field")
    sdf.upload(atlas_b)
    s_b = check(atlas_b, "after ilm_sdf_upload")
    assert s_a != s_b                                   # the two fields really trace differently
    check(atlas_b, "unchanged field, cached cells")
    # a field generated on the device, then regenerated with other obstructions
    gen = native.DistanceFieldTexture(ctx, None, sfmt, size=(layout.atlas_width, layout.atlas_height))
    desc = scenes.render_desc(layout)
    triplets = list(range(0, layout.slice_count, 3))
    for seed in (3, 4):
        obs = scenes.obstruction_array(scenes.random_obstructions(seed, 8, (256, 192), 8.0, 30.0, 40.0))
        gen.render_slices(desc, triplets, obs)
        generated = gen.download()
        stats = native.render_sphere_lights(ctx, lights, env, dfu, None, gen, amb, lm, want_stats=True)
        want, ostats = oracle.render_sphere_lights(lights, env, dfu, None, oracle.make_texture(generated, sfmt), amb, w, h, want_stats=True)
        assert stats.SdfSamples == ostats.SdfSamples, "after ilm_sdf_render_slices (seed %d)" % seed
        assert_close(lm.download(), want, "after ilm_sdf_render_slices (seed %d)" % seed)
    # ONE slice triplet regenerated (the reference's cadence: MaximumFieldUpdatesPerFrame = 1, LightingRenderer.Configuration.cs:91): the
    # frame follows, and only the cells of that triplet and of the slice below it were re-derived (a cell of slice v holds the pairs (v, v + 1))
    n_slices = layout.slice_count
    assert gen.trace_info().TableSlices == n_slices and gen.trace_info().CellBytes > 0
    obs = scenes.obstruction_array(scenes.random_obstructions(5, 8, (256, 192), 8.0, 30.0, 40.0))
    for first in (0, 3, n_slices - 3):
        before = gen.trace_info().CellSlicesRebuilt
        gen.render_slices(desc, [first], obs)
        generated = gen.download()
        stats = native.render_sphere_lights(ctx, lights, env, dfu, None, gen, amb, lm, want_stats=True)
        want, ostats = oracle.render_sphere_lights(lights, env, dfu, None, oracle.make_texture(generated, sfmt), amb, w, h, want_stats=True)
        assert stats.SdfSamples == ostats.SdfSamples, "after one triplet (%d)" % first
        assert_close(lm.download(), want, "after one triplet (%d)" % first)
        info = gen.trace_info()
        assert info.LastRebuiltSlices == (3 if first == 0 else 4) and info.CellSlicesRebuilt - before == info.LastRebuiltSlices
        assert info.LastRebuiltSlices * 33 <= 4 * max(n_slices, 33) or n_slices < 33
    native.render_sphere_lights(ctx, lights, env, dfu, None, gen, amb, lm)
    assert gen.trace_info().LastRebuiltSlices == 0                     # nothing changed: the cells are kept
    # a field whose device pointer was handed out is re-read before every frame ...
    assert sdf.device_ptr() != 0
    sdf.upload(atlas_a)
    assert check(atlas_a, "after the pointer escaped") == s_a
    assert sdf.trace_info().RebuiltEveryFrame == 1 and sdf.trace_info().LastRebuiltSlices == n_slices
    check(atlas_a, "escaped, unchanged")
    assert sdf.trace_info().LastRebuiltSlices == n_slices
    # ... until the caller reports its writes itself (ilm_sdf_mark_dirty)
    sdf.mark_dirty(3, 3)
    assert check(atlas_a, "after ilm_sdf_mark_dirty") == s_a
    assert sdf.trace_info().RebuiltEveryFrame == 0 and sdf.trace_info().LastRebuiltSlices == 4
    check(atlas_a, "reported, unchanged")
    assert sdf.trace_info().LastRebuiltSlices == 0
    gen.close(); lm.close(); sdf.close()


@pytest.mark.parametrize("width,height,rows", [(96, 96, None), (97, 95, None), (112, 208, None), (193, 97, None), (16, 16, None), (5, 3, None),
                                                (300, 220, (32, 176)), (208, 400, (96, 112)), (1, 1, None)])
def test_every_pixel_of_any_frame_size_is_rendered_once(ctx, oracle, width, height, rows):
    """The light pass deals its 16 x 16 tiles to the XCDs in groups of 6 x 6 (lighting.hip, tile_map 4): frames whose tile counts are
    multiples of the group edge, one more, one less, smaller than a group, and strips that begin and end inside a group.  A tile rendered
    twice would be harmless; one never rendered keeps the poison the lightmap is filled with."""
    lights = scenes.random_lights(width * 1000 + height, 5, width, height, z=(8.0, 48.0), radius=10.0, ramp=(60.0, 200.0))
    lights = (abi.LightVertex * len(lights))(*lights)
    env = scenes.environment()
    dfu = scenes.DistanceFieldLayout(64, 64, 32.0, 3, 1.0, 64).uniforms()
    row_begin, row_end = rows if rows is not None else (0, height)
    lm = native.Lightmap(ctx, width, height)
    lm.upload(np.full((height, width, 4), -7.0, np.float32))
    native.render_sphere_lights(ctx, lights, env, dfu, None, None, (0.1, 0.2, 0.3, 1.0), lm, row_begin, row_end)
    got = lm.download()
    want, _ = oracle.render_sphere_lights(lights, env, dfu, None, None, (0.1, 0.2, 0.3, 1.0), width, height, row_begin, row_end)
    assert np.all(got[row_begin:row_end, :, 3] >= 1.0), "a pixel of the strip was not rendered"
    assert_close(got[row_begin:row_end], want[row_begin:row_end], "lightmap %dx%d rows %d..%d" % (width, height, row_begin, row_end))
    assert np.all(got[:row_begin] == -7.0) and np.all(got[row_end:] == -7.0), "rows outside the strip were written"
    lm.close()


@pytest.mark.parametrize("sfmt", [abi.SDF_UNORM16, abi.SDF_FP16])
def test_light_split_keeps_every_bit(ctx, oracle, sfmt):
    """ilm_ctx_set_light_split: K workgroups per tile share the 8 parts of the light list; the lightmap's bits depend on the light list and
    the pixel alone -- not on K, not on where a strip starts (tiles are anchored at row_begin), not on the lightmap's format rounding
    order -- and the SDF-sample / pair / trace counts stay the oracle's.  Also a second light group added onto the first (ambient NULL:
    the base value is the lightmap's contents) and a list shorter than the parts."""
    layout, atlas, dfu, lights, w, h = small_scene(sfmt, n_lights=37, width=203, height=150)
    env = scenes.environment()
    ambient = (0.05, 0.06, 0.07, 1.0)
    sdf = native.DistanceFieldTexture(ctx, atlas, sfmt)
    want, ostats = oracle.render_sphere_lights(lights, env, dfu, None, oracle.make_texture(atlas, sfmt), ambient, w, h, want_stats=True)
    few = (abi.LightVertex * 3)(*[lights[i] for i in (5, 6, 7)])

    def frame(split, strips, fmt=abi.LIGHTMAP_FLOAT4, second_group=False):
        ctx.set_light_split(split)
        lm = native.Lightmap(ctx, w, h, fmt)
        counts = [0, 0, 0]
        for (b, e) in strips:
            st = native.render_sphere_lights(ctx, lights, env, dfu, None, sdf, ambient, lm, b, e, want_stats=True)
            counts = [counts[0] + st.SdfSamples, counts[1] + st.PixelLightPairs, counts[2] + st.TracedPairs]
            if second_group:
                native.render_sphere_lights(ctx, few, env, dfu, None, sdf, None, lm, b, e)
        out = lm.download()
        lm.close()
        return out, tuple(counts)

    try:
        whole, counts = frame(1, [(0, h)])
        assert counts == (ostats.SdfSamples, ostats.PixelLightPairs, ostats.TracedPairs)
        assert_close(whole, want, "one workgroup per tile vs the oracle")
        uneven = [(0, 7), (7, 100), (100, 101), (101, h)]
        half_whole, _ = frame(1, [(0, h)], abi.LIGHTMAP_HALF4)
        two_whole, _ = frame(1, [(0, h)], second_group=True)
        assert not np.array_equal(two_whole, whole)
        for split in (2, 4, 8, 0):
            for strips in ([(0, h)], uneven):
                got, c = frame(split, strips)
                assert np.array_equal(got, whole), "split %d, strips %s" % (split, strips)
                assert c == counts
            assert np.array_equal(frame(split, uneven, abi.LIGHTMAP_HALF4)[0], half_whole)
            assert np.array_equal(frame(split, uneven, second_group=True)[0], two_whole)
        # the fp16-per-light model is one chain of roundings: the setting is ignored there
        ctx.set_lightmap_blend(True)
        a, _ = frame(1, [(0, h)], abi.LIGHTMAP_HALF4)
        b, _ = frame(8, uneven, abi.LIGHTMAP_HALF4)
        assert np.array_equal(a, b)
    finally:
        ctx.set_lightmap_blend(False)
        ctx.set_light_split(0)
        sdf.close()
    with pytest.raises(native.IlluminantError):
        ctx.set_light_split(3)


@pytest.mark.parametrize("sfmt", [abi.SDF_UNORM16, abi.SDF_FP16])
def test_circle_cull_keeps_every_bit(ctx, oracle, sfmt):
    """r05: waves whose shaded points lie outside a light's circle (radius + max(ramp, 1), y scaled by falloffY) skip the light, entries no
    wave of a tile needs are not listed (lighting.hip, culled_waves).  The instrumented variant does not cull (its pair count is the raster
    footprint's): the plain frame must equal it bit for bit -- small lights on a frame of many tiles, a G-buffer whose relativeY and z
    move the shaded points off their pixels under a 2.5D footprint, falloffY != 1, the three falloff modes, ramps below 1 and not
    positive, a light far outside the frame -- and the instrumented counts stay the oracle's.  Particle lights (the wide-binning
    instantiation) the same way."""
    from tests import lights_common as lc
    layout, atlas, dfu, _, w, h = small_scene(sfmt, width=211, height=157)
    sdf = native.DistanceFieldTexture(ctx, atlas, sfmt)
    otex = oracle.make_texture(atlas, sfmt)
    ambient = (0.05, 0.06, 0.07, 1.0)
    L = []
    xs, ys = scenes.uniform(61, (40,), -20, w + 20), scenes.uniform(62, (40,), -20, h + 20)
    for i in range(40):
        kw = dict(ramp_mode=i % 3, falloff_y=(1.0, 0.6, 2.5, -1.5)[i % 4], casts_shadows=(i % 5 != 0))
        ramp = (22.0, 9.0, 0.4, 35.0, 0.0, -3.0, 14.0)[i % 7]
        L.append(scenes.sphere_light((xs[i], ys[i], 4.0 + i), (3.0, 0.0, 7.5)[i % 3], ramp, color=(0.9, 0.5 + 0.01 * i, 0.3, 1.0), **kw))
    L.append(scenes.sphere_light((-5000.0, 80.0, 10.0), 4.0, 30.0))
    lights = (abi.LightVertex * len(L))(*L)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    nx = 0.3 * np.sin(xx / 9.0); ny = 0.3 * np.cos(yy / 7.0)
    normal = np.stack([nx, ny, np.sqrt(np.maximum(1.0 - nx * nx - ny * ny, 0.0))], axis=-1)
    z = 6.0 + 5.0 * np.sin(xx / 17.0) * np.cos(yy / 13.0)
    g = scenes.encode_gbuffer(normal, 9.0 * np.sin(xx / 23.0), z)          # relativeY moves the shaded point up to 9 px off its pixel
    gb = native.GBufferTexture(ctx, g, abi.GBUFFER_FLOAT4)
    ogb = oracle.make_texture(g, abi.GBUFFER_FLOAT4)
    try:
        for env, gbuf, ogbuf in ((scenes.environment(), None, None), (scenes.environment(gbuffer_size=(w, h), z_to_y=0.4), gb, ogb)):
            frames = []
            for want_stats in (True, False):
                lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
                st = native.render_sphere_lights(ctx, lights, env, dfu, gbuf, sdf, ambient, lm, want_stats=want_stats)
                frames.append(lm.download())
                lm.close()
                if want_stats:
                    _, ost = oracle.render_sphere_lights(lights, env, dfu, ogbuf, otex, ambient, w, h, want_stats=True)
                    assert (st.SdfSamples, st.PixelLightPairs, st.TracedPairs) == (ost.SdfSamples, ost.PixelLightPairs, ost.TracedPairs)
            assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32)), "the culled frame differs from the instrumented one"
            assert (frames[0][..., 3] > 1.5).mean() > 0.05 and (frames[0][..., 3] == 1.0).mean() > 0.2       # lit and unlit pixels both
        # particle lights: 600 lights of reach 3 + 18 on the same frame
        cs = 32
        n = cs * cs
        pos, vel, attr = scenes.make_particles(91, n, pos_lo=(0, 0, 2), pos_hi=(w, h, 30), dead_fraction=0.4)
        rc = scenes.uniform(92, (n, 4), 0.2, 1.0).astype(np.float32)
        rc[:, :3] *= rc[:, 3:4]
        eng = native.Engine(ctx, cs, scenes.randomness_table(7))
        sysm = native.System(eng)
        sysm.add_chunk()
        sysm.upload(0, abi.PLANE_POSITION, pos); sysm.upload(0, abi.PLANE_RENDER_COLOR, rc)
        for fy in (1.0, 0.7):
            params = lc.particle_light_params(3.0, 18.0, (1.0, 0.9, 0.8, 1.0), casts_shadows=True, falloff_y=fy)
            frames = []
            for want_stats in (True, False):
                lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
                native.render_sphere_lights(ctx, None, scenes.environment(), dfu, None, sdf, ambient, lm)
                native.render_particle_lights(ctx, sysm, params, scenes.environment(), dfu, None, sdf, lm, want_stats=want_stats)
                frames.append(lm.download())
                lm.close()
            assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32)), "particle lights: the culled frame differs (falloffY %g)" % fy
            assert (frames[0][..., 3] > 2.5).any()
        sysm.close(); eng.close()
    finally:
        gb.close(); sdf.close()


@pytest.mark.gpu
def test_crowded_tiles_end_their_batches_early(ctx, oracle):
    """The wide binning's batches (r06): up to 4 096 lights each, ended early when another binning round of 256 lights might not fit the
    tile's 1 024-entry list (lighting.hip, BIG).  Two chunks of 64^2 slots, ~4 900 live particle lights: the first chunk's crowd around
    one spot -- the tiles there list well over a thousand of them, so their batches end after a few rounds while the other tiles' run
    to 4 096 -- the second's are spread over the frame.  The plain frame must equal the instrumented one (batches of 1 024, no cull)
    bit for bit, and the instrumented counts are the oracle's."""
    from tests import lights_common as lc
    layout, atlas, dfu, _, w, h = small_scene(abi.SDF_UNORM16, width=200, height=144)
    sdf = native.DistanceFieldTexture(ctx, atlas, abi.SDF_UNORM16)
    otex = oracle.make_texture(atlas, abi.SDF_UNORM16)
    ambient = (0.05, 0.06, 0.07, 1.0)
    cs = 64
    n = cs * cs
    chunks = []
    for c, (lo, hi) in enumerate((((70, 50, 2), (120, 90, 30)), ((0, 0, 2), (w, h, 30)))):
        pos, vel, attr = scenes.make_particles(191 + c, n, pos_lo=lo, pos_hi=hi, dead_fraction=0.4)
        rc = scenes.uniform(192 + c, (n, 4), 0.2, 1.0).astype(np.float32)
        rc[:, :3] *= rc[:, 3:4]
        chunks.append([pos, vel, attr, rc, np.zeros((n, 4), np.float32)])
    eng = native.Engine(ctx, cs, scenes.randomness_table(7))
    sysm = native.System(eng)
    for c, planes in enumerate(chunks):
        sysm.add_chunk()
        sysm.upload(c, abi.PLANE_POSITION, planes[0]); sysm.upload(c, abi.PLANE_RENDER_COLOR, planes[3])
    params = lc.particle_light_params(2.0, 9.0, (1.0, 0.9, 0.8, 0.05), casts_shadows=True)
    env = scenes.environment()
    frames = []
    try:
        for want_stats in (True, False):
            lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
            native.render_sphere_lights(ctx, None, env, dfu, None, sdf, ambient, lm)
            st = native.render_particle_lights(ctx, sysm, params, env, dfu, None, sdf, lm, want_stats=want_stats)
            frames.append(lm.download())
            lm.close()
            if want_stats:
                want = np.empty((h, w, 4), np.float32)
                want[...] = np.asarray(ambient, np.float32)
                ost = oracle.render_particle_lights(chunks, [n, n], params, env, dfu, None, otex, want, want_stats=True)
                assert (st.SdfSamples, st.PixelLightPairs, st.TracedPairs) == (ost.SdfSamples, ost.PixelLightPairs, ost.TracedPairs)
        assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32)), "the frame of early-ended batches differs from the instrumented one"
        # the tile of pixels (96..111, 64..79) lists every live light whose square footprint (half edge radius + ramp + 1 = 12) touches it:
        # far more than one list's worth, so its first batch cannot run to 4 096 lights
        p0 = chunks[0][0]
        listed = (p0[:, 3] > 0) & (p0[:, 0] > 96 - 12) & (p0[:, 0] < 112 + 12) & (p0[:, 1] > 64 - 12) & (p0[:, 1] < 80 + 12)
        assert int(listed.sum()) > 1300, int(listed.sum())
        assert frames[0][..., 3].max() > 60.0, frames[0][..., 3].max()
        assert_close(frames[0], want, "lightmap")
    finally:
        sysm.close(); eng.close(); sdf.close()
