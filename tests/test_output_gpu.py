"""Parity of the output-side kernels (ordered read-back compaction, lightmap resolve -- SURVEY 8f-4) against the CPU oracle."""
import numpy as np
import pytest

from illuminant_amd import abi, native, scenes
from tests import output_common as oc
from tests.util import assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("index", range(len(oc.load_cases())))
def test_closed_form_case(ctx, index):
    oc.check_case(oc.load_cases()[index], oc.GpuBackend(ctx))


def test_readback_matches_oracle_record_for_record(ctx, oracle):
    """3 chunks of 64^2 slots with dead slots, a 4 x 4 frame sheet, velocity-driven frames, sorted: integer fields (colour bytes),
    record count and order are exact; float fields within 1e-6."""
    cs, n_chunks = 64, 3
    n = cs * cs
    eng = native.Engine(ctx, cs, scenes.randomness_table(7))
    sysm = native.System(eng)
    chunks = []
    for c in range(n_chunks):
        sysm.add_chunk()
        pos, vel, attr = scenes.make_particles(60 + c, n, pos_lo=(0, 0, 0), pos_hi=(1920, 1080, 32), dead_fraction=0.4)
        rc = scenes.uniform(70 + c, (n, 4), 0.0, 1.2).astype(np.float32)
        rd = np.stack([scenes.uniform(80 + c, (n,), 0.2, 3.0), scenes.uniform(81 + c, (n,), -10.0, 20.0),
                       scenes.uniform(82 + c, (n,), 0.0, 90.0), np.floor(scenes.uniform(83 + c, (n,), 0.0, 6.0))], axis=1).astype(np.float32)
        sysm.upload(c, abi.PLANE_POSITION, pos); sysm.upload(c, abi.PLANE_RENDER_COLOR, rc); sysm.upload(c, abi.PLANE_RENDER_DATA, rd)
        chunks.append([pos, vel, attr, rc, rd])
    params = oc.readback_params((2.0, 3.0), (0.0, 0.0, 0.25, 0.25), (1.7, -0.6), 0.35, True, True, True, True)
    elems = [n, 40 * cs, 17 * cs]        # ceil(TotalSpawned / ChunkSize) * ChunkSize per chunk
    got, gn = sysm.readback(params, element_counts=elems)
    want, wn = oracle.fill_readback_result(chunks, params, element_counts=elems)
    assert gn == wn and 0.3 * sum(elems) < gn < 0.9 * sum(elems)
    g, w = np.frombuffer(got, dtype=np.uint8).reshape(-1, 48)[:gn], np.frombuffer(want, dtype=np.uint8).reshape(-1, 48)[:wn]
    assert np.array_equal(g[:, 40:44], w[:, 40:44])                    # MultiplyColor bytes
    gf, wf = g[:, :40].copy().view(np.float32), w[:, :40].copy().view(np.float32)
    assert np.array_equal(gf[:, :2], wf[:, :2])                        # positions are copies: bit-equal, which also pins the order
    assert_close(gf, wf, "draw call floats", rtol=1e-6, atol=1e-6)
    # a capacity smaller than the live count: the total is still reported, the prefix is returned
    got2, gn2 = sysm.readback(params, element_counts=elems, capacity=100)
    assert gn2 == gn
    assert np.array_equal(np.frombuffer(got2, dtype=np.uint8).reshape(-1, 48)[:100], g[:100])
    # the zero-copy form: the same records, viewed in the context's pinned buffer
    view = sysm.readback_view(params, element_counts=elems)
    assert view.shape == (gn, 12) and np.array_equal(view.view(np.uint8).reshape(-1, 48), g)
    # nothing examined -> nothing returned
    assert sysm.readback_view(params, element_counts=[0, 0, 0]).shape[0] == 0
    sysm.close(); eng.close()


@pytest.mark.parametrize("mode", [abi.HDR_NONE, abi.HDR_GAMMA_COMPRESS, abi.HDR_TONE_MAP])
@pytest.mark.parametrize("fmt", [abi.LIGHTMAP_FLOAT4, abi.LIGHTMAP_HALF4])
def test_resolve_matches_oracle_on_a_lit_frame(ctx, oracle, mode, fmt):
    w, h = 160, 112
    layout = scenes.DistanceFieldLayout(256, 192, 96.0, 12, 0.5, 128)
    atlas = scenes.build_sdf_atlas(layout, scenes.random_obstacles(5, 14, (256, 192), size_lo=8.0, size_hi=30.0, z_hi=40.0))
    dfu = layout.uniforms(max_cone_radius=24.0, power=0.7, step_limit=64, min_step_size=1.0, long_step_factor=0.5)
    lights = scenes.random_lights(6, 12, w, h, z=(8.0, 48.0), radius=10.0, ramp=(40.0, 120.0))
    env = scenes.environment()
    sdf = native.DistanceFieldTexture(ctx, atlas)
    src = native.Lightmap(ctx, w, h, fmt)
    native.render_sphere_lights(ctx, lights, env, dfu, None, sdf, (0.05, 0.06, 0.07, 1.0), src)
    lit = src.download().astype(np.float32)            # what the resolve reads (fp16-rounded for the HalfVector4 lightmap)
    hdr = oc.hdr_configuration(mode, 0.5, 0.02, 1.3, 0.9, 0.5, 0.8, 3.0, 2.5)
    dst = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
    native.resolve_lighting(src, dst, hdr, 8, h - 8)
    got = dst.download()
    want = oracle.resolve_lighting(np.ascontiguousarray(lit), hdr, 8, h - 8)
    assert_close(got[8:h - 8], want[8:h - 8], "resolved frame")
    assert not got[:8].any() and not got[h - 8:].any()             # rows outside the strip untouched
    # the back-buffer case: RGBA8 destination = round(saturate(x) * 255)
    dst8 = native.Lightmap(ctx, w, h, abi.LIGHTMAP_RGBA8)
    native.resolve_lighting(src, dst8, hdr)
    full = oracle.resolve_lighting(np.ascontiguousarray(lit), hdr)
    want8 = np.rint(np.clip(full, 0.0, 1.0) * 255.0).astype(np.int32)
    assert np.abs(dst8.download().astype(np.int32) - want8).max() <= 1
    for x in (dst8, dst, src, sdf):
        x.close()


@pytest.mark.parametrize("mode", [abi.HDR_NONE, abi.HDR_GAMMA_COMPRESS, abi.HDR_TONE_MAP])
@pytest.mark.parametrize("albedo_fmt", [abi.LIGHTMAP_RGBA8, abi.LIGHTMAP_FLOAT4])
def test_resolve_with_albedo_matches_oracle_on_a_lit_frame(ctx, oracle, mode, albedo_fmt):
    """The ...WithAlbedo techniques (Resolve.fx:43-60,141-233): a lit half4 frame over a Color (or float) albedo texture, ragged width so the
    two-texel fast path and the single-texel tail both run."""
    w, h = 157, 96
    layout = scenes.DistanceFieldLayout(256, 192, 96.0, 12, 0.5, 128)
    atlas = scenes.build_sdf_atlas(layout, scenes.random_obstacles(5, 14, (256, 192), size_lo=8.0, size_hi=30.0, z_hi=40.0))
    dfu = layout.uniforms(max_cone_radius=24.0, power=0.7, step_limit=64, min_step_size=1.0, long_step_factor=0.5)
    lights = scenes.random_lights(6, 12, w, h, z=(8.0, 48.0), radius=10.0, ramp=(40.0, 120.0))
    sdf = native.DistanceFieldTexture(ctx, atlas)
    src = native.Lightmap(ctx, w, h, abi.LIGHTMAP_HALF4)
    native.render_sphere_lights(ctx, lights, scenes.environment(), dfu, None, sdf, (0.05, 0.06, 0.07, 0.35), src)
    lit = src.download().astype(np.float32)
    texels = (scenes.uniform(91, (h, w, 4), 0.0, 1.0) * 255.0).astype(np.uint8)
    tex = native.Lightmap(ctx, w, h, albedo_fmt)
    if albedo_fmt == abi.LIGHTMAP_RGBA8:
        tex.upload(texels)
        albedo = texels.astype(np.float32) / np.float32(255.0)
    else:
        albedo = scenes.uniform(92, (h, w, 4), 0.0, 1.5).astype(np.float32)
        tex.upload(albedo)
    hdr = oc.hdr_configuration(mode, 0.75, 0.02, 1.3, 0.9, 0.5, 0.8, 3.0, 2.5)
    dst = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
    native.resolve_lighting(src, dst, hdr, albedo=tex)
    want = oracle.resolve_lighting(np.ascontiguousarray(lit), hdr, albedo=albedo)
    assert_close(dst.download(), want, "resolved frame with albedo")
    dst8 = native.Lightmap(ctx, w, h, abi.LIGHTMAP_RGBA8)
    native.resolve_lighting(src, dst8, hdr, albedo=tex)
    want8 = np.rint(np.clip(want, 0.0, 1.0) * 255.0).astype(np.int32)
    got8 = dst8.download().astype(np.int32)
    assert np.abs(got8 - want8).max() <= 1
    assert np.array_equal(got8[..., 3], want8[..., 3])          # the albedo's alpha goes straight through
    # strips: rows outside stay untouched
    dst.clear((0.0, 0.0, 0.0, 0.0))
    native.resolve_lighting(src, dst, hdr, 16, 48, albedo=tex)
    part = dst.download()
    assert not part[:16].any() and not part[48:].any()
    assert_close(part[16:48], want[16:48], "strip of the resolved frame with albedo")
    for x in (dst8, dst, tex, src, sdf):
        x.close()


def test_fracture_only_options_are_refused(ctx):
    a = native.Lightmap(ctx, 8, 8, abi.LIGHTMAP_FLOAT4)
    b = native.Lightmap(ctx, 8, 8, abi.LIGHTMAP_FLOAT4)
    hdr = oc.hdr_configuration()
    hdr.ResolveToSRGB = 1
    with pytest.raises(native.IlluminantError) as e:
        native.resolve_lighting(a, b, hdr)
    assert e.value.code == abi.ERR_INVALID_ARGUMENT and "pLinearToPSRGB" in str(e.value)
    hdr = oc.hdr_configuration()
    hdr.AlbedoIsSRGB = 1
    native.resolve_lighting(a, b, hdr)                      # only the with-albedo techniques read it
    t = native.Lightmap(ctx, 8, 8, abi.LIGHTMAP_RGBA8)
    with pytest.raises(native.IlluminantError) as e:
        native.resolve_lighting(a, b, hdr, albedo=t)
    assert e.value.code == abi.ERR_INVALID_ARGUMENT and "pSRGBToPLinear" in str(e.value)
    small = native.Lightmap(ctx, 4, 8, abi.LIGHTMAP_RGBA8)
    with pytest.raises(native.IlluminantError):
        native.resolve_lighting(a, b, oc.hdr_configuration(), albedo=small)
    a.close(); b.close(); t.close(); small.close()
