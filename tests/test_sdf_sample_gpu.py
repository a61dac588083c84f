"""sampleDistanceFieldEx on the device (ilm_sdf_sample) against the oracle, BIT FOR BIT.

The cone trace's loop exits are discontinuous in the sampled distance, and the renderer tests require the oracle's
exact SDF sample counts, so the device sampler must reproduce every rounding of the restated HLSL arithmetic even
though its integer bookkeeping (slice / 3, % 3, WRAP, unorm16 decode) is implemented differently.
"""
import os

import numpy as np
import pytest

from illuminant_amd import abi, native, scenes

pytestmark = pytest.mark.gpu


def oracle_samples(oracle, positions, dfu, tex):
    return np.array([oracle.sample_distance_field(p, dfu, tex) for p in positions], dtype=np.float32)


def test_unorm16_decode_is_exact_for_every_code(ctx, oracle):
    """All 65536 unorm16 values, read back at texel centres of slice 0 (weights 0: the sample is the texel itself):
    the 3-instruction decode equals value / 65535 exactly."""
    layout = scenes.DistanceFieldLayout(256, 256, 64.0, 3, 1.0)
    assert (layout.atlas_width, layout.atlas_height) == (256, 256)
    dfu = layout.uniforms()
    atlas = np.zeros((256, 256, 4), np.uint16)
    atlas[..., 0] = np.arange(65536, dtype=np.uint32).reshape(256, 256).astype(np.uint16)
    atlas[..., 1] = atlas[..., 0][::-1, ::-1]
    sdf = native.DistanceFieldTexture(ctx, atlas)
    ys, xs = np.mgrid[0:256, 0:256]
    pos = np.stack([xs + 0.5, ys + 0.5, np.zeros_like(xs, dtype=np.float64)], axis=-1).reshape(-1, 3).astype(np.float32)
    got = sdf.sample(dfu, pos)
    want = ((np.float32(192.0) / np.float32(255.0)) - (np.arange(65536, dtype=np.float32) / np.float32(65535.0))) * np.float32(128.0)
    assert np.array_equal(got, want.astype(np.float32))
    # the oracle agrees on a subsample (it is slow per call through ctypes)
    idx = np.arange(0, 65536, 97)
    assert np.array_equal(got[idx], oracle_samples(oracle, pos[idx], dfu, oracle.make_texture(atlas, abi.SDF_UNORM16)))
    sdf.close()


@pytest.mark.parametrize("fmt", [abi.SDF_UNORM16, abi.SDF_FP16])
@pytest.mark.parametrize("packed1", [True, False])
def test_random_positions_match_the_oracle_bit_for_bit(ctx, oracle, fmt, packed1):
    """Multi-row atlas (U WRAP carries the column index past 1.0), every slice pair, positions inside, on the faces
    of and outside the volume, negative z offsets."""
    layout = scenes.DistanceFieldLayout(2048, 2048, 128.0, 32, 0.25)      # 512x512 slices, 3 x 4 atlas (cfg3 / cfg5 shape)
    rng = np.random.default_rng(5)
    atlas = rng.integers(0, 65536, size=(layout.atlas_height, layout.atlas_width, 4), dtype=np.uint16)
    if fmt == abi.SDF_FP16:
        atlas = rng.uniform(0.0, 1.5, size=atlas.shape).astype(np.float16).view(np.uint16)
    dfu = layout.uniforms(z_offset=-3.0, packed1=packed1)
    sdf = native.DistanceFieldTexture(ctx, atlas, fmt)
    n = 6000
    pos = np.empty((n, 3), np.float32)
    pos[:, 0] = rng.uniform(-40.0, 2090.0, n)
    pos[:, 1] = rng.uniform(-40.0, 2090.0, n)
    pos[:, 2] = rng.uniform(-20.0, 140.0, n)
    pos[:200] = np.round(pos[:200])                       # integer coordinates: taps exactly on texel boundaries
    pos[200:260, 0] = 0.0; pos[260:320, 0] = 2048.0      # faces of the volume
    pos[320:380, 2] = 128.0 - 3.0
    got = sdf.sample(dfu, pos)
    want = oracle_samples(oracle, pos, dfu, oracle.make_texture(atlas, fmt))
    assert np.array_equal(got, want), "%d of %d samples differ" % (int((got != want).sum()), n)
    sdf.close()


def test_small_single_column_atlas(ctx, oracle):
    layout = scenes.DistanceFieldLayout(100, 60, 30.0, 3, 1.0)             # 1 physical slice: 100 x 60 atlas, width not a power of two
    rng = np.random.default_rng(9)
    atlas = rng.integers(0, 65536, size=(layout.atlas_height, layout.atlas_width, 4), dtype=np.uint16)
    dfu = layout.uniforms()
    sdf = native.DistanceFieldTexture(ctx, atlas)
    pos = np.stack([rng.uniform(-5, 105, 3000), rng.uniform(-5, 65, 3000), rng.uniform(-3, 33, 3000)], axis=-1).astype(np.float32)
    got = sdf.sample(dfu, pos)
    want = oracle_samples(oracle, pos, dfu, oracle.make_texture(atlas, abi.SDF_UNORM16))
    assert np.array_equal(got, want)
    sdf.close()


def test_non_finite_positions_follow_the_min_max_semantics(ctx, oracle):
    """NaN coordinates reach the sampler in the particle path (normalize(0), UpdateParticleSystemWithDistanceField.fx);
    the device's median-of-three clamp must treat them exactly like the restated min/max form does."""
    layout = scenes.DistanceFieldLayout(256, 256, 64.0, 9, 1.0)
    rng = np.random.default_rng(3)
    atlas = rng.integers(0, 65536, size=(layout.atlas_height, layout.atlas_width, 4), dtype=np.uint16)
    dfu = layout.uniforms()
    sdf = native.DistanceFieldTexture(ctx, atlas)
    nan, inf = np.float32(np.nan), np.float32(np.inf)
    pos = np.array([[nan, 10, 5], [10, nan, 5], [10, 20, nan], [nan, nan, nan], [inf, 10, 5], [-inf, 10, 5], [10, 20, inf],
                    [10, -inf, 5], [-0.0, -0.0, -0.0], [256.0, 256.0, 64.0]], dtype=np.float32)
    got = sdf.sample(dfu, pos)
    want = oracle_samples(oracle, pos, dfu, oracle.make_texture(atlas, abi.SDF_UNORM16))
    assert np.array_equal(got, want, equal_nan=True), (got, want)
    sdf.close()


@pytest.mark.parametrize("fmt", [abi.SDF_UNORM16, abi.SDF_FP16])
def test_multi_row_atlas_relies_on_the_u_wrap(ctx, oracle, fmt):
    """The lighting configs' field: 33 slices in a 3 x 4 atlas.  u = physicalSlice / columns + ... runs past 1 for every atlas row but
    the first, and it is the sampler's U WRAP that folds slice p onto column p % columns (DistanceFieldCommon.fxh:303-311): the
    device's wrap arithmetic must agree with the oracle's on every row, at slice seams and at the volume's faces."""
    layout = scenes.DistanceFieldLayout(2048, 2048, 128.0, 32, 0.25, 128)
    assert (layout.column_count, layout.row_count, layout.slice_count) == (3, 4, 33)
    atlas = scenes.build_sdf_atlas(layout, scenes.random_obstacles(11, 96, (2048, 2048)), fmt=fmt)
    dfu = layout.uniforms(max_cone_radius=24.0, power=0.7, step_limit=64, min_step_size=1.0, long_step_factor=0.5)
    sdf = native.DistanceFieldTexture(ctx, atlas, fmt)
    n = 6000
    pts = scenes.uniform(5, (n, 3), 0, 1) * np.array([2048, 2048, 128], np.float32)
    # seams: x on slice borders (the left tap of texel 0 wraps to the previous column), z on slice boundaries, outside the volume
    pts[:300, 0] = np.repeat(np.array([0.0, 0.5, 1.9, 2047.9, 2048.0, 2100.0], np.float32), 50)
    pts[300:600, 2] = np.linspace(0.0, 128.0, 300, dtype=np.float32)
    pts[600:700, 1] = np.linspace(-40.0, 2090.0, 100, dtype=np.float32)
    got = sdf.sample(dfu, pts)
    want = oracle_samples(oracle, pts, dfu, oracle.make_texture(atlas, fmt))
    assert np.array_equal(got, want), "%d of %d samples differ" % (int((got != want).sum()), n)
    assert np.unique((pts[:, 2] * 33 / 128).astype(int) // 3 // 3).size == 4      # all four atlas rows were visited
    sdf.close()


def test_cone_trace_division_equals_the_ieee_division(ctx):
    """div_no_scale (hlsl_math.hpp) -- the division of the in-volume cone-trace loop, the compiler's IEEE sequence without its two
    v_div_scale_f32 -- against `/` on the device, bit for bit: random mantissas over the exponent range the kernel admits
    (2^-60 <= |d| <= 2^60, |n| <= 2^60), the cone-trace's own operand ranges, and zero / infinite / NaN numerators.  The CPU's
    division (numpy float32) is the third witness."""
    rng = np.random.default_rng(5)
    n_parts, d_parts = [], []
    # the loop's own operands: n = distance + 1.5 in about [-130, 130], d = cone radius in [0.33, 64]
    n_parts.append(rng.uniform(-130.0, 130.0, 1 << 20).astype(np.float32)); d_parts.append(rng.uniform(0.33, 64.0, 1 << 20).astype(np.float32))
    # numerators next to zero: (s + 1.5) cancels down to one ulp of 1.5
    k = rng.integers(-64, 65, 1 << 16).astype(np.float32) * np.float32(2.0 ** -23)
    n_parts.append(k); d_parts.append(rng.uniform(0.33, 24.0, 1 << 16).astype(np.float32))
    # the whole admitted exponent range, random mantissas and signs
    def wide(count, lo, hi):
        m = rng.uniform(1.0, 2.0, count).astype(np.float32)
        e = rng.integers(lo, hi + 1, count)
        sgn = rng.choice(np.array([-1.0, 1.0], np.float32), count)
        return (np.ldexp(m, e) * sgn).astype(np.float32)
    n_parts.append(wide(1 << 20, -40, 59)); d_parts.append(wide(1 << 20, -60, 59))
    # specials in the numerator (an fp16 atlas may hold inf / NaN texels), zero numerators
    sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan], np.float32)
    n_parts.append(np.repeat(sp, 64)); d_parts.append(np.tile(rng.uniform(0.33, 24.0, 64).astype(np.float32), 5))
    n = np.concatenate(n_parts); d = np.concatenate(d_parts)
    fast, ieee = ctx.debug_divide(n, d)
    with np.errstate(all="ignore"):
        cpu = (n / d).astype(np.float32)
    same = (fast.view(np.uint32) == ieee.view(np.uint32)) | (np.isnan(fast) & np.isnan(ieee))
    assert same.all(), "first difference: n=%r d=%r fast=%r ieee=%r" % (n[~same][0], d[~same][0], fast[~same][0], ieee[~same][0])
    same_cpu = (ieee.view(np.uint32) == cpu.view(np.uint32)) | (np.isnan(ieee) & np.isnan(cpu))
    assert same_cpu.all()


# ---- the cone trace's table-driven in-volume sampler (hlsl_math.hpp, sample_inside_table) --------------------------------------

@pytest.mark.parametrize("fmt", [abi.SDF_UNORM16, abi.SDF_FP16])
@pytest.mark.parametrize("resolution,virtual", [(0.25, 2048), (0.125, 4096), (1.0, 256), (0.5, 300), (0.3, 256)])   # 0.3: 77-texel slices, VirtualWidth / SliceWidth is not a float
def test_in_volume_sampler_matches_the_oracle_bit_for_bit(ctx, oracle, fmt, resolution, virtual):
    """The lighting configs' fields (cfg3: 1/4 texel per unit, cfg5: 1/8) and two small ones, random texels (every channel pair,
    every atlas row), positions all over the volume and on the sampler's box faces.  Where the precondition holds the table form
    runs (that must be nearly everywhere inside the volume); every result equals the oracle's bit for bit."""
    layout = scenes.DistanceFieldLayout(virtual, virtual, 128.0, 32 if virtual > 1000 else 9, resolution, 128)
    rng = np.random.default_rng(int(virtual + 1000 * resolution))
    atlas = rng.integers(0, 65536, size=(layout.atlas_height, layout.atlas_width, 4), dtype=np.uint16)
    if fmt == abi.SDF_FP16:
        atlas = rng.uniform(0.0, 1.5, size=atlas.shape).astype(np.float16).view(np.uint16)
        atlas.reshape(-1)[::977] = np.float16(3e-6).view(np.uint16)          # fp16 denormals: v_fma_mix_f32 must read them like v_cvt_f32_f16
    dfu = layout.uniforms(z_offset=-3.0, max_cone_radius=24.0, power=0.7, step_limit=64, min_step_size=1.0, long_step_factor=0.5)
    sdf = native.DistanceFieldTexture(ctx, atlas, fmt)
    n = 8000
    pos = np.empty((n, 3), np.float32)
    pos[:, 0] = rng.uniform(-10.0, virtual + 10.0, n)
    pos[:, 1] = rng.uniform(-10.0, virtual + 10.0, n)
    pos[:, 2] = rng.uniform(-8.0, 130.0, n)
    pos[:300] = np.round(pos[:300])                           # taps exactly on texel boundaries
    isx = virtual / layout.slice_width
    edge = np.float32([0.5 * isx, 0.5625 * isx, 0.6 * isx, virtual - 0.5625 * isx, virtual - 0.6 * isx, isx, virtual - isx])
    pos[300:1000, 0] = np.tile(edge, 100)                     # the faces of the sampler's box
    pos[1000:1700, 1] = np.tile(edge, 100)
    pos[1700:2000, 2] = np.linspace(-3.0, 125.0, 300, dtype=np.float32)   # every slice boundary region
    # the half texel before a slice's first column / row and after its last one (tap origin -1 and slice size - 1: the bordered cells,
    # whose taps bleed into the neighbouring slice like the atlas'), the volume's faces and beyond, z at and past the last valid slice
    rim = np.float32([0.0, 1e-3, 0.25 * isx, 0.49 * isx, 0.5 * isx, virtual, virtual - 1e-3, virtual - 0.25 * isx, virtual - 0.49 * isx, -1.0, virtual + 3.0])
    pos[2000:2550, 0] = np.tile(rim, 50)
    pos[2550:3100, 1] = np.tile(rim, 50)
    pos[3100:3400, 2] = np.float32(rng.choice([-3.0, -3.0 + 1e-3, 124.9, 125.0, 125.0 + 1e-3, 127.0, 128.0, 140.0], 300))
    got, used = sdf.sample_inside(dfu, pos)
    want = oracle_samples(oracle, pos, dfu, oracle.make_texture(atlas, fmt))
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), "%d of %d samples differ (table form used for %d of them); first: %r got %r want %r" % (
        int((~same).sum()), n, int(used[~same].sum()), pos[~same][0], got[~same][0], want[~same][0])
    interior = (pos[:, 0] > isx) & (pos[:, 0] < virtual - isx) & (pos[:, 1] > isx) & (pos[:, 1] < virtual - isx) & (pos[:, 2] > -2.9) & (pos[:, 2] < 120.0)
    if os.environ.get("ILM_SDF_CELLS") != "0":               # (the experiment switch turns the table form off: the results above still agree)
        assert used[interior].mean() > 0.99 and used.sum() > 0.5 * n
    assert not used[(pos[:, 0] < 0) | (pos[:, 1] < 0) | (pos[:, 0] > virtual) | (pos[:, 1] > virtual)].any()
    sdf.close()


def test_in_volume_sampler_is_refused_for_uniforms_that_do_not_describe_the_atlas(ctx, oracle):
    """Uniforms with a doubled slice count or texel size do not describe the bound atlas as whole slices: the table form is never
    used (the general sampler wraps and clamps into the real atlas whatever the uniforms say) and the oracle still agrees."""
    layout = scenes.DistanceFieldLayout(256, 256, 64.0, 9, 1.0)
    rng = np.random.default_rng(4)
    atlas = rng.integers(0, 65536, size=(layout.atlas_height, layout.atlas_width, 4), dtype=np.uint16)
    sdf = native.DistanceFieldTexture(ctx, atlas)
    pos = np.stack([rng.uniform(5, 250, 2000), rng.uniform(5, 250, 2000), rng.uniform(1, 60, 2000)], axis=-1).astype(np.float32)
    good = layout.uniforms()
    _, used = sdf.sample_inside(good, pos)
    assert used.mean() > 0.9
    for tweak in ("slices", "texel", "columns"):
        bad = layout.uniforms()
        if tweak == "slices":
            bad.TextureSliceCount.w = 400.0           # more slices than the table holds
        elif tweak == "texel":
            bad.TextureSliceAndTexelSize.z *= 2.0     # texel size that is not 1 / (virtual width x columns)
        else:
            bad.TextureSliceCount.x = 3.0             # 3 columns of 256 != atlas width 512
        got, used = sdf.sample_inside(bad, pos)
        assert not used.any(), tweak
        want = oracle_samples(oracle, pos, bad, oracle.make_texture(atlas, abi.SDF_UNORM16))
        assert np.array_equal(got, want), tweak
    sdf.close()


def test_divisions_by_the_light_pass_constants_are_exact_for_every_numerator(ctx):
    """Proof by exhaustion: for DOT_RAMP_RANGE and (UNSHADOWED - FULLY_SHADOWED) the shared-reciprocal division of lighting.hip equals
    `/` for every numerator bit pattern inside the range the unscaled division is specified for (2^-60 <= |n| <= 2^60, zero, infinite,
    NaN; the kernel's numerators are a saturate and a sum of unit-vector components plus 0.15)."""
    results = ctx.debug_divide_by_constants()
    assert [round(d, 4) for d, _, _ in results] == [0.15, 0.875]
    assert all(inside == 0 for _, inside, _ in results), results
