"""Parity of the HIP distance-field generation pass (ilm_sdf_render_slices, csrc/fields.hip) against the CPU oracle:
stored codes are integers, so the comparison is bit-exact."""
import numpy as np
import pytest

from illuminant_amd import abi, native, scenes
from tests import fields_common as fc

pytestmark = pytest.mark.gpu


def render_gpu(ctx, layout, obs=None, vols=None, poly=None, slices=None, fmt=abi.SDF_UNORM16, flt=-1, clear_source=None, keep=False):
    sdf = native.DistanceFieldTexture(ctx, None, fmt, size=(layout.atlas_width, layout.atlas_height))
    d = scenes.render_desc(layout, dynamic_flag_filter=flt)
    sdf.render_slices(d, fc.all_triplets(layout) if slices is None else slices, obs, vols, poly, clear_source=clear_source)
    out = sdf.download()
    if keep:
        return out, sdf
    sdf.close()
    return out


def render_oracle(oracle, layout, obs=None, vols=None, poly=None, slices=None, fmt=abi.SDF_UNORM16, flt=-1, clear_source=None):
    atlas = np.zeros((layout.atlas_height, layout.atlas_width, 4), np.uint16)
    d = scenes.render_desc(layout, dynamic_flag_filter=flt)
    return oracle.render_distance_field_slices(atlas, fmt, d, fc.all_triplets(layout) if slices is None else slices, obs, vols, poly,
                                               clear_source=clear_source)


def test_closed_form_codes(ctx):
    doc = fc.load_generation_fixture()
    layout = fc.fixture_layout(doc)
    for case in doc["cases"]:
        obs, vols, poly = fc.fixture_case_inputs(case)
        atlas = render_gpu(ctx, layout, obs, vols, poly)
        got = fc.texel_of(atlas, layout, case["texel"][0], case["texel"][1], case["slice"])
        assert abs(got - case["expected_code"]) <= 1, (case, got)


@pytest.mark.parametrize("fmt", [abi.SDF_UNORM16, abi.SDF_FP16])
@pytest.mark.parametrize("resolution", [1.0, 0.5, 0.3])
def test_every_type_matches_oracle_bit_for_bit(ctx, oracle, fmt, resolution):
    layout, obs, volumes = fc.mixed_scene(resolution=resolution)
    vols, poly = scenes.height_volume_arrays(volumes)
    arr = scenes.obstruction_array(obs)
    got = render_gpu(ctx, layout, arr, vols, poly, fmt=fmt)
    want = render_oracle(oracle, layout, arr, vols, poly, fmt=fmt)
    assert np.array_equal(got, want), "%d texel channels differ" % int((got != want).sum())
    assert (want > 0).mean() > 0.3


@pytest.mark.parametrize("fmt", [abi.SDF_UNORM16, abi.SDF_FP16])
@pytest.mark.parametrize("max_encoded", [48.0, 128.0, 1400.0])
def test_many_height_volumes_far_apart_match_oracle_bit_for_bit(ctx, oracle, fmt, max_encoded):
    """70 small polygons (some concave, some degenerate: one vertex, two vertices) scattered over a 1024 x 768 world: most texels are
    farther from most volumes than DISTANCE_ZERO x MaximumEncodedDistance, where the kernel culls a volume on the circle around its polygon
    (the reference rasterises 520 units around each) -- every code must still be the oracle's, for a short reach, the usual one, and one
    beyond sdPolygon's own distance cap (999: no culling there)."""
    extent = (1024, 768)
    layout = scenes.DistanceFieldLayout(extent[0], extent[1], 64.0, 6, 0.25, max_encoded)
    r = scenes.uniform(321, (70, 8))
    volumes = []
    for v in range(70):
        cx, cy, rad = r[v, 0] * extent[0], r[v, 1] * extent[1], 5 + r[v, 2] * 40
        nv = 1 + (v % 7)                                     # 1 .. 7 vertices
        ang = np.sort(scenes.uniform(1300 + v, (nv,)) * 2 * np.pi)
        rr = rad * (0.4 + 0.6 * scenes.uniform(1400 + v, (nv,)))       # uneven radii: concave outlines
        volumes.append(([(float(cx + rr[k] * np.cos(ang[k])), float(cy + rr[k] * np.sin(ang[k]))) for k in range(nv)],
                        float(r[v, 4] * 20), float(2 + r[v, 5] * 50), bool(r[v, 6] < 0.5)))
    vols, poly = scenes.height_volume_arrays(volumes)
    got = render_gpu(ctx, layout, None, vols, poly, fmt=fmt)
    want = render_oracle(oracle, layout, None, vols, poly, fmt=fmt)
    assert np.array_equal(got, want), "%d texel channels differ" % int((got != want).sum())
    assert 0.02 < (want > 0).mean()


@pytest.mark.parametrize("typ", [abi.OBSTRUCTION_ELLIPSOID, abi.OBSTRUCTION_BOX, abi.OBSTRUCTION_CYLINDER, abi.OBSTRUCTION_SPHEROID,
                                 abi.OBSTRUCTION_OCTAGON])
def test_single_type_scenes(ctx, oracle, typ):
    layout = scenes.DistanceFieldLayout(200, 150, 80.0, 7, 0.6, 128)
    obs = scenes.random_obstructions(40 + typ, 9, (200, 150), size_lo=5.0, size_hi=45.0, z_hi=60.0, types=(typ,))
    arr = scenes.obstruction_array(obs)
    assert np.array_equal(render_gpu(ctx, layout, arr), render_oracle(oracle, layout, arr))


@pytest.mark.parametrize("fmt", [abi.SDF_UNORM16, abi.SDF_FP16])
@pytest.mark.parametrize("typ", [abi.OBSTRUCTION_ELLIPSOID, abi.OBSTRUCTION_BOX, abi.OBSTRUCTION_CYLINDER, abi.OBSTRUCTION_SPHEROID,
                                 abi.OBSTRUCTION_OCTAGON])
def test_unrotated_obstructions_skip_the_rotation_exactly(ctx, oracle, typ, fmt):
    """An identity orientation lets the kernel skip rotateLocalPosition (two quaternion products whose result is the input up to the
    sign of zero components); the oracle always runs them.  Centres ON texel positions and on slice planes put exact zeros -- of
    both signs -- into the local coordinates; one rotated and one negative-zero-quaternion obstruction keep the general path beside it."""
    layout = scenes.DistanceFieldLayout(192, 128, 64.0, 8, 1.0, 128)        # resolution 1: world = pixel coordinate
    obs = [(typ, (40.0, 32.0, 0.0), (18.0, 11.0, 9.0), 0.0), (typ, (97.0, 64.0, 16.0), (25.0, 25.0, 25.0), 0.0),
           (typ, (150.0, 90.0, 32.0), (9.0, 30.0, 14.0), 0.0), (typ, (60.0, 100.0, 8.0), (14.0, 22.0, 10.0), 0.7),
           (typ, (120.5, 20.25, 40.0), (12.0, 7.0, 20.0), 0.0)]
    arr = scenes.obstruction_array(obs)
    arr[4].Orientation[0] = -0.0; arr[4].Orientation[1] = -0.0; arr[4].Orientation[2] = -0.0      # still the identity (== 0 compares true)
    got = render_gpu(ctx, layout, arr, fmt=fmt)
    want = render_oracle(oracle, layout, arr, fmt=fmt)
    assert np.array_equal(got, want), "%d texel channels differ" % int((got != want).sum())
    assert (want > 0).mean() > 0.2


def test_empty_scene_and_empty_slice_list(ctx, oracle):
    layout = scenes.DistanceFieldLayout(96, 64, 32.0, 6, 1.0, 128)
    assert (render_gpu(ctx, layout) == 0).all()
    sdf = native.DistanceFieldTexture(ctx, np.full((layout.atlas_height, layout.atlas_width, 4), 9, np.uint16))
    sdf.render_slices(scenes.render_desc(layout), [], None)
    assert (sdf.download() == 9).all()          # nothing listed: nothing touched
    sdf.close()


def test_partial_update_and_dynamic_partition(ctx, oracle):
    layout, obs, volumes = fc.mixed_scene(dynamic_fraction=0.4)
    vols, poly = scenes.height_volume_arrays(volumes)
    arr = scenes.obstruction_array(obs)
    # partial update: only the listed triplet changes
    sentinel = np.full((layout.atlas_height, layout.atlas_width, 4), 7, np.uint16)
    sdf = native.DistanceFieldTexture(ctx, sentinel)
    sdf.render_slices(scenes.render_desc(layout), [6], arr, vols, poly)
    got = sdf.download()
    want = sentinel.copy()
    from oracle import oracle as orc
    orc.render_distance_field_slices(want, abi.SDF_UNORM16, scenes.render_desc(layout), [6], arr, vols, poly)
    assert np.array_equal(got, want)
    sdf.close()
    # DynamicDistanceField: static partition, then dynamic partition cleared from the static texture
    static, static_tex = render_gpu(ctx, layout, arr, vols, poly, flt=0, keep=True)
    dynamic = render_gpu(ctx, layout, arr, vols, poly, flt=1, clear_source=static_tex)
    static_tex.close()
    everything = render_oracle(oracle, layout, arr, vols, poly, flt=-1)
    assert np.array_equal(static, render_oracle(oracle, layout, arr, vols, poly, flt=0))
    assert np.array_equal(dynamic, everything)


def test_argument_validation(ctx):
    layout = scenes.DistanceFieldLayout(96, 64, 32.0, 6, 1.0, 128)
    sdf = native.DistanceFieldTexture(ctx, None, size=(layout.atlas_width, layout.atlas_height))
    d = scenes.render_desc(layout)
    with pytest.raises(native.IlluminantError) as e:
        sdf.render_slices(d, [1], None)                     # not a triplet start
    assert e.value.code == abi.ERR_OUT_OF_RANGE
    with pytest.raises(native.IlluminantError) as e:
        sdf.render_slices(d, [3 * layout.column_count * layout.row_count], None)
    assert e.value.code == abi.ERR_OUT_OF_RANGE
    bad = scenes.obstruction_array([(7, (1, 1, 1), (1, 1, 1))])
    with pytest.raises(native.IlluminantError) as e:
        sdf.render_slices(d, [0], bad)
    assert e.value.code == abi.ERR_INVALID_ARGUMENT
    d.SliceWidth += 1
    with pytest.raises(native.IlluminantError) as e:
        sdf.render_slices(d, [0], None)
    assert e.value.code == abi.ERR_INVALID_ARGUMENT
    sdf.close()


def test_full_size_field_of_the_lighting_configs(ctx, oracle):
    """cfg3 / cfg5 field (2048^2 virtual at 1/4 -> 512^2 x 33 slices, 1536 x 2048 atlas, 256 obstructions): the whole atlas in one
    launch equals the oracle's, and the lit frame rendered from the generated field equals the one from the uploaded field."""
    layout = scenes.DistanceFieldLayout(2048, 2048, 128.0, 32, 0.25, 128)
    assert (layout.atlas_width, layout.atlas_height, layout.slice_count) == (1536, 2048, 33)
    old = scenes.random_obstacles(11, 256, (2048, 2048))
    arr = scenes.obstruction_array([(t - 1, c, s) for (t, c, s) in old])
    got = render_gpu(ctx, layout, arr)
    want = render_oracle(oracle, layout, arr)
    assert np.array_equal(got, want)
    # independent numpy rasteriser (the atlas the lighting bench uploaded before this pass existed)
    ref = scenes.build_sdf_atlas(layout, old)
    assert np.abs(got.astype(np.int64) - ref.astype(np.int64)).max() <= 1


@pytest.mark.parametrize("fmt", [abi.GBUFFER_FLOAT4, abi.GBUFFER_HALF4])
def test_gbuffer_generation_matches_oracle_and_feeds_the_light_pass(ctx, oracle, fmt):
    w, h = 160, 112
    volumes = [
        ([(10, 10), (70, 14), (64, 60), (30, 40), (12, 70)], 0.0, 24.0, True, True),       # concave
        ([(50, 30), (150, 30), (150, 100), (50, 100)], 6.0, 30.0, True, False),
        ([(100, 5), (140, 12), (120, 40)], 0.0, 12.0, True, True),
    ]
    vols, poly = scenes.height_volume_arrays(volumes)
    d = scenes.gbuffer_render_desc(ground_z=0.0, viewport_position=(4.0, -3.0), viewport_scale=(1.25, 1.25))
    gb = native.GBufferTexture(ctx, None, fmt, size=(w, h))
    gb.render(d, vols, poly)
    got = gb.download()
    want = oracle.render_gbuffer(w, h, d, vols, poly)
    if fmt == abi.GBUFFER_HALF4:
        assert np.array_equal(got, want.astype(np.float16).view(np.uint16))
    else:
        assert np.array_equal(got, want)
    assert len(np.unique(want[..., 3])) == 4                 # ground + three tops are all visible
    # the generated G-buffer drives the light pass like an uploaded one
    layout = scenes.DistanceFieldLayout(256, 192, 96.0, 12, 0.5, 128)
    atlas = scenes.build_sdf_atlas(layout, scenes.random_obstacles(5, 14, (256, 192), size_lo=8.0, size_hi=30.0, z_hi=40.0))
    dfu = layout.uniforms(max_cone_radius=24.0, power=0.7, step_limit=64, min_step_size=1.0, long_step_factor=0.5)
    lights = scenes.random_lights(6, 6, w, h, z=(30.0, 60.0), radius=10.0, ramp=(40.0, 120.0))
    env = scenes.environment(gbuffer_size=(w, h))
    sdf = native.DistanceFieldTexture(ctx, atlas)
    lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
    native.render_sphere_lights(ctx, lights, env, dfu, gb, sdf, (0.05, 0.05, 0.05, 1.0), lm)
    lit = lm.download()
    ogb = want if fmt == abi.GBUFFER_FLOAT4 else want.astype(np.float16).view(np.uint16)
    ref, _ = oracle.render_sphere_lights(lights, env, dfu, oracle.make_texture(np.ascontiguousarray(ogb), fmt),
                                         oracle.make_texture(atlas, abi.SDF_UNORM16), (0.05, 0.05, 0.05, 1.0), w, h)
    from tests.util import assert_close
    assert_close(lit, ref, "lightmap from the generated G-buffer")
    for x in (lm, sdf, gb):
        x.close()


def test_gbuffer_generation_with_hundreds_of_volumes(ctx, oracle):
    """600 overlapping volumes on a ragged 333 x 197 target: three batches of 256 through the tile lists, later volumes over earlier ones
    in the array's order; a second view (scaled and shifted) moves every volume to other tiles."""
    w, h = 333, 197
    r = scenes.uniform(123, (600, 8))
    volumes = []
    for v in range(600):
        cx, cy, rad = r[v, 0] * w, r[v, 1] * h, 4 + r[v, 2] * 30
        nv = 3 + int(r[v, 3] * 5)
        ang = np.sort(scenes.uniform(900 + v, (nv,)) * 2 * np.pi)
        volumes.append(([(float(cx + rad * np.cos(a_)), float(cy + rad * np.sin(a_))) for a_ in ang], float(r[v, 4] * 8), float(2 + r[v, 5] * 60), True, bool(r[v, 6] < 0.7)))
    vols, poly = scenes.height_volume_arrays(volumes)
    for scale, position in (((1.0, 1.0), (0.0, 0.0)), ((1.5, 0.75), (-30.0, -20.0))):
        d = scenes.gbuffer_render_desc(ground_z=0.0, viewport_position=position, viewport_scale=scale)
        gb = native.GBufferTexture(ctx, None, abi.GBUFFER_FLOAT4, size=(w, h))
        gb.render(d, vols, poly)
        got = gb.download()
        want = oracle.render_gbuffer(w, h, d, vols, poly)
        assert np.array_equal(got, want)
        assert len(np.unique(want[..., 3])) > 100
        gb.close()


def test_generated_ground_plane_equals_the_no_gbuffer_path(ctx):
    """A G-buffer holding only the ground plane at z = 0 decodes to exactly what sampleGBuffer assumes without one (LightCommon.fxh:130-141)."""
    w, h = 96, 64
    gb = native.GBufferTexture(ctx, None, abi.GBUFFER_FLOAT4, size=(w, h))
    gb.render(scenes.gbuffer_render_desc(ground_z=0.0))
    lights = scenes.random_lights(3, 5, w, h, z=(8.0, 40.0), radius=6.0, ramp=(30.0, 80.0))
    dfu = scenes.DistanceFieldLayout(128, 128, 64.0, 6, 0.5).uniforms()
    a = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
    b = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
    native.render_sphere_lights(ctx, lights, scenes.environment(gbuffer_size=(w, h)), dfu, gb, None, (0.1, 0.1, 0.1, 1.0), a)
    native.render_sphere_lights(ctx, lights, scenes.environment(), dfu, None, None, (0.1, 0.1, 0.1, 1.0), b)
    assert np.allclose(a.download(), b.download(), rtol=1e-6, atol=1e-7)
    for x in (a, b, gb):
        x.close()
