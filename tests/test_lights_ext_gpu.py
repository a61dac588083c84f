"""Parity of particle lights (device-side ordered compaction of live particles into light records + the tile-binned light kernel in
accumulate mode) and light probes against the CPU oracle."""
import numpy as np
import pytest

from illuminant_amd import abi, native, scenes
from tests import lights_common as lc
from tests.util import assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("index", range(len(lc.load_cases())))
def test_closed_form_case(ctx, index):
    lc.check_case(lc.load_cases()[index], lc.GpuBackend(ctx))


def particle_scene(cs, n_chunks, width, height, seed=31, dead_fraction=0.5):
    n = cs * cs
    chunks = []
    for c in range(n_chunks):
        pos, vel, attr = scenes.make_particles(seed + c, n, pos_lo=(0, 0, 2), pos_hi=(width, height, 40), dead_fraction=dead_fraction)
        rc = scenes.uniform(seed + 100 + c, (n, 4), 0.0, 1.0).astype(np.float32)
        rc[:, :3] *= rc[:, 3:4]                              # premultiplied, as computeRenderData leaves it
        rc[scenes.uniform(seed + 200 + c, (n,)) < 0.15, 3] = 0.0   # some fully transparent
        chunks.append([pos, vel, attr, rc, np.zeros((n, 4), np.float32)])
    return chunks


def small_field(fmt=abi.SDF_UNORM16):
    layout = scenes.DistanceFieldLayout(256, 192, 96.0, 12, 0.5, 128)
    obstacles = scenes.random_obstacles(5, 14, (256, 192), size_lo=8.0, size_hi=30.0, z_hi=40.0)
    atlas = scenes.build_sdf_atlas(layout, obstacles, fmt=fmt)
    dfu = layout.uniforms(max_cone_radius=24.0, power=0.7, step_limit=64, min_step_size=1.0, long_step_factor=0.5)
    return atlas, dfu


@pytest.mark.parametrize("sfmt", [abi.SDF_UNORM16, abi.SDF_FP16])
def test_particle_lights_on_top_of_sphere_lights(ctx, oracle, sfmt):
    """One frame the way RenderLighting builds it: clear to ambient + sphere lights, then the particle light render state blended on
    top; 2 chunks of 16^2 slots (about 200 lights), with a distance field, AO and specular."""
    w, h, cs = 160, 112, 16
    atlas, dfu = small_field(sfmt)
    env = scenes.environment()
    chunks = particle_scene(cs, 2, w, h)
    quads = [cs * cs, 150]        # the second chunk only spawned 149 particles so far (TotalSpawned + 1)
    params = lc.particle_light_params(3.0, 30.0, (0.9, 0.8, 0.7, 0.6), casts_shadows=True, ao_radius=6.0, ao_opacity=0.7,
                                      spec=(0.2, 0.3, 0.1), spec_power=3.0)
    lights = scenes.random_lights(8, 5, w, h, z=(8.0, 48.0), radius=10.0, ramp=(40.0, 120.0))
    ambient = (0.05, 0.06, 0.07, 1.0)

    eng = native.Engine(ctx, cs, scenes.randomness_table(7))
    sysm = native.System(eng)
    for c, planes in enumerate(chunks):
        sysm.add_chunk()
        sysm.upload(c, abi.PLANE_POSITION, planes[0])
        sysm.upload(c, abi.PLANE_RENDER_COLOR, planes[3])
    sdf = native.DistanceFieldTexture(ctx, atlas, sfmt)
    lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
    native.render_sphere_lights(ctx, lights, env, dfu, None, sdf, ambient, lm)
    stats = native.render_particle_lights(ctx, sysm, params, env, dfu, None, sdf, lm, quad_counts=quads, want_stats=True)
    got = lm.download()

    otex = oracle.make_texture(atlas, sfmt)
    want, _ = oracle.render_sphere_lights(lights, env, dfu, None, otex, ambient, w, h)
    ostats = oracle.render_particle_lights(chunks, quads, params, env, dfu, None, otex, want, want_stats=True)
    assert (stats.SdfSamples, stats.PixelLightPairs, stats.TracedPairs) == (ostats.SdfSamples, ostats.PixelLightPairs, ostats.TracedPairs)
    assert stats.TracedPairs > 1000
    assert_close(got, want, "lightmap")
    assert (want[..., 3] > 3.5).any()       # several particle lights overlap somewhere
    for x in (lm, sdf, sysm, eng):
        x.close()


def test_particle_lights_with_gbuffer_and_row_strip(ctx, oracle):
    w, h, cs = 96, 64, 8
    atlas, dfu = small_field()
    env = scenes.environment()
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    nx = 0.3 * np.sin(xx / 9.0); ny = 0.3 * np.cos(yy / 7.0)
    nz = np.sqrt(np.maximum(1.0 - nx * nx - ny * ny, 0.0))
    z = 6.0 + 5.0 * np.sin(xx / 17.0) * np.cos(yy / 13.0)
    g = scenes.encode_gbuffer(np.stack([nx, ny, nz], axis=-1), 0.0, z)
    g[10:14, :, 3] = 99999.0                                      # a fullbright band: discarded
    envg = scenes.environment(gbuffer_size=(w, h))
    chunks = particle_scene(cs, 3, w, h, seed=77, dead_fraction=0.3)
    params = lc.particle_light_params(2.0, 22.0, (1.0, 1.0, 1.0, 0.5), casts_shadows=True)
    eng = native.Engine(ctx, cs, scenes.randomness_table(7))
    sysm = native.System(eng)
    for c, planes in enumerate(chunks):
        sysm.add_chunk()
        sysm.upload(c, abi.PLANE_POSITION, planes[0])
        sysm.upload(c, abi.PLANE_RENDER_COLOR, planes[3])
    sdf = native.DistanceFieldTexture(ctx, atlas)
    gb = native.GBufferTexture(ctx, g, abi.GBUFFER_FLOAT4)
    lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
    native.render_sphere_lights(ctx, None, envg, dfu, gb, sdf, (0.1, 0.1, 0.1, 1.0), lm)
    native.render_particle_lights(ctx, sysm, params, envg, dfu, gb, sdf, lm, row_begin=16, row_end=48)   # a screen strip (multi-GPU split)
    got = lm.download()
    want = np.zeros((h, w, 4), np.float32)
    want[:] = (0.1, 0.1, 0.1, 1.0)
    oracle.render_particle_lights(chunks, [cs * cs] * 3, params, envg, dfu, oracle.make_texture(g, abi.GBUFFER_FLOAT4),
                                  oracle.make_texture(atlas, abi.SDF_UNORM16), want, row_begin=16, row_end=48)
    assert_close(got, want, "lightmap strip")
    assert np.array_equal(got[:16], want[:16]) and (got[20:44, :, 3] > 1.5).any()
    for x in (lm, gb, sdf, sysm, eng):
        x.close()


def test_particle_lights_follow_the_particle_system(ctx, oracle):
    """P joins L: step a system on the GPU (spawner + update), then light the frame from its live particles; the oracle does both."""
    w, h, cs = 128, 96, 32
    n = cs * cs
    rnd = scenes.randomness_table(7)
    eng = native.Engine(ctx, cs, rnd)
    sysm = native.System(eng)
    sysm.add_chunk()
    chunk = [np.zeros((n, 4), np.float32) for _ in range(5)]
    d = abi.StepDesc()
    d.FirstChunk, d.ChunkCount = 0, -1
    d.System = scenes.system_uniforms(cs, life_decay=0.8)
    d.Update = abi.UpdateParams.default()
    d.UpdateMode = abi.UPDATE_POSITIONS
    first = 0
    for step in range(4):
        d.SpawnCount = 1
        d.Spawns[0].ChunkIndex = 0
        d.Spawns[0].Params = scenes.spawn_params(cs, first, first + 39, first, (0.3 * 253, 0.6 * 127),
                                                 position=((64, 48, 10), (50, 36, 4), (0, 0, 0), scenes.FORMULA_SPHERICAL),
                                                 velocity=((0, 0, 0), (40, 40, 0), (0, 0, 0), scenes.FORMULA_SPHERICAL), life=(0.02 + 0.01 * step, 1.5, 0.0),
                                                 color=((0.8, 0.6, 0.4, 0.9), (0.2, 0.2, 0.2, 0.1), (0, 0, 0, 0)))
        first += 40
        sysm.step(d)
        oracle.step([chunk], cs, rnd, d)
    atlas, dfu = small_field()
    env = scenes.environment()
    params = lc.particle_light_params(2.0, 25.0, (1.0, 0.9, 0.8, 1.0), casts_shadows=True)
    sdf = native.DistanceFieldTexture(ctx, atlas)
    lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
    native.render_sphere_lights(ctx, None, env, dfu, None, sdf, (0.02, 0.02, 0.02, 1.0), lm)
    native.render_particle_lights(ctx, sysm, params, env, dfu, None, sdf, lm, quad_counts=[first + 1])
    got = lm.download()
    want = np.zeros((h, w, 4), np.float32)
    want[:] = (0.02, 0.02, 0.02, 1.0)
    oracle.render_particle_lights([chunk], [first + 1], params, env, dfu, None, oracle.make_texture(atlas, abi.SDF_UNORM16), want)
    live = int((chunk[0][:, 3] > 0).sum())
    assert 40 < live < 160                           # some of the short-lived early particles are already dead
    assert_close(got, want, "lightmap", rtol=2e-4, atol=2e-5)     # particle state itself carries 1e-4-level differences
    for x in (lm, sdf, sysm, eng):
        x.close()


@pytest.mark.parametrize("n_lights", [9, 1, 0, 70])
def test_light_probes_match_oracle(ctx, oracle, n_lights):
    """9 / 70 lights: one wave per (64 probes, light) and the ordered sum (150 probes = three blocks, the last one ragged); one light: the
    one-kernel form; no light: zeros."""
    atlas, dfu = small_field()
    env = scenes.environment()
    lights = scenes.random_lights(8, n_lights, 256, 192, z=(8.0, 48.0), radius=10.0, ramp=(60.0, 160.0))
    n = 150
    pp = np.ones((n, 4), np.float32)
    pp[:, 0] = scenes.uniform(1, (n,), 0, 256); pp[:, 1] = scenes.uniform(2, (n,), 0, 192); pp[:, 2] = scenes.uniform(3, (n,), 0, 40)
    pp[::17, 3] = 0.0                                    # unused probe slots (opacity 0) are discarded
    pn = np.zeros((n, 4), np.float32)
    nrm = scenes.uniform(4, (n, 3), -1, 1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    pn[:, :3] = nrm
    pn[::3, :3] = 0.0                                    # probes without a normal
    pn[:, 3] = (scenes.uniform(5, (n,)) < 0.7)           # EnableShadows
    sdf = native.DistanceFieldTexture(ctx, atlas)
    got = native.render_light_probes(ctx, lights, pp, pn, env, dfu, sdf)
    want = oracle.render_light_probes(lights, pp, pn, env, dfu, oracle.make_texture(atlas, abi.SDF_UNORM16))
    assert np.array_equal(got[:, 3], want[:, 3])         # the number of contributing lights per probe: exact
    assert_close(got, want, "probe values")
    assert not want[::17].any()
    if n_lights >= 9:
        assert (want[:, 3] > 0).mean() > 0.5
    sdf.close()


def test_stipple_factor_below_one_is_refused(ctx):
    eng = native.Engine(ctx, 8, scenes.randomness_table(7))
    sysm = native.System(eng)
    sysm.add_chunk()
    lm = native.Lightmap(ctx, 16, 16, abi.LIGHTMAP_FLOAT4)
    p = lc.particle_light_params(2.0, 5.0)
    p.StippleFactor = 0.5
    with pytest.raises(native.IlluminantError) as e:
        native.render_particle_lights(ctx, sysm, p, scenes.environment(), lc.no_field_uniforms(), None, None, lm)
    assert e.value.code == abi.ERR_INVALID_ARGUMENT and "StippleReject" in str(e.value)
    for x in (lm, sysm, eng):
        x.close()


def test_distance_ramp_technique_matches_oracle(ctx, oracle):
    """Technique SphereLightWithDistanceRamp (SphereLight.fx:48-86): two light groups of one frame -- the first without a ramp texture
    (clears to ambient), the second with an 8 x 4 RGBA ramp, random offsets / rates, added on top -- against the oracle doing the same."""
    from tests.test_lighting_gpu import small_scene
    layout, atlas, dfu, lights, w, h = small_scene(n_lights=10)
    env = scenes.environment()
    plain = scenes.random_lights(21, 5, w, h, z=(8.0, 48.0), radius=10.0, ramp=(40.0, 120.0))
    ramped = (abi.LightVertex * 6)()
    offs = scenes.uniform(31, (6,), -3.0, 3.0); rates = scenes.uniform(32, (6,), 0.5, 3.0)
    xs = scenes.uniform(33, (6,), 0, w); ys = scenes.uniform(34, (6,), 0, h)
    for i in range(6):
        ramped[i] = scenes.sphere_light((float(xs[i]), float(ys[i]), 20.0), 12.0, 90.0, color=(0.9, 0.8, 1.0, 0.9), specular=(0.2, 0.1, 0.3), specular_power=6.0,
                                        ramp_offset=float(offs[i]), ramp_rate=float(rates[i]))
    ramp = scenes.uniform(35, (4, 8, 4), 0.0, 1.0)
    ambient = (0.05, 0.06, 0.07, 1.0)
    sdf = native.DistanceFieldTexture(ctx, atlas, abi.SDF_UNORM16)
    lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
    s1 = native.render_sphere_lights(ctx, plain, env, dfu, None, sdf, ambient, lm, want_stats=True)
    ctx.set_light_ramp(ramp)
    s2 = native.render_sphere_lights(ctx, ramped, env, dfu, None, sdf, None, lm, want_stats=True)      # ambient None: add to the lightmap
    ctx.set_light_ramp(None)
    got = lm.download()
    otex = oracle.make_texture(atlas, abi.SDF_UNORM16)
    want1, o1 = oracle.render_sphere_lights(plain, env, dfu, None, otex, ambient, w, h, want_stats=True)
    oracle.set_light_ramp(ramp)
    want2, o2 = oracle.render_sphere_lights(ramped, env, dfu, None, otex, (0.0, 0.0, 0.0, 0.0), w, h, want_stats=True)
    oracle.set_light_ramp(None)
    for a, b in ((s1, o1), (s2, o2)):
        assert (a.SdfSamples, a.PixelLightPairs, a.TracedPairs) == (b.SdfSamples, b.PixelLightPairs, b.TracedPairs)
    assert_close(got, want1 + want2, "two light groups")
    # the ramp really colours the light: the ramped group is not grey although its lights are nearly white
    lit = want2[..., 3] > 0.5
    assert np.abs(want2[lit][:, 0] - want2[lit][:, 2]).mean() > 0.01
    lm.close(); sdf.close()
