"""BASELINE.json's full sizes, checked through size-independent properties (the oracle cannot replay them in
seconds): chunk independence, count checksums, erase idempotence for the particle path (cfg2: 1 M particles);
strip invariance, additivity over lights and a cropped oracle comparison for the ray-march (cfg3: 1080p, 64 lights,
512x512x33 field).
"""
import numpy as np
import pytest

from illuminant_amd import abi, native, scenes
from tests.util import assert_bits_equal, assert_close

pytestmark = pytest.mark.gpu

P, V, A, RC, RD = abi.PLANE_POSITION, abi.PLANE_VELOCITY, abi.PLANE_ATTRIBUTES, abi.PLANE_RENDER_COLOR, abi.PLANE_RENDER_DATA


def cfg2_step(cs, spawn_chunk=None, first=0, last=-1):
    d = abi.StepDesc()
    d.FirstChunk, d.ChunkCount = 0, -1
    d.System = scenes.system_uniforms(cs, friction=0.02, max_velocity=2048.0, life_decay=6.0)
    d.Update = abi.UpdateParams.default()
    d.OpCount = 2
    d.Ops[0].Type = abi.OP_GRAVITY
    d.Ops[0].u.Gravity = scenes.gravity_params([((400., 300., 0.), 70., 600., 1), ((1500., 300., 0.), 150., 900., 1),
                                                ((400., 800., 0.), 200., 1200., 1), ((1500., 800., 0.), 100., 1500., 1)], maximum_acceleration=1024.0)
    d.Ops[1].Type = abi.OP_NOISE
    d.Ops[1].u.Noise = scenes.noise_params(scenes.area_none(), (0.37 * 253, 0.81 * 127), (0.12 * 253, 0.55 * 127), 0.35)
    d.UpdateMode = abi.UPDATE_POSITIONS
    d.Flags = abi.STEP_COUNT_LIVE
    if spawn_chunk is not None:
        d.SpawnCount = 1
        d.Spawns[0].ChunkIndex = spawn_chunk
        d.Spawns[0].Params = scenes.spawn_params(cs, first, last, 0, (0.3 * 253, 0.6 * 127),
                                                 position=((960, 540, 0), (900, 450, 0), (0, 0, 0), scenes.FORMULA_SPHERICAL),
                                                 velocity=((0, 0, 0), (60, 60, 60), (0, 0, 0), scenes.FORMULA_SPHERICAL), life=(3.3, 2.7, 0.0))
    return d


def test_cfg2_one_million_particles_properties(ctx, oracle):
    cs, n_chunks = 256, 16
    n = cs * cs
    rnd = scenes.randomness_table(7)
    eng = native.Engine(ctx, cs, rnd)
    fused = native.System(eng)
    split = native.System(eng)
    pos, vel, attr = scenes.make_particles(1000, n * n_chunks, pos_lo=(0, 0, 0), pos_hi=(1920, 1080, 32), life=(0.01, 0.5), dead_fraction=0.1)
    for s in (fused, split):
        for c in range(n_chunks + 1):
            s.add_chunk()
        for c in range(n_chunks):
            sl = slice(c * n, (c + 1) * n)
            s.upload(c, P, pos[sl]); s.upload(c, V, vel[sl]); s.upload(c, A, attr[sl])
    before = fused.live_counts()
    assert int(before[:n_chunks].sum()) == int((pos[:, 3] > 0).sum()) and before[n_chunks] == 0

    # (1) chunk independence: ONE launch over the 17-chunk table == 17 single-chunk launches, bit for bit.  The 17-chunk launch
    # carries a spawn record and runs the spawning instantiation of the kernel, the spawn-free single-chunk launches the plain
    # one: the same particle gets the same floats whichever variant steps it (no FMA contraction anywhere in the step).
    # The oracle replays the spawn-target chunk and one full chunk through the same three steps.
    d = cfg2_step(cs, spawn_chunk=n_chunks, first=100, last=100 + 1092)
    replay = {c: [pos[c * n:(c + 1) * n].copy(), vel[c * n:(c + 1) * n].copy(), attr[c * n:(c + 1) * n].copy(),
                  np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)] for c in (3,)}
    replay[n_chunks] = [np.zeros((n, 4), np.float32) for _ in range(5)]
    for _ in range(3):
        fused.step(d)
        d3 = _single_chunk(d)
        d3.SpawnCount = 0
        oracle.step([replay[3]], cs, rnd, d3)
        ds = _single_chunk(d)
        ds.SpawnCount = 1
        ds.Spawns[0] = d.Spawns[0]
        ds.Spawns[0].ChunkIndex = 0
        oracle.step([replay[n_chunks]], cs, rnd, ds)
        for c in range(n_chunks + 1):
            dc = cfg2_step(cs, spawn_chunk=n_chunks, first=100, last=100 + 1092)
            dc.FirstChunk, dc.ChunkCount = c, 1
            if c != n_chunks:
                dc.SpawnCount = 0
            split.step(dc)
    counts_fused = fused.step_counts()
    total_live = 0
    for c in range(n_chunks + 1):
        for plane in (P, V, RC, RD):
            a, b = fused.download(c, plane), split.download(c, plane)
            assert_bits_equal(a, b, "chunk %d plane %d fused vs per-chunk launches" % (c, plane))
        # (2) count checksum: fused ballot/popcount == standalone count kernel == count of the downloaded life plane
        life = fused.download(c, P)[:, 3]
        assert counts_fused[c] == int((life > 0).sum())
        total_live += int((life > 0).sum())
    assert np.array_equal(fused.live_counts(), counts_fused)
    assert fused.live_counts(saturate16=True)[n_chunks] == min(counts_fused[n_chunks], 65535)
    # life_decay 6/s over 3 steps of 1/60 s kills every particle that started below 0.3 s: the population moved
    assert 0 < total_live < int(before.sum()) + 1093
    assert counts_fused[n_chunks] == 1093                      # the spawned range is alive (life >= 3.3 s)
    # dead slots are fully zero in every output plane (readStateOrDiscard + cleared target)
    c0 = [fused.download(0, k) for k in (P, V, RC, RD)]
    dead = c0[0][:, 3] <= 0
    assert dead.any() and all(not plane[dead].any() for plane in c0)

    # (2b) the oracle on the spawn-target chunk (the variant the bench's headline times) and on a full chunk, at full size
    for c, want in replay.items():
        got = [fused.download(c, k) for k in (P, V, A, RC, RD)]
        assert counts_fused[c] == oracle.count_live(want[0])
        for k, name in enumerate(("position", "velocity", "attributes", "render color", "render data")):
            assert_close(got[k], want[k], "cfg2 chunk %d %s vs oracle" % (c, name), life_exact=(k == 0))
    assert (replay[n_chunks][0][:, 3] > 0).sum() == 1093

    # (3) Erase is idempotent and total (UpdateParticleSystem.fx:40-49 run twice on Clear, ParticleSystem.cs:819-831)
    fused.erase(-1)
    fused.erase(-1)
    assert not fused.live_counts().any()
    for c in (0, n_chunks):
        for plane in (P, V, RC, RD):
            assert not fused.download(c, plane).any()
    for s in (fused, split):
        s.close()
    eng.close()


@pytest.fixture(scope="module")
def cfg3_scene():
    w, h, n_lights = 1920, 1080, 64
    layout = scenes.DistanceFieldLayout(2048, 2048, 128.0, 32, 0.25)
    assert (layout.atlas_width, layout.atlas_height, layout.slice_count) == (1536, 2048, 33)
    atlas = scenes.build_sdf_atlas(layout, scenes.random_obstacles(11, 256, (2048, 2048)))
    dfu = layout.uniforms(power=0.7, min_step_size=1.0, long_step_factor=0.5)
    lights = scenes.random_lights(12, n_lights, w, h)
    return w, h, layout, atlas, dfu, lights


def render(ctx, lights, env, dfu, sdf, ambient, w, h, fmt=abi.LIGHTMAP_FLOAT4, strips=None, gbuffer=None):
    lm = native.Lightmap(ctx, w, h, fmt)
    for (b, e) in (strips or [(0, h)]):
        native.render_sphere_lights(ctx, lights, env, dfu, gbuffer, sdf, ambient, lm, b, e)
    out = lm.download()
    lm.close()
    return out


def test_cfg3_1080p_properties(ctx, oracle, cfg3_scene):
    w, h, layout, atlas, dfu, lights = cfg3_scene
    # the configured frame (RendererConfiguration.EnableGBuffer's default): the ground-plane G-buffer is bound, every pixel goes
    # through sampleGBuffer's texture branch (LightCommon.fxh:69-144); Vector4 texels here, HalfVector4 in the cfg5 test
    garr = scenes.ground_plane_gbuffer(w, h, abi.GBUFFER_FLOAT4)
    gb = native.GBufferTexture(ctx, garr, abi.GBUFFER_FLOAT4)
    env = scenes.environment(gbuffer_size=(w, h))
    ambient = (0.05, 0.05, 0.05, 1.0)
    sdf = native.DistanceFieldTexture(ctx, atlas)
    whole = render(ctx, lights, env, dfu, sdf, ambient, w, h, gbuffer=gb)
    assert np.isfinite(whole).all()
    # (the generated ground plane reproduces the frame without a G-buffer: sampleGBuffer's else-branch)
    assert_close(render(ctx, lights, scenes.environment(), dfu, sdf, ambient, w, h), whole, "no G-buffer vs the ground-plane G-buffer", rtol=1e-6, atol=1e-7)

    # (1) strip invariance: 8 strips (the 8-GPU split) == one launch, bit for bit
    from illuminant_amd import sharding
    strips = sharding.row_strips(h, 8)
    assert np.array_equal(render(ctx, lights, env, dfu, sdf, ambient, w, h, strips=strips, gbuffer=gb), whole)
    uneven = [(0, 7), (7, 500), (500, 501), (501, h)]          # not tile aligned: still exact
    assert np.array_equal(render(ctx, lights, env, dfu, sdf, ambient, w, h, strips=uneven, gbuffer=gb), whole)

    # (2) additivity: lights [0,32) and [32,64) rendered apart sum to the whole frame (rgb and the alpha light count)
    n = len(lights)
    first = (abi.LightVertex * (n // 2))(*[lights[i] for i in range(n // 2)])
    second = (abi.LightVertex * (n - n // 2))(*[lights[i] for i in range(n // 2, n)])
    zero = (0.0, 0.0, 0.0, 0.0)
    a = render(ctx, first, env, dfu, sdf, ambient, w, h, gbuffer=gb)
    b = render(ctx, second, env, dfu, sdf, zero, w, h, gbuffer=gb)
    assert_close(a + b, whole, "additivity over light subsets", rtol=1e-5, atol=1e-6)
    assert np.array_equal((a + b)[..., 3], whole[..., 3])      # alpha = 1 + number of contributing lights: exact integers

    # (3) no lights => the clear colour (Ambient * intensityScale, LightingRenderer.cs:1013-1024)
    none = render(ctx, (abi.LightVertex * 0)(), env, dfu, sdf, ambient, w, h, gbuffer=gb)
    assert np.array_equal(none, np.broadcast_to(np.float32(ambient), none.shape))

    # (3b) linearity in the lights' colours: every rgb of the frame is a sum of (Color1.rgb * Color1.a) x opacity products, so doubling
    # every light's rgb doubles the frame over a black clear colour EXACTLY (a power of two passes through every product and sum
    # unchanged in its mantissa), and leaves the alpha channel's light count alone
    doubled = (abi.LightVertex * n)(*[lights[i] for i in range(n)])
    for i in range(n):
        doubled[i].Color1.x *= 2.0; doubled[i].Color1.y *= 2.0; doubled[i].Color1.z *= 2.0
    once = render(ctx, lights, env, dfu, sdf, zero, w, h, gbuffer=gb)
    twice = render(ctx, doubled, env, dfu, sdf, zero, w, h, gbuffer=gb)
    assert np.array_equal(twice[..., :3], 2.0 * once[..., :3]) and np.array_equal(twice[..., 3], once[..., 3])
    assert float(once[..., :3].max()) > 0.1

    # (4) the oracle on a 24-row crop of the full-size frame (same lights, same 25 MB atlas)
    b0, b1 = 528, 552
    want, _ = oracle.render_sphere_lights(lights, env, dfu, oracle.make_texture(garr, abi.GBUFFER_FLOAT4), oracle.make_texture(atlas, abi.SDF_UNORM16),
                                          ambient, w, h, row_begin=b0, row_end=b1)
    assert_close(whole[b0:b1], want[b0:b1], "cfg3 crop vs oracle")

    # (5) the fp16-sample variant of the same atlas (config 5's storage) stays within half precision of the unorm16 one
    atlas16 = scenes.build_sdf_atlas(layout, scenes.random_obstacles(11, 256, (2048, 2048)), fmt=abi.SDF_FP16)
    sdf16 = native.DistanceFieldTexture(ctx, atlas16, abi.SDF_FP16)
    half = render(ctx, lights, env, dfu, sdf16, ambient, w, h, gbuffer=gb)
    assert np.abs(half[..., :3] - whole[..., :3]).mean() < 2e-3
    sdf16.close()
    sdf.close()
    gb.close()


def test_cfg4_share_eight_million_particles(ctx, oracle):
    """cfg4's per-GPU share: 8 chunks of 1024^2 slots (8.4 M particles), Gravity + Noise + UpdatePositions.  One launch over the table ==
    per-chunk launches (life bit-exact); live-count checksum; the oracle replays one whole 1 M-slot chunk."""
    cs, n_chunks = 1024, 8
    n = cs * cs
    rnd = scenes.randomness_table(7)
    eng = native.Engine(ctx, cs, rnd)
    fused = native.System(eng)
    split = native.System(eng)
    d = cfg2_step(cs)
    d.System = scenes.system_uniforms(cs, friction=0.02, max_velocity=2048.0, life_decay=4.0)
    keep = {}
    for c in range(n_chunks):
        pos, vel, attr = scenes.make_particles(4000 + c, n, pos_lo=(0, 0, 0), pos_hi=(1920, 1080, 32), life=(0.01, 0.4), dead_fraction=0.05)
        for s in (fused, split):
            s.add_chunk()
            s.upload(c, P, pos); s.upload(c, V, vel); s.upload(c, A, attr)
        if c == 5:
            keep = [pos.copy(), vel.copy(), attr.copy(), np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)]
    steps = 2
    for _ in range(steps):
        fused.step(d)
        for c in range(n_chunks):
            dc = cfg2_step(cs)
            dc.System = d.System
            dc.FirstChunk, dc.ChunkCount = c, 1
            split.step(dc)
        oracle.step([keep], cs, rnd, _single_chunk(d))
    counts = fused.step_counts()
    total = 0
    for c in range(n_chunks):
        a, b = fused.download(c, P), split.download(c, P)
        assert np.array_equal(a, b), "chunk %d: one launch vs per-chunk launches" % c       # same kernel variant: bit-equal
        assert counts[c] == int((a[:, 3] > 0).sum())
        total += int(counts[c])
    assert np.array_equal(fused.live_counts(), counts) and 0 < total < n * n_chunks
    # the oracle on chunk 5 (its 1 M slots through 2 steps)
    got = [fused.download(5, k) for k in (P, V, A, RC, RD)]
    assert np.array_equal(got[0][:, 3] > 0, keep[0][:, 3] > 0)
    m = keep[0][:, 3] > 0
    for k in (0, 1, 3, 4):
        assert_close(got[k][m], keep[k][m], "cfg4 chunk 5 plane %d vs oracle" % k, life_exact=(k == 0))
    for s in (fused, split):
        s.close()
    eng.close()


def test_cfg4_full_64m_particles_on_one_gpu(ctx, oracle):
    """cfg4 whole, resident on ONE device: 64 chunks of 1024^2 slots = 67 M particles (the reference's own ceiling, MaxChunkCount = 64,
    ParticleSystem.cs:49; 5.4 GB of state), Gravity + Noise + UpdatePositions -- the denominator of the north star's 1 -> 8 GPU scaling
    target.  Chunks never interact (ParticleSystem.cs:743-745): the one launch over the 64-chunk table must equal, bit for bit, eight
    launches of an 8-chunk system holding the same images (what one rank of the 8-GPU job steps); the live-count checksum over 64 chunks;
    the oracle replays one whole chunk."""
    cs, n_images, n_groups = 1024, 8, 8
    n = cs * cs
    rnd = scenes.randomness_table(7)
    eng = native.Engine(ctx, cs, rnd)
    big = native.System(eng)
    small = native.System(eng)
    d = cfg2_step(cs)
    d.System = scenes.system_uniforms(cs, friction=0.02, max_velocity=2048.0, life_decay=4.0)
    images = [scenes.make_particles(4100 + c, n, pos_lo=(0, 0, 0), pos_hi=(1920, 1080, 32), life=(0.01, 0.4), dead_fraction=0.05) for c in range(n_images)]

    def image(g, c):
        """group g's copy of image c: shifted, so that no two of the 64 chunks hold the same particles"""
        pos, vel, attr = images[c]
        p = pos.copy()
        live = p[:, 3] > 0
        p[live, 0] += np.float32(3.0 * g)
        p[live, 1] += np.float32(2.0 * g)
        return p, vel, attr

    for c in range(n_images):
        small.add_chunk()
    for g in range(n_groups):
        for c in range(n_images):
            big.add_chunk()
            p, v, a = image(g, c)
            big.upload(g * n_images + c, P, p); big.upload(g * n_images + c, V, v); big.upload(g * n_images + c, A, a)
    assert big.chunk_count() == 64
    steps = 2
    for _ in range(steps):
        big.step(d)
    counts = big.step_counts()
    assert counts.shape[0] == 64
    keep_g, keep_c = 5, 3
    total = 0
    for g in range(n_groups):
        for c in range(n_images):
            p, v, a = image(g, c)
            small.upload(c, P, p); small.upload(c, V, v); small.upload(c, A, a)
            if (g, c) == (keep_g, keep_c):
                keep = [p.copy(), v.copy(), a.copy(), np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)]
        for _ in range(steps):
            small.step(d)
        small_counts = small.step_counts()
        for c in range(n_images):
            k = g * n_images + c
            a_, b_ = big.download(k, P), small.download(c, P)
            assert np.array_equal(a_.view(np.uint32), b_.view(np.uint32)), "chunk %d of the 64-chunk launch vs chunk %d of an 8-chunk launch" % (k, c)
            assert counts[k] == small_counts[c] == int((a_[:, 3] > 0).sum())
            total += int(counts[k])
        for plane in (V, RC, RD):      # one chunk per group through the other output planes
            assert np.array_equal(big.download(g * n_images + g, plane).view(np.uint32), small.download(g, plane).view(np.uint32))
    assert np.array_equal(big.live_counts(), counts) and 0 < total < 64 * n
    for _ in range(steps):
        oracle.step([keep], cs, rnd, _single_chunk(d))
    k = keep_g * n_images + keep_c
    got = [big.download(k, plane) for plane in (P, V, A, RC, RD)]
    assert np.array_equal(got[0][:, 3] > 0, keep[0][:, 3] > 0)
    m = keep[0][:, 3] > 0
    for plane in (0, 1, 3, 4):
        assert_close(got[plane][m], keep[plane][m], "cfg4 (64 M) chunk %d plane %d vs oracle" % (k, plane), life_exact=(plane == 0))
    for s in (big, small):
        s.close()
    eng.close()


def _single_chunk(d):
    import ctypes
    c = abi.StepDesc()
    ctypes.memmove(ctypes.byref(c), ctypes.byref(d), ctypes.sizeof(d))
    c.FirstChunk, c.ChunkCount = 0, -1
    return c


def test_cfg5_4k_256_lights_fp16_properties(ctx, oracle):
    """cfg5's frame: 3840 x 2160, 256 lights, fp16-sample field generated on the device from 256 obstructions.  8-strip invariance (the
    8-GPU screen split), the oracle on an 8-row crop with exact SDF sample / pair / trace counts."""
    w, h = 3840, 2160
    layout = scenes.DistanceFieldLayout(4096, 4096, 128.0, 32, 0.125, 128)
    obstacles = scenes.random_obstacles(11, 256, (4096, 4096))
    sdf = native.DistanceFieldTexture(ctx, None, abi.SDF_FP16, size=(layout.atlas_width, layout.atlas_height))
    sdf.render_slices(scenes.render_desc(layout), list(range(0, layout.slice_count, 3)), scenes.obstruction_array([(t - 1, c, s) for (t, c, s) in obstacles]))
    atlas = sdf.download()
    dfu = layout.uniforms(max_cone_radius=24.0, power=0.7, step_limit=64, min_step_size=1.0, long_step_factor=0.5)
    lights = scenes.random_lights(13, 256, w, h, z=(8.0, 64.0), radius=24.0, ramp=(400.0, 1100.0))
    # the configured frame: the ground-plane G-buffer bound (HalfVector4 texels here, Vector4 in the cfg3 test)
    garr = scenes.ground_plane_gbuffer(w, h, abi.GBUFFER_HALF4)
    gb = native.GBufferTexture(ctx, garr, abi.GBUFFER_HALF4)
    env = scenes.environment(gbuffer_size=(w, h))
    ambient = (0.05, 0.05, 0.05, 1.0)
    whole = render(ctx, lights, env, dfu, sdf, ambient, w, h, gbuffer=gb)
    assert np.isfinite(whole).all()
    from illuminant_amd import sharding
    assert np.array_equal(render(ctx, lights, env, dfu, sdf, ambient, w, h, strips=sharding.row_strips(h, 8), gbuffer=gb), whole)
    b0, b1 = 1076, 1084
    lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
    stats = native.render_sphere_lights(ctx, lights, env, dfu, gb, sdf, ambient, lm, b0, b1, want_stats=True)
    lm.close()
    gb.close()
    want, ostats = oracle.render_sphere_lights(lights, env, dfu, oracle.make_texture(garr, abi.GBUFFER_HALF4), oracle.make_texture(atlas, abi.SDF_FP16), ambient, w, h,
                                               row_begin=b0, row_end=b1, want_stats=True)
    assert (stats.SdfSamples, stats.PixelLightPairs, stats.TracedPairs) == (ostats.SdfSamples, ostats.PixelLightPairs, ostats.TracedPairs)
    assert stats.SdfSamples > 5_000_000
    assert_close(whole[b0:b1], want[b0:b1], "cfg5 crop vs oracle")
    sdf.close()


def test_largest_chunk_size_indexes_every_slot(ctx, oracle):
    """ilm_engine_create's upper bound: one chunk of 4096^2 = 2^24 slots (the largest count a float slot index still holds exactly,
    ChunkSizeAndIndices carries indices as floats; 268 MB per plane).  Gravity + Noise (64 units per row: the per-slot lookup path, not the
    host-evaluated run tables) + UpdatePositions, and a spawn range that ends on the chunk's very last slot; the oracle replays the chunk."""
    cs = 4096
    n = cs * cs
    rnd = scenes.randomness_table(7)
    eng = native.Engine(ctx, cs, rnd)
    with pytest.raises(native.IlluminantError):
        native.Engine(ctx, cs + 1, rnd)
    sysm = native.System(eng)
    sysm.add_chunk()
    pos, vel, attr = scenes.make_particles(77, n, pos_lo=(0, 0, 0), pos_hi=(1920, 1080, 32), life=(0.01, 0.4), dead_fraction=0.3)
    pos[n - 5000:, 3] = 0.0                       # room for the spawner at the end of the chunk
    sysm.upload(0, P, pos); sysm.upload(0, V, vel); sysm.upload(0, A, attr)
    chunk = [pos, vel, attr, np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)]
    d = cfg2_step(cs, spawn_chunk=0, first=n - 3000, last=n - 1)
    d.System = scenes.system_uniforms(cs, friction=0.02, max_velocity=2048.0, life_decay=4.0)
    sysm.step(d)
    want_counts = oracle.step([chunk], cs, rnd, d, want_counts=True)
    assert np.array_equal(sysm.step_counts(), want_counts)
    got_p = sysm.download(0, P)
    assert np.array_equal(got_p[:, 3] > 0, chunk[0][:, 3] > 0)
    assert (got_p[n - 3000:, 3] > 0).all() and not (got_p[n - 5000:n - 3000, 3] > 0).any()
    m = chunk[0][:, 3] > 0
    assert_close(got_p[m], chunk[0][m], "4096^2 chunk: position/life")
    del got_p
    for k, pl in ((1, V), (3, RC), (4, RD)):
        got = sysm.download(0, pl)
        assert_close(got[m], chunk[k][m], "4096^2 chunk: plane %d" % k)
        del got
    sysm.close(); eng.close()
