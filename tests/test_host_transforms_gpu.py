"""The host mirror's remaining transforms / spawners (MatrixMultiply, SpatialNoise, Spawner with a position buffer,
FeedbackSpawner -- SURVEY 8f-2) driving the extended step kernel, replayed descriptor by descriptor on the oracle."""
import numpy as np
import pytest

from illuminant_amd import abi, scenes
from tests.util import assert_close

pytestmark = pytest.mark.gpu

P, V, A, RC, RD = abi.PLANE_POSITION, abi.PLANE_VELOCITY, abi.PLANE_ATTRIBUTES, abi.PLANE_RENDER_COLOR, abi.PLANE_RENDER_DATA


@pytest.fixture(scope="module")
def H():
    from illuminant_amd import _host
    return _host


@pytest.fixture(scope="module")
def hctx(H):
    return H.DeviceContext(0)


def make_engine(H, hctx, chunk_size, seed=7):
    rnd = scenes.randomness_table(seed)
    tp = H.ManualTimeProvider()
    ecfg = H.ParticleEngineConfiguration(chunk_size)
    ecfg.TimeProvider = tp
    return H.ParticleEngine(hctx, ecfg, rnd), tp, rnd


def empty_chunk(n):
    return [np.zeros((n, 4), np.float32) for _ in range(5)]


def compare_system(ps, chunks, rtol=2e-4, atol=2e-5):
    assert len(ps.Chunks) == len(chunks)
    live = 0
    for ci in range(len(chunks)):
        got = [ps.Readback(ci, k) for k in (P, V, A, RC, RD)]
        want = chunks[ci]
        assert np.array_equal(got[0][:, 3] > 0, want[0][:, 3] > 0), "chunk %d live mask" % ci
        m = want[0][:, 3] > 0
        live += int(m.sum())
        for k, name in ((0, "position"), (1, "velocity"), (2, "attributes"), (3, "render color"), (4, "render data")):
            assert_close(got[k][m], want[k][m], "chunk %d %s" % (ci, name), rtol=rtol, atol=atol)
    return live


def test_position_buffer_spawner_with_matrix_multiply_and_spatial_noise(H, hctx, oracle):
    cs = 64
    n = cs * cs
    engine, tp, rnd = make_engine(H, hctx, cs)
    cfg = H.ParticleSystemConfiguration()
    cfg.Friction = 0.05; cfg.MaximumVelocity = 1024.0; cfg.LifeDecayPerSecond = 1.0
    ps = H.ParticleSystem(engine, cfg)
    sp = H.Spawner(5)
    sp.MinRate = sp.MaxRate = 3000.0
    sp.RatePerPosition = False
    f = H.Formula3(); f.Constant = [100, 100, 0]; f.RandomScale = [4, 4, 1]; f.Type = H.FormulaType.Spherical
    sp.Position = f
    sp.AdditionalPositions = [[100.0 + 40.0 * i, 100.0 + 25.0 * (i % 3), float(i)] for i in range(1, 8)]      # 8 positions > 4 inline
    sp.PolygonRate = 3.0
    g = H.Formula3(); g.RandomScale = [30, 30, 5]; g.Type = H.FormulaType.Spherical
    sp.Velocity = g
    life = H.Formula1(); life.Constant = 1.0; life.RandomScale = 2.0
    sp.Life = life
    mm = H.MatrixMultiply()
    mm.Strength = 0.5
    c, s_ = np.cos(0.3), np.sin(0.3)
    mm.Position = [c, s_, 0, 0, -s_, c, 0, 0, 0, 0, 1, 0, 3, -2, 0, 1]      # rotate about z + translate (row-vector convention)
    mm.Velocity = [0.95, 0, 0, 0, 0, 0.95, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]
    sn = H.SpatialNoise(4)
    sn.SpaceScale = [9.0, 5.0]
    v3 = H.NoiseParameters3(); v3.Scale = [25.0, 25.0, 2.0]
    sn.Velocity = v3
    sn.ReplaceOldVelocity = False
    for t in (sp, mm, sn):
        ps.AddTransform(t)
    chunks = []
    for frame in range(10):
        tp.Advance(1.0 / 60.0)
        ps.Update(frame)
        d = abi.StepDesc.from_buffer_copy(ps.LastStepBytes())
        while len(chunks) < len(ps.Chunks):
            chunks.append(empty_chunk(n))
        extras = {}
        for k in range(d.SpawnCount):
            assert d.Spawns[k].Kind == abi.SPAWN_POSITION_BUFFER
            buf = np.zeros((8, 4), np.float32)
            buf[0] = [100, 100, 0, 1.0]
            for i, ap in enumerate(sp.AdditionalPositions):
                buf[i + 1] = list(ap) + [1.0]
            extras[k] = buf
        oracle.step(chunks, cs, rnd, d, spawn_positions=extras)
    hctx.Sync()
    assert 495 <= sp.TotalSpawned <= 500          # 3000/s over 10 steps of 1/60 s, up to the RateError carry
    live = compare_system(ps, chunks)
    assert live > 300


def test_feedback_spawner_consumes_the_source_window(H, hctx, oracle):
    cs = 32
    n = cs * cs
    engine, tp, rnd = make_engine(H, hctx, cs)
    cfg = H.ParticleSystemConfiguration()
    cfg.LifeDecayPerSecond = 0.5
    src = H.ParticleSystem(engine, cfg)
    dst = H.ParticleSystem(engine, cfg)
    sp = H.Spawner(2)
    sp.MinRate = sp.MaxRate = 1200.0                 # 20 per 1/60 s step
    f = H.Formula3(); f.Constant = [200, 100, 5]; f.RandomScale = [50, 50, 0]; f.Type = H.FormulaType.Spherical
    sp.Position = f
    life = H.Formula1(); life.Constant = 3.0; life.RandomScale = 1.0
    sp.Life = life
    src.AddTransform(sp)
    fb = H.FeedbackSpawner(6)
    fb.SourceSystem = src
    fb.MinRate = fb.MaxRate = 1800.0                 # 30 per step = 10 sources x InstanceMultiplier 3
    fb.InstanceMultiplier = 3
    fb.SourceVelocityFactor = 0.5
    fb.MultiplyLife = True
    pf = H.Formula3(); pf.Constant = [0, 0, 1]; pf.RandomScale = [2, 2, 0]; pf.Type = H.FormulaType.Spherical
    fb.Position = pf
    lf = H.Formula1(); lf.Constant = 0.5
    fb.Life = lf
    dst.AddTransform(fb)

    src_chunks, dst_chunks = [], []
    consumed = []
    for frame in range(14):
        tp.Advance(1.0 / 60.0)
        src.Update(frame)
        ds = abi.StepDesc.from_buffer_copy(src.LastStepBytes())
        while len(src_chunks) < len(src.Chunks):
            src_chunks.append(empty_chunk(n))
        oracle.step(src_chunks, cs, rnd, ds)
        dst.Update(frame)
        if len(dst.Chunks) == 0:
            continue
        dd = abi.StepDesc.from_buffer_copy(dst.LastStepBytes())
        while len(dst_chunks) < len(dst.Chunks):
            dst_chunks.append(empty_chunk(n))
        sources = {}
        for k in range(dd.SpawnCount):
            assert dd.Spawns[k].Kind == abi.SPAWN_FEEDBACK and dd.Spawns[k].Feedback.SourceSystem == src.Handle
            sc = src_chunks[dd.Spawns[k].Feedback.SourceChunkIndex]
            sources[k] = (sc[0], sc[1], sc[2])
            consumed.append(int(dd.Spawns[k].Feedback.FeedbackSourceIndex))
        oracle.step(dst_chunks, cs, rnd, dd, feedback_sources=sources)
    hctx.Sync()
    # the feedback source index walks forward by spawnCount / InstanceMultiplier per step (RunSpawner, ParticleSpawning.cs:159-166)
    assert consumed == sorted(consumed) and len(set(consumed)) > 5
    assert all(9 <= b - a <= 11 for a, b in zip(consumed, consumed[1:]))
    assert consumed[-1] + 9 <= src.Chunks[0].TotalConsumedForFeedback <= consumed[-1] + 11
    assert dst.Chunks[0].IsFeedbackSource and not src.Chunks[0].IsFeedbackSource
    compare_system(src, src_chunks)
    live = compare_system(dst, dst_chunks)
    assert live > 100


def test_particle_light_source_and_probes_through_the_renderer(H, hctx, oracle):
    """LightingEnvironment with a sphere light, a ParticleLightSource on a stepped system and two LightProbes: one RenderLighting call
    produces the lightmap and the probe values; the oracle redoes it from the replayed particle state."""
    cs, w, h = 32, 96, 64
    n = cs * cs
    engine, tp, rnd = make_engine(H, hctx, cs)
    cfg = H.ParticleSystemConfiguration()
    cfg.LifeDecayPerSecond = 0.5
    col = H.ParticleColor(); col.OpacityFromLife = 2.0
    cfg.Color = col
    ps = H.ParticleSystem(engine, cfg)
    sp = H.Spawner(8)
    sp.MinRate = sp.MaxRate = 2400.0
    f = H.Formula3(); f.Constant = [48, 32, 6]; f.RandomScale = [40, 26, 3]; f.Type = H.FormulaType.Spherical
    sp.Position = f
    life = H.Formula1(); life.Constant = 0.3; life.RandomScale = 2.0
    sp.Life = life
    c4 = H.Formula4(); c4.Constant = [0.9, 0.7, 0.5, 1.0]
    sp.Color = c4
    ps.AddTransform(sp)
    chunks = []
    for frame in range(5):
        tp.Advance(1.0 / 60.0)
        ps.Update(frame)
        d = abi.StepDesc.from_buffer_copy(ps.LastStepBytes())
        while len(chunks) < len(ps.Chunks):
            chunks.append(empty_chunk(n))
        oracle.step(chunks, cs, rnd, d)

    env = H.LightingEnvironment()
    env.Ambient = [0.03, 0.03, 0.03, 1.0]
    l = H.SphereLightSource()
    l.Position = [20.0, 12.0, 20.0]; l.Radius = 6.0; l.RampLength = 50.0; l.Color = [0.5, 0.6, 0.7, 1.0]
    env.Lights = [l]
    pls = H.ParticleLightSource()
    t = H.SphereLightSource()
    t.Radius = 1.5; t.RampLength = 14.0; t.Color = [1.0, 0.8, 0.6, 0.7]; t.AmbientOcclusionRadius = 4.0; t.AmbientOcclusionOpacity = 0.5
    pls.Template = t
    pls.System = ps
    env.ParticleLights = [pls]
    rc = H.RendererConfiguration(w, h)
    rc.FloatLightmap = True
    r = H.LightingRenderer(hctx, rc, env)
    for (pos, nrm) in (([30.0, 20.0, 4.0], None), ([60.0, 40.0, 2.0], [0.0, 0.0, 1.0])):
        p = H.LightProbe()
        p.Position = pos
        p.Normal = nrm
        r.Probes.Add(p)
    r.RenderLighting(0.5)       # intensityScale 0.5: folded into the sphere light's alpha and divided back out of the probe values
    got = r.ReadLightmap()

    dfu = abi.DistanceFieldUniforms.from_buffer_copy(r.GetDistanceFieldUniformsBytes())
    envu = abi.Environment.from_buffer_copy(r.GetEnvironmentUniformsBytes())
    verts = (abi.LightVertex * 1)()
    verts[0] = abi.LightVertex.from_buffer_copy(H.LightingRenderer.PackSphereLightBytes(l, 0.5, False))
    want, _ = oracle.render_sphere_lights(verts, envu, dfu, None, None, (0.015, 0.015, 0.015, 0.5), w, h)
    params = abi.ParticleLightParams.from_buffer_copy(H.LightingRenderer.PackParticleLightBytes(pls, False))
    assert params.MoreLightProperties.x == 4.0 and params.StippleFactor == 1.0
    quads = [min(n, c.TotalSpawned + 1) for c in ps.Chunks]
    oracle.render_particle_lights(chunks, quads, params, envu, dfu, None, None, want)
    assert_close(got, want, "lightmap", rtol=2e-4, atol=2e-5)
    assert (want[..., 3] > 2.0).any()
    pp = np.asarray([[30.0, 20.0, 4.0, 1.0], [60.0, 40.0, 2.0, 1.0]], np.float32)
    pn = np.asarray([[0, 0, 0, 1.0], [0, 0, 1.0, 1.0]], np.float32)
    pv = oracle.render_light_probes(verts, pp, pn, envu, dfu, None) / 0.5
    pv = (pv * 0.5).astype(np.float16).astype(np.float32) / 0.5        # the probe target is a HalfVector4
    for i in range(2):
        assert_close(np.asarray(r.Probes[i].Value), pv[i], "probe %d" % i, rtol=1e-3, atol=1e-4)
    assert pv[0][3] == 2.0      # one light reached the probe; alpha 1 / intensityScale


def test_auto_readback_and_resolve_through_the_host_mirror(H, hctx, oracle):
    """Configuration.AutoReadback fills ReadbackResult on every Update (MaybePerformReadback); LightingRenderer.Resolve tone-maps the
    frame.  Both against the oracle on the replayed state."""
    from tests import output_common as oc
    cs = 32
    n = cs * cs
    engine, tp, rnd = make_engine(H, hctx, cs)
    cfg = H.ParticleSystemConfiguration()
    cfg.LifeDecayPerSecond = 0.7
    cfg.AutoReadback = True
    cfg.SortedReadback = True
    cfg.RotationFromVelocity = True
    cfg.ZToY = 0.25
    cfg.Size = [6.0, 4.0]
    ap = H.ParticleAppearance()
    ap.TextureSize = [256.0, 128.0]
    ap.SizePx = [64.0, 64.0]
    ap.AnimationRate = [3.0, 0.0]
    ap.RelativeSize = False
    cfg.Appearance = ap
    ps = H.ParticleSystem(engine, cfg)
    sp = H.Spawner(4)
    sp.MinRate = sp.MaxRate = 3000.0
    f = H.Formula3(); f.Constant = [100, 60, 3]; f.RandomScale = [80, 50, 2]; f.Type = H.FormulaType.Spherical
    sp.Position = f
    g = H.Formula3(); g.RandomScale = [50, 50, 0]; g.Type = H.FormulaType.Spherical
    sp.Velocity = g
    life = H.Formula1(); life.Constant = 0.03; life.RandomScale = 1.5
    sp.Life = life
    ps.AddTransform(sp)
    chunks = []
    for frame in range(6):
        tp.Advance(1.0 / 60.0)
        ps.Update(frame)
        d = abi.StepDesc.from_buffer_copy(ps.LastStepBytes())
        while len(chunks) < len(ps.Chunks):
            chunks.append(empty_chunk(n))
        oracle.step(chunks, cs, rnd, d)
    params = abi.ReadbackParams.from_buffer_copy(ps.GetReadbackParamsBytes())
    assert tuple(params.TextureRegion) == (0.0, 0.0, 0.25, 0.5) and abs(params.Size[0] - 6.0 / 256.0) < 1e-9
    elems = [min(n, -(-c.TotalSpawned // cs) * cs) for c in ps.Chunks]
    want, wn = oracle.fill_readback_result(chunks, params, element_counts=elems)
    got = np.frombuffer(ps.ReadbackResultBytes, dtype=np.uint8).reshape(-1, 48)
    assert got.shape[0] == wn and 100 < wn < 300           # some of the shortest-lived particles are already dead
    w = np.frombuffer(want, dtype=np.uint8).reshape(-1, 48)[:wn]
    assert np.array_equal(got[:, 40:44], w[:, 40:44])
    assert_close(got[:, :40].copy().view(np.float32), w[:, :40].copy().view(np.float32), "draw calls", rtol=2e-4, atol=2e-4)

    env = H.LightingEnvironment()
    env.Ambient = [0.2, 0.1, 0.3, 1.0]
    l = H.SphereLightSource()
    l.Position = [40.0, 30.0, 15.0]; l.Radius = 8.0; l.RampLength = 60.0; l.Color = [2.0, 1.5, 1.0, 1.0]
    env.Lights = [l]
    rc = H.RendererConfiguration(96, 64)
    r = H.LightingRenderer(hctx, rc, env)        # HighQuality: HalfVector4 lightmap
    r.RenderLighting(1.0)
    hdr = oc.hdr_configuration(abi.HDR_TONE_MAP, 1.0, 0.0, 1.2, 1.0 / 2.2, white_point=3.0)
    got_img = r.ResolveToArray(bytes(hdr))
    lit = r.ReadLightmap().view(np.float16).astype(np.float32)
    want_img = oracle.resolve_lighting(np.ascontiguousarray(lit), hdr)
    assert_close(got_img, want_img, "resolved frame")
    assert 0.2 < float(want_img[30, 40, 0]) < 1.0
    plain = r.ResolveToArray()                     # hdr == null: the lightmap itself, alpha forced to 1
    assert_close(plain[..., :3], lit[..., :3], "plain resolve")
    assert (plain[..., 3] == 1.0).all()


def replay_pattern(ps, sp, levels, chunks, cs, rnd, oracle):
    d = abi.StepDesc.from_buffer_copy(ps.LastStepBytes())
    while len(chunks) < len(ps.Chunks):
        chunks.append(empty_chunk(cs * cs))
    extras = {}
    for k in range(d.SpawnCount):
        assert d.Spawns[k].Kind == abi.SPAWN_PATTERN
        extras[k] = levels
    oracle.step(chunks, cs, rnd, d, spawn_patterns=extras)
    return d


def test_pattern_spawner_row_by_row(H, hctx, oracle):
    """PatternSpawner (SpecialSpawners.cs:15-264) in incremental mode: one texture row of particles per ParticlesPerRow spawned,
    RowsSpawned cycling through RowsPerInstance; the descriptors the mirror builds equal scenes.pattern_params (the float32
    restatement of SetParameters) and the chunks equal the oracle's replay."""
    cs = 32
    engine, tp, rnd = make_engine(H, hctx, cs)
    cfg = H.ParticleSystemConfiguration()
    cfg.LifeDecayPerSecond = 0.25
    ps = H.ParticleSystem(engine, cfg)
    tw, th = 13, 6                                            # => 16 particles per row, 8 rows per instance
    levels = scenes.pattern_mip_chain(scenes.uniform(77, (th, tw, 4), 0.2, 1.0))
    sp = H.PatternSpawner(3)
    sp.SetTexture(levels)
    assert (sp.ParticlesPerRow, sp.RowsPerInstance, sp.ParticlesPerInstance) == (16, 8, 128) == scenes.pattern_counts(tw, th) + (128,)
    sp.MinRate = sp.MaxRate = 61.0                            # x CountScale 16 x 1/60 s: a little over one row per step
    f = H.Formula3(); f.Constant = [300, 200, 1]; f.RandomScale = [0.5, 0.5, 0]; f.Type = H.FormulaType.Spherical
    sp.Position = f
    life = H.Formula1(); life.Constant = 4.0
    sp.Life = life
    sp.MultiplyColorConstant = False
    ps.AddTransform(sp)
    chunks, rows_seen = [], []
    for frame in range(12):
        tp.Advance(1.0 / 60.0)
        ps.Update(frame)
        d = replay_pattern(ps, sp, levels, chunks, cs, rnd, oracle)
        for k in range(d.SpawnCount):
            rec = d.Spawns[k]
            row = int(rec.Pattern.YOffsetsAndCoordScale[0])
            rows_seen.append(row)
            want = scenes.pattern_params(tw, th, 1, row, multiply_color_constant=False)
            assert bytes(rec.Pattern) == bytes(want)
            assert (rec.Params.ChunkSizeAndIndices[2] - rec.Params.ChunkSizeAndIndices[1] + 1) % 16 == 0     # whole rows only
    hctx.Sync()
    assert rows_seen == [r % 8 for r in range(len(rows_seen))] and len(rows_seen) >= 10
    assert sp.TotalSpawned % 16 == 0 and sp.TotalSpawned >= 160
    live = compare_system(ps, chunks)
    assert 0 < live <= sp.TotalSpawned


def test_pattern_spawner_whole_instances(H, hctx, oracle):
    """WholeSpawn + InstantInitialSpawn (AdjustCurrentRate, SpecialSpawners.cs:141-160): a full instance on the first tick the spawner
    runs, later instances only once the rate has accumulated ParticlesPerInstance; MaximumTotal counts instances."""
    cs = 32
    engine, tp, rnd = make_engine(H, hctx, cs)
    ps = H.ParticleSystem(engine, H.ParticleSystemConfiguration())
    tw, th = 16, 8
    levels = scenes.pattern_mip_chain(scenes.uniform(78, (th, tw, 4), 0.2, 1.0))
    sp = H.PatternSpawner(3)
    sp.SetTexture(levels)
    sp.Divisor = 2                                           # 8 x 4 = 32 particles per instance, colours from mip level 1
    assert sp.Divisor == 2 and sp.ParticlesPerInstance == 32
    sp.Divisor = 99
    assert sp.Divisor == 10                                  # Arithmetic.Clamp(value, 1, 10)
    sp.Divisor = 2
    sp.WholeSpawn = True
    sp.MaximumTotal = 3
    sp.MinRate = sp.MaxRate = 6.0                            # x 32 x 1/60 = 3.2 per tick: ten ticks per instance
    f = H.Formula3(); f.Constant = [300, 200, 1]; f.Type = H.FormulaType.Linear
    sp.Position = f
    ps.AddTransform(sp)
    chunks, totals = [], []
    for frame in range(40):
        tp.Advance(1.0 / 60.0)
        ps.Update(frame)
        replay_pattern(ps, sp, levels, chunks, cs, rnd, oracle)
        totals.append(sp.TotalSpawned)
    hctx.Sync()
    assert totals[0] == 32                                   # instant initial spawn
    assert all(t % 32 == 0 for t in totals) and totals[-1] == 96 and totals[5] == 32     # MaximumTotal 3 instances, none early
    assert sp.RowsSpawned == 0
    live = compare_system(ps, chunks)
    assert live == 96
