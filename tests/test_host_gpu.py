"""The host mirror of the reference's C# interface (illuminant_amd/host: ParticleEngine / ParticleSystem /
Transforms / DistanceField / LightingRenderer) driving the HIP kernels through the C ABI, checked against the
oracle.  These read like tests the reference would have: build a system the way TestGame's scenes do, Update it,
read the state back.
"""
import json
import os

import numpy as np
import pytest

from illuminant_amd import abi, scenes
from tests.util import assert_close

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
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


def test_spawner_slot_allocation_matches_the_fixture(H, hctx):
    """RunSpawner / PickTargetForSpawn through ParticleSystem.Update: the slot ranges the native step receives are
    the ones derived from ParticleSpawning.cs:115-231 (tests/golden/spawner.json) -- bit-exact slot indices."""
    doc = json.load(open(os.path.join(GOLDEN, "spawner.json")))
    for case in doc["cases"]:
        if case["kind"] != "allocation":
            continue
        cs = int(round(case["chunk_capacity"] ** 0.5))
        engine, tp, _rnd = make_engine(H, hctx, cs)
        cfg = H.ParticleSystemConfiguration()
        cfg.LifeDecayPerSecond = 0.0
        ps = H.ParticleSystem(engine, cfg)
        sp = H.Spawner(5)
        sp.MinRate, sp.MaxRate = case["min_rate"], case["max_rate"]
        life = H.Formula1(); life.Constant = 10.0
        sp.Life = life
        sp.ScriptedDraws = [d for t in case["trace"] for d in t["draws"][:max(1, len(t["issued"]))]]
        # the very first Update of a system runs with dt = 1/60 whatever the clock says (ParticleSystem.cs:647-650):
        # take it before the spawner is attached, so every tick of the fixture sees the clock's dt
        tp.Advance(case["dt"])
        ps.Update(0)
        ps.AddTransform(sp)
        for frame, tick in enumerate(case["trace"]):
            tp.Advance(case["dt"])
            ps.Update(frame + 1)
            assert ps.LastDeltaTimeSeconds == case["dt"]
            d = abi.StepDesc.from_buffer_copy(ps.LastStepBytes())
            got = [[d.Spawns[k].ChunkIndex + 1, int(d.Spawns[k].Params.ChunkSizeAndIndices[1]), int(d.Spawns[k].Params.ChunkSizeAndIndices[2])]
                   for k in range(d.SpawnCount)]
            assert got == tick["issued"], (frame, got, tick["issued"])
            assert sp.TotalSpawned == tick["total_spawned_after"]
            assert sp.RateError == pytest.approx(tick["rate_error_after"], abs=1e-9)
        # every spawned slot is alive on the device (life 10, no decay): the GPU saw exactly those ranges
        hctx.Sync()
        total = 0
        for ci in range(len(ps.Chunks)):
            pos = ps.Readback(ci, P)
            alive = np.flatnonzero(pos[:, 3] > 0)
            n = ps.Chunks[ci].NextSpawnOffset
            assert np.array_equal(alive, np.arange(n)), "chunk %d: live slots are not the bump-allocated prefix" % ci
            total += n
        assert total == case["trace"][-1]["total_spawned_after"]


def test_update_loop_matches_the_oracle_step_by_step(H, hctx, oracle):
    """SimpleParticles-like system (Spawner + Gravity + Noise, Scenes/SimpleParticles.cs) for 12 Updates; the oracle
    replays the very descriptors the host mirror launched, pass by pass."""
    cs = 64
    n = cs * cs
    engine, tp, rnd = make_engine(H, hctx, cs)
    cfg = H.ParticleSystemConfiguration()
    cfg.Friction = 0.1; cfg.MaximumVelocity = 2048.0; cfg.LifeDecayPerSecond = 1.5
    col = H.ParticleColor(); col.OpacityFromLife = 2.5
    cfg.Color = col
    ps = H.ParticleSystem(engine, cfg)
    ps.BlockingLivenessReadback = True
    pos, vel, attr = scenes.make_particles(77, n * 2, pos_lo=(0, 0, 0), pos_hi=(1920, 1080, 32), life=(0.05, 4.0))
    ps.Spawn(n * 2, pos, vel, attr)
    sp = H.Spawner(3)
    sp.MinRate, sp.MaxRate = 20000.0, 60000.0
    f = H.Formula3(); f.Constant = [960, 540, 0]; f.RandomScale = [900, 450, 0]; f.Type = H.FormulaType.Spherical
    sp.Position = f
    g = H.Formula3(); g.RandomScale = [60, 60, 60]; g.Type = H.FormulaType.Spherical
    sp.Velocity = g
    life = H.Formula1(); life.Constant = 0.2; life.RandomScale = 2.7
    sp.Life = life
    gr = H.Gravity(); gr.MaximumAcceleration = 1024.0
    atts = []
    for (p, r, s) in (((400., 300., 0.), 70., 600.), ((1500., 800., 0.), 100., 1500.)):
        a = H.Attractor(); a.Position = list(p); a.Radius = r; a.Strength = s; a.Type = H.AttractorType.Linear
        atts.append(a)
    gr.Attractors = atts
    nz = H.Noise(9)
    for t in (sp, gr, nz):
        ps.AddTransform(t)

    chunks = [[pos[c * n:(c + 1) * n].copy(), vel[c * n:(c + 1) * n].copy(), attr[c * n:(c + 1) * n].copy(),
               np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)] for c in range(2)]
    for frame in range(12):
        tp.Advance(1.0 / 60.0)
        ps.Update(frame)
        d = abi.StepDesc.from_buffer_copy(ps.LastStepBytes())
        while len(chunks) < len(ps.Chunks):
            chunks.append([np.zeros((n, 4), np.float32) for _ in range(5)])
        oracle.step(chunks, cs, rnd, d)
    hctx.Sync()
    assert len(ps.Chunks) == len(chunks) >= 3
    live = 0
    for ci in range(len(chunks)):
        got = [ps.Readback(ci, k) for k in (P, V, A, RC, RD)]
        want = chunks[ci]
        # liveness: bit-exact
        assert np.array_equal(got[0][:, 3] > 0, want[0][:, 3] > 0), "chunk %d live mask" % ci
        m = want[0][:, 3] > 0
        live += int(m.sum())
        for k, name in ((0, "position"), (1, "velocity"), (3, "render color"), (4, "render data")):
            assert_close(got[k][m], want[k][m], "chunk %d %s" % (ci, name), rtol=2e-4, atol=2e-5)   # 12 steps of 1e-4-level error
        assert not got[0][~m].any() and not got[3][~m].any()
    # particles died and were spawned during the run (the scenario exercises both)
    assert sp.TotalSpawned > 0 and 0 < live < 2 * n + sp.TotalSpawned
    # LiveCount from the fused ballot/popcount reduction == the oracle's count at the last liveness check
    assert ps.LiveCount > 0


def test_lighting_renderer_matches_the_oracle(H, hctx, oracle):
    w, h = 160, 96
    env = H.LightingEnvironment()
    env.Ambient = [0.05, 0.04, 0.03, 1.0]
    lv = scenes.random_lights(21, 9, w, h, z=(8.0, 40.0), radius=8.0, ramp=(40.0, 90.0))
    lights = []
    for i in range(len(lv)):
        l = H.SphereLightSource()
        l.Position = [lv[i].LightPosition1.x, lv[i].LightPosition1.y, lv[i].LightPosition1.z]
        l.Radius = lv[i].LightProperties.x; l.RampLength = lv[i].LightProperties.y
        l.Color = [lv[i].Color1.x, lv[i].Color1.y, lv[i].Color1.z, 1.0]
        if i % 3 == 0:
            l.AmbientOcclusionRadius = 12.0; l.AmbientOcclusionOpacity = 0.8
        if i % 4 == 1:
            l.SpecularColor = [0.3, 0.2, 0.1]; l.SpecularPower = 8.0
        lights.append(l)
    env.Lights = lights
    rc = H.RendererConfiguration(w, h)
    rc.FloatLightmap = True
    q = H.RendererQualitySettings(); q.MinStepSize = 1.0; q.LongStepFactor = 0.5; q.MaxStepCount = 64; q.MaxConeRadius = 24.0; q.OcclusionToOpacityPower = 0.7
    rc.DefaultQuality = q
    r = H.LightingRenderer(hctx, rc, env)
    field = H.DistanceField(hctx, 256, 256, 64.0, 9, 0.5)
    layout = scenes.DistanceFieldLayout(256, 256, 64.0, 9, 0.5)
    atlas = scenes.build_sdf_atlas(layout, scenes.random_obstacles(5, 10, (256, 256), 6.0, 24.0, 40.0))
    field.Load(atlas)
    r.DistanceField = field
    stats = r.RenderLighting(1.0, 0, -1, True)
    got = r.ReadLightmap()
    dfu = abi.DistanceFieldUniforms.from_buffer_copy(r.GetDistanceFieldUniformsBytes())
    envu = abi.Environment.from_buffer_copy(r.GetEnvironmentUniformsBytes())
    # the uniforms the host mirror packs are the ones the scene builder derives from the same reference lines
    assert bytes(dfu) == bytes(layout.uniforms(power=0.7, min_step_size=1.0, long_step_factor=0.5))
    packed = (abi.LightVertex * len(lights))()
    for i in range(len(lights)):
        packed[i] = scenes.sphere_light(tuple(lights[i].Position), lights[i].Radius, lights[i].RampLength, color=tuple(lights[i].Color),
                                        ao_radius=lights[i].AmbientOcclusionRadius, ao_opacity=lights[i].AmbientOcclusionOpacity,
                                        specular=tuple(lights[i].SpecularColor), specular_power=lights[i].SpecularPower)
    want, wstats = oracle.render_sphere_lights(packed, envu, dfu, None, oracle.make_texture(atlas, abi.SDF_UNORM16), tuple(env.Ambient), w, h,
                                               want_stats=True)
    assert_close(got, want, "lightmap via LightingRenderer.RenderLighting")
    # integer statistics of the trace are bit-exact: same pixel-light pairs, same traced pairs, same SDF sample count
    assert tuple(int(x) for x in stats) == (wstats.SdfSamples, wstats.PixelLightPairs, wstats.TracedPairs)


def test_reference_error_behaviour(H, hctx):
    engine, tp, _ = make_engine(H, hctx, 16)
    ps = H.ParticleSystem(engine, H.ParticleSystemConfiguration())
    tp.Advance(1 / 60); ps.Update(0)
    with pytest.raises(Exception, match="Cannot update twice in a single frame"):     # ParticleSystem.cs:641-642
        ps.Update(0)
    gr = H.Gravity()
    atts = []
    for i in range(17):
        a = H.Attractor(); a.Position = [float(i), 0.0, 0.0]
        atts.append(a)
    gr.Attractors = atts
    ps.AddTransform(gr)
    pos, vel, attr = scenes.make_particles(1, 256)
    ps.Spawn(256, pos, vel, attr)
    tp.Advance(1 / 60)
    with pytest.raises(Exception, match="Maximum number of attractors"):              # Transforms.cs:348-349
        ps.Update(1)


def test_lighting_renderer_groups_lights_by_ramp_texture(H, hctx, oracle):
    """GetLightRenderState (LightingRenderer.cs:799-845): lights sharing a ramp texture form one render state; a light's TextureRef wins over
    Configuration.DefaultRampTexture; a 1 x 1 ramp is no ramp.  Three groups on one frame: default ramp, own ramp, 1 x 1 ramp."""
    w, h = 128, 80
    env = H.LightingEnvironment()
    env.Ambient = [0.02, 0.03, 0.04, 1.0]
    ramp_a = scenes.uniform(61, (2, 6, 4), 0.0, 1.0)
    ramp_b = scenes.uniform(62, (5, 3, 4), 0.0, 1.0)
    ta, tb, t1 = H.RampTexture(ramp_a), H.RampTexture(ramp_b), H.RampTexture(np.full((1, 1, 4), 0.3, np.float32))
    lv = scenes.random_lights(23, 9, w, h, z=(8.0, 40.0), radius=8.0, ramp=(40.0, 90.0))
    lights, groups = [], {"a": [], "b": [], "none": []}
    for i in range(len(lv)):
        l = H.SphereLightSource()
        l.Position = [lv[i].LightPosition1.x, lv[i].LightPosition1.y, lv[i].LightPosition1.z]
        l.Radius = lv[i].LightProperties.x; l.RampLength = lv[i].LightProperties.y
        l.Color = [lv[i].Color1.x, lv[i].Color1.y, lv[i].Color1.z, 1.0]
        l.RampOffset = 0.3 * i; l.RampRate = 1.0 + 0.25 * i
        key = ("a", "b", "none")[i % 3]
        if key == "b":
            l.TextureRef = tb
        elif key == "none":
            l.TextureRef = t1
        lights.append(l)
        groups[key].append(scenes.sphere_light(tuple(l.Position), l.Radius, l.RampLength, color=tuple(l.Color), ramp_offset=l.RampOffset, ramp_rate=l.RampRate,
                                                have_distance_field=False))
    env.Lights = lights
    rc = H.RendererConfiguration(w, h)
    rc.FloatLightmap = True
    rc.DefaultRampTexture = ta
    r = H.LightingRenderer(hctx, rc, env)
    stats = r.RenderLighting(1.0, 0, -1, True)
    got = r.ReadLightmap()
    dfu = abi.DistanceFieldUniforms.from_buffer_copy(r.GetDistanceFieldUniformsBytes())
    envu = abi.Environment.from_buffer_copy(r.GetEnvironmentUniformsBytes())
    want = np.zeros((h, w, 4), np.float32)
    totals = [0, 0, 0]
    first = True
    for key, ramp in (("a", ramp_a), ("b", ramp_b), ("none", None)):       # order of first appearance
        arr = (abi.LightVertex * len(groups[key]))(*groups[key])
        oracle.set_light_ramp(ramp)
        part, st = oracle.render_sphere_lights(arr, envu, dfu, None, None, tuple(env.Ambient) if first else (0.0, 0.0, 0.0, 0.0), w, h, want_stats=True)
        oracle.set_light_ramp(None)
        want += part
        totals = [totals[0] + st.SdfSamples, totals[1] + st.PixelLightPairs, totals[2] + st.TracedPairs]
        first = False
    assert_close(got, want, "three ramp groups")
    assert [int(x) for x in stats] == totals


def test_lighting_renderer_groups_lights_by_quality(H, hctx, oracle):
    """LightSource.Quality (LightSource.cs:95) is part of the render-state key: lights with their own quality settings are traced with
    those (SetDistanceFieldParameters(material, true, key.Quality), LightingRenderer.cs:792), the others with Configuration.DefaultQuality."""
    w, h = 128, 80
    env = H.LightingEnvironment()
    env.Ambient = [0.02, 0.03, 0.04, 1.0]
    lv = scenes.random_lights(29, 8, w, h, z=(8.0, 40.0), radius=8.0, ramp=(40.0, 90.0))
    coarse = H.RendererQualitySettings(); coarse.MinStepSize = 4.0; coarse.LongStepFactor = 1.0; coarse.MaxStepCount = 12; coarse.MaxConeRadius = 8.0
    coarse.OcclusionToOpacityPower = 1.0
    lights, packed = [], {"default": [], "coarse": []}
    for i in range(len(lv)):
        l = H.SphereLightSource()
        l.Position = [lv[i].LightPosition1.x, lv[i].LightPosition1.y, lv[i].LightPosition1.z]
        l.Radius = lv[i].LightProperties.x; l.RampLength = lv[i].LightProperties.y
        l.Color = [lv[i].Color1.x, lv[i].Color1.y, lv[i].Color1.z, 1.0]
        if i % 2:
            l.Quality = coarse
        lights.append(l)
        packed["coarse" if i % 2 else "default"].append(scenes.sphere_light(tuple(l.Position), l.Radius, l.RampLength, color=tuple(l.Color)))
    env.Lights = lights
    rc = H.RendererConfiguration(w, h)
    rc.FloatLightmap = True
    q = H.RendererQualitySettings(); q.MinStepSize = 1.0; q.LongStepFactor = 0.5; q.MaxStepCount = 64; q.MaxConeRadius = 24.0; q.OcclusionToOpacityPower = 0.7
    rc.DefaultQuality = q
    r = H.LightingRenderer(hctx, rc, env)
    field = H.DistanceField(hctx, 256, 256, 64.0, 9, 0.5)
    layout = scenes.DistanceFieldLayout(256, 256, 64.0, 9, 0.5)
    atlas = scenes.build_sdf_atlas(layout, scenes.random_obstacles(5, 10, (256, 256), 6.0, 24.0, 40.0))
    field.Load(atlas)
    r.DistanceField = field
    stats = r.RenderLighting(1.0, 0, -1, True)
    got = r.ReadLightmap()
    envu = abi.Environment.from_buffer_copy(r.GetEnvironmentUniformsBytes())
    tex = oracle.make_texture(atlas, abi.SDF_UNORM16)
    dfu_default = layout.uniforms(power=0.7, min_step_size=1.0, long_step_factor=0.5, step_limit=64, max_cone_radius=24.0)
    dfu_coarse = layout.uniforms(power=1.0, min_step_size=4.0, long_step_factor=1.0, step_limit=12, max_cone_radius=8.0)
    a0 = (abi.LightVertex * len(packed["default"]))(*packed["default"]); a1 = (abi.LightVertex * len(packed["coarse"]))(*packed["coarse"])
    want0, s0 = oracle.render_sphere_lights(a0, envu, dfu_default, None, tex, tuple(env.Ambient), w, h, want_stats=True)
    want1, s1 = oracle.render_sphere_lights(a1, envu, dfu_coarse, None, tex, (0.0, 0.0, 0.0, 0.0), w, h, want_stats=True)
    assert_close(got, want0 + want1, "two quality groups")
    assert [int(x) for x in stats] == [s0.SdfSamples + s1.SdfSamples, s0.PixelLightPairs + s1.PixelLightPairs, s0.TracedPairs + s1.TracedPairs]
    assert s1.SdfSamples < s0.SdfSamples          # the coarse group really traced with fewer, longer steps


def test_light_source_replicator_expands_the_template(H, hctx, oracle):
    """LightSourceReplicator (LightSource.cs:601-620) draws its Template once per ReplicatedLight with the per-placement overrides
    (RenderReplicatorLightSource, LightingRenderer.cs:1221-1255); placements whose alpha ends up <= 0 are dropped, disabled lights are
    skipped (:1062), and SortKey orders replicators among the other lights."""
    w, h = 144, 88
    env = H.LightingEnvironment()
    env.Ambient = [0.03, 0.03, 0.05, 1.0]
    lv = scenes.random_lights(33, 14, w, h, z=(6.0, 36.0), radius=6.0, ramp=(30.0, 70.0))
    rep = H.LightSourceReplicator()
    rep.SortKey = -1                                  # ahead of the plain lights although it lives in its own list
    t = rep.Template
    t.Radius = 5.0; t.RampLength = 45.0; t.RampMode = 1; t.Opacity = 0.8; t.Color = [0.9, 0.7, 0.5, 1.0]
    t.AmbientOcclusionRadius = 10.0; t.AmbientOcclusionOpacity = 0.6; t.FalloffYFactor = 1.5
    t.SpecularColor = [0.2, 0.2, 0.3]; t.SpecularPower = 6.0
    rep.Template = t
    tkw = dict(ramp_mode=1, ao_radius=10.0, ao_opacity=0.6, falloff_y=1.5)
    packed = []
    for i in range(11):
        rl = H.ReplicatedLight()
        pos = (lv[i].LightPosition1.x, lv[i].LightPosition1.y, lv[i].LightPosition1.z)
        rl.Position = list(pos)
        radius, ramp, color, opacity, spec, power = 5.0, 45.0, (0.9, 0.7, 0.5, 1.0), 0.8, (0.2, 0.2, 0.3), 6.0
        if i % 2:
            radius = rl.Radius = 3.0 + i
        if i % 3 == 0:
            ramp = rl.RampLength = 60.0
        if i % 4 == 1:
            color = (lv[i].Color1.x, lv[i].Color1.y, lv[i].Color1.z, 0.9); rl.Color = list(color)
        if i % 5 == 2:
            spec = (0.5, 0.1, 0.1); rl.SpecularColor = list(spec); power = rl.SpecularPower = 12.0
        if i == 4:
            opacity = rl.Opacity = 0.0              # dropped: colour.W <= 0
        if i == 6:
            opacity = rl.Opacity = 0.35
        rep.Add(rl)
        if opacity > 0:
            packed.append(scenes.sphere_light(pos, radius, ramp, color=color, opacity=opacity, specular=spec, specular_power=power, **tkw))
    assert len(rep.Lights) == 11 and rep.Lights[1].Radius == 4.0 and rep.Lights[0].Radius is None and rep.Lights[0].Color is None
    env.Replicators = [rep]
    lights = []
    for i in range(11, 14):
        l = H.SphereLightSource()
        l.Position = [lv[i].LightPosition1.x, lv[i].LightPosition1.y, lv[i].LightPosition1.z]
        l.Radius = lv[i].LightProperties.x; l.RampLength = lv[i].LightProperties.y
        l.Color = [lv[i].Color1.x, lv[i].Color1.y, lv[i].Color1.z, 1.0]
        l.Enabled = i != 12
        lights.append(l)
        if l.Enabled:
            packed.append(scenes.sphere_light(tuple(l.Position), l.Radius, l.RampLength, color=tuple(l.Color)))
    env.Lights = lights
    off = H.LightSourceReplicator(); off.Enabled = False; off.Add(H.ReplicatedLight())
    env.Replicators = [rep, off]
    rc = H.RendererConfiguration(w, h)
    rc.FloatLightmap = True
    q = H.RendererQualitySettings(); q.MinStepSize = 1.0; q.LongStepFactor = 0.5; q.MaxStepCount = 64; q.MaxConeRadius = 24.0; q.OcclusionToOpacityPower = 0.7
    rc.DefaultQuality = q
    r = H.LightingRenderer(hctx, rc, env)
    field = H.DistanceField(hctx, 256, 256, 64.0, 9, 0.5)
    layout = scenes.DistanceFieldLayout(256, 256, 64.0, 9, 0.5)
    atlas = scenes.build_sdf_atlas(layout, scenes.random_obstacles(6, 10, (256, 256), 6.0, 24.0, 40.0))
    field.Load(atlas)
    r.DistanceField = field
    stats = r.RenderLighting(1.0, 0, -1, True)
    got = r.ReadLightmap()
    assert r.LastLightCount == len(packed) == 12
    dfu = abi.DistanceFieldUniforms.from_buffer_copy(r.GetDistanceFieldUniformsBytes())
    envu = abi.Environment.from_buffer_copy(r.GetEnvironmentUniformsBytes())
    arr = (abi.LightVertex * len(packed))(*packed)
    assert bytes(r.GetPackedLightVertices()) == bytes(arr)       # the vertex stream itself, byte for byte and in draw order
    want, wstats = oracle.render_sphere_lights(arr, envu, dfu, None, oracle.make_texture(atlas, abi.SDF_UNORM16), tuple(env.Ambient), w, h,
                                               want_stats=True)
    assert_close(got, want, "lightmap with a replicator")
    assert tuple(int(x) for x in stats) == (wstats.SdfSamples, wstats.PixelLightPairs, wstats.TracedPairs)
