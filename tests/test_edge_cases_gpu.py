"""Edge cases through the C ABI: empty systems and light lists, frames that do not fill whole tiles, descriptors at their limits, invalid
handles and descriptor values, non-finite particle state.  Results are the oracle's; errors are the codes include/illuminant_hip.h documents."""
import ctypes as C

import numpy as np
import pytest

from illuminant_amd import abi, native, scenes
from tests.test_lighting_gpu import render_both, small_scene
from tests.util import assert_close

pytestmark = pytest.mark.gpu

P, V, A, RC, RD = abi.PLANE_POSITION, abi.PLANE_VELOCITY, abi.PLANE_ATTRIBUTES, abi.PLANE_RENDER_COLOR, abi.PLANE_RENDER_DATA


def plain_desc(cs, mode=abi.UPDATE_POSITIONS):
    d = abi.StepDesc()
    d.FirstChunk, d.ChunkCount = 0, -1
    d.System = scenes.system_uniforms(cs)
    d.Update = abi.UpdateParams.default()
    d.UpdateMode = mode
    return d


def test_system_without_chunks(ctx):
    eng = native.Engine(ctx, 64, scenes.randomness_table(7))
    sysm = native.System(eng)
    assert sysm.chunk_count() == 0
    d = plain_desc(64)
    d.Flags = abi.STEP_COUNT_LIVE
    sysm.step(d)                                   # nothing to do is not an error
    assert len(sysm.step_counts()) == 0 and len(sysm.live_counts()) == 0
    d.FirstChunk, d.ChunkCount = 0, 1              # but naming a chunk that is not there is
    with pytest.raises(native.IlluminantError) as e:
        sysm.step(d)
    assert e.value.code == abi.ERR_OUT_OF_RANGE
    with pytest.raises(native.IlluminantError):
        sysm.download(0, P)
    sysm.close(); eng.close()


def test_all_dead_chunk_and_no_ops(ctx, oracle):
    cs = 64
    n = cs * cs
    rnd = scenes.randomness_table(7)
    eng = native.Engine(ctx, cs, rnd); sysm = native.System(eng)
    sysm.add_chunk()
    pos, vel, attr = scenes.make_particles(1, n, dead_fraction=1.0)
    for pl, a in ((P, pos), (V, vel), (A, attr)):
        sysm.upload(0, pl, a)
    d = plain_desc(cs)
    d.Flags = abi.STEP_COUNT_LIVE
    sysm.step(d)
    assert list(sysm.step_counts()) == [0]
    for pl in (P, V, RC, RD):
        assert not sysm.download(0, pl).any()
    # transforms only, no ops: the state is left exactly as it is
    pos2, vel2, attr2 = scenes.make_particles(2, n, dead_fraction=0.3)
    for pl, a in ((P, pos2), (V, vel2), (A, attr2)):
        sysm.upload(0, pl, a)
    sysm.step(plain_desc(cs, abi.UPDATE_NONE))
    assert np.array_equal(sysm.download(0, P), pos2) and np.array_equal(sysm.download(0, V), vel2)
    sysm.close(); eng.close()


def test_descriptor_at_its_limits(ctx, oracle):
    """ILM_MAX_OPS transforms and ILM_MAX_SPAWNS spawn records in one launch, 16 attractors, on a chunk size that is not a multiple of 64."""
    cs = 48
    n = cs * cs
    rnd = scenes.randomness_table(7)
    eng = native.Engine(ctx, cs, rnd); sysm = native.System(eng)
    chunks = []
    for c in range(2):
        sysm.add_chunk()
        pos, vel, attr = scenes.make_particles(20 + c, n, dead_fraction=0.5)
        for pl, a in ((P, pos), (V, vel), (A, attr)):
            sysm.upload(c, pl, a)
        chunks.append([pos.copy(), vel.copy(), attr.copy(), np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)])
    d = plain_desc(cs)
    d.OpCount = abi.MAX_OPS
    att = [((float(20 + 13 * i), float(30 + 7 * i), float(i)), 40.0 + i, 100.0 + 20 * i, i % 3) for i in range(abi.MAX_ATTRACTORS)]
    d.Ops[0].Type = abi.OP_GRAVITY; d.Ops[0].u.Gravity = scenes.gravity_params(att, 500.0)
    d.Ops[1].Type = abi.OP_NOISE
    d.Ops[1].u.Noise = scenes.noise_params(scenes.area_none(), (0.3 * 253, 0.6 * 127), (0.8 * 253, 0.1 * 127), 0.4)
    d.Ops[2].Type = abi.OP_FMA
    d.Ops[2].u.FMA = scenes.fma_params(scenes.area(2, (100, 100, 10), (60, 60, 30), falloff=20.0, rotation=0.3, strength=0.8), position_add=(1, 2, 0), velocity_multiply=(0.9, 0.9, 1))
    d.Ops[3].Type = abi.OP_GRAVITY; d.Ops[3].u.Gravity = scenes.gravity_params(att[:1], 10.0)
    d.SpawnCount = abi.MAX_SPAWNS
    for k in range(abi.MAX_SPAWNS):
        d.Spawns[k].ChunkIndex = k
        d.Spawns[k].Params = scenes.spawn_params(cs, 100 * k, 100 * k + 300, 7 * k, (0.1 * 253 * (k + 1), 0.2 * 127))
    d.Flags = abi.STEP_COUNT_LIVE
    sysm.step(d)
    want_counts = oracle.step(chunks, cs, rnd, d, want_counts=True)
    assert np.array_equal(sysm.step_counts(), want_counts)
    for c in range(2):
        for k, pl in enumerate((P, V, A, RC, RD)):
            got = sysm.download(c, pl)
            if k == 0:
                assert np.array_equal(got[:, 3] > 0, chunks[c][0][:, 3] > 0)
            assert_close(got, chunks[c][k], "chunk %d plane %d" % (c, k))
    # one more than the limits is refused
    d.OpCount = abi.MAX_OPS + 1
    with pytest.raises(native.IlluminantError):
        sysm.step(d)
    d.OpCount = 1
    d.SpawnCount = abi.MAX_SPAWNS + 1
    with pytest.raises(native.IlluminantError):
        sysm.step(d)
    sysm.close(); eng.close()


def test_non_finite_particle_state_follows_the_oracle(ctx, oracle):
    cs = 64
    n = cs * cs
    rnd = scenes.randomness_table(7)
    pos, vel, attr = scenes.make_particles(5, n, dead_fraction=0.2)
    pos[::97, 0] = np.nan; pos[5::89, 1] = np.inf; vel[7::83, 2] = -np.inf; vel[11::79, 0] = np.nan; pos[13::71, 3] = np.nan
    eng = native.Engine(ctx, cs, rnd); sysm = native.System(eng)
    sysm.add_chunk()
    for pl, a in ((P, pos), (V, vel), (A, attr)):
        sysm.upload(0, pl, a)
    d = plain_desc(cs)
    d.OpCount = 1
    d.Ops[0].Type = abi.OP_GRAVITY
    d.Ops[0].u.Gravity = scenes.gravity_params([((128., 128., 0.), 150., 60., 1)], 8.0)
    d.Flags = abi.STEP_COUNT_LIVE
    sysm.step(d)
    chunk = [pos.copy(), vel.copy(), attr.copy(), np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)]
    want_counts = oracle.step([chunk], cs, rnd, d, want_counts=True)
    got_p = sysm.download(0, P)
    # liveness (life > 0, false for NaN) is bit-exact even for the poisoned slots; NaNs sit in the same places
    assert np.array_equal(sysm.step_counts(), want_counts)
    assert np.array_equal(got_p[:, 3] > 0, chunk[0][:, 3] > 0)
    assert np.array_equal(np.isnan(got_p), np.isnan(chunk[0]))
    inf = np.isinf(chunk[0])
    assert np.array_equal(np.isinf(got_p), inf) and np.array_equal(got_p[inf], chunk[0][inf])
    fin = np.isfinite(chunk[0])
    assert_close(np.where(fin, got_p, 0.0), np.where(fin, chunk[0], 0.0), "position")
    sysm.close(); eng.close()


@pytest.mark.parametrize("width,height", [(1, 1), (17, 5), (161, 113), (16, 16)])
def test_frames_that_do_not_fill_whole_tiles(ctx, oracle, width, height):
    layout, atlas, dfu, lights, _, _ = small_scene(n_lights=6)
    lights = scenes.random_lights(9, 6, max(width, 64), max(height, 64), z=(8.0, 48.0), radius=10.0, ramp=(40.0, 120.0))
    env = scenes.environment()
    got, want, stats, ostats = render_both(ctx, oracle, lights, env, dfu, None, 0, atlas, abi.SDF_UNORM16, (0.1, 0.1, 0.1, 1.0), width, height)
    assert (stats.SdfSamples, stats.PixelLightPairs, stats.TracedPairs) == (ostats.SdfSamples, ostats.PixelLightPairs, ostats.TracedPairs)
    assert got.shape == want.shape == (height, width, 4)
    assert_close(got, want, "lightmap %dx%d" % (width, height))


def test_no_lights_and_lights_off_screen(ctx, oracle):
    layout, atlas, dfu, lights, w, h = small_scene(n_lights=4)
    env = scenes.environment()
    ambient = (0.2, 0.3, 0.4, 1.0)
    empty = (abi.LightVertex * 0)()
    got, want, stats, ostats = render_both(ctx, oracle, empty, env, dfu, None, 0, atlas, abi.SDF_UNORM16, ambient, w, h)
    assert stats.PixelLightPairs == ostats.PixelLightPairs == 0 and stats.SdfSamples == 0
    assert np.array_equal(got, want) and np.allclose(got, np.asarray(ambient, np.float32))
    far = scenes.random_lights(3, 5, w, h, z=(8.0, 48.0), radius=10.0, ramp=(40.0, 80.0))
    for l in far:
        l.LightPosition1.x += 5000.0; l.LightPosition2.x += 5000.0; l.LightPosition3.x += 5000.0
    got, want, stats, ostats = render_both(ctx, oracle, far, env, dfu, None, 0, atlas, abi.SDF_UNORM16, ambient, w, h)
    assert stats.PixelLightPairs == ostats.PixelLightPairs == 0
    assert np.array_equal(got, want)


def test_invalid_handles_and_arguments_are_reported(ctx):
    lib = native.lib()
    bogus = abi.Handle(0x1234)
    out = abi.Handle(0)
    assert lib.ilm_system_create(bogus, C.byref(out)) == abi.ERR_INVALID_HANDLE
    assert lib.ilm_system_step(bogus, None) == abi.ERR_INVALID_HANDLE
    assert lib.ilm_ctx_sync(bogus) == abi.ERR_INVALID_HANDLE
    assert b"handle" in lib.ilm_last_error()
    eng = native.Engine(ctx, 16, scenes.randomness_table(7))
    sysm = native.System(eng)
    assert lib.ilm_system_step(sysm.handle, None) == abi.ERR_INVALID_ARGUMENT
    # a system handle where an engine handle is expected
    assert lib.ilm_system_create(sysm.handle, C.byref(out)) == abi.ERR_INVALID_HANDLE
    sysm.add_chunk()
    d = plain_desc(16)
    d.UpdateMode = 17
    with pytest.raises(native.IlluminantError) as e:
        sysm.step(d)
    assert e.value.code == abi.ERR_INVALID_ARGUMENT
    d = plain_desc(16)
    d.OpCount = 1
    d.Ops[0].Type = 99
    with pytest.raises(native.IlluminantError):
        sysm.step(d)
    with pytest.raises(native.IlluminantError):
        sysm.upload(0, P, np.zeros((16 * 16 + 1, 4), np.float32))       # more slots than the chunk has
    with pytest.raises(native.IlluminantError):
        native.Engine(ctx, 0, scenes.randomness_table(7))
    # parents outlive children: nothing is destroyed under a live object
    assert lib.ilm_engine_destroy(eng.handle) == abi.ERR_STATE
    assert lib.ilm_ctx_destroy(ctx.handle) == abi.ERR_STATE
    h_sys = abi.Handle(sysm.handle.value)
    sysm.close(); eng.close()
    # a destroyed handle is dead
    assert lib.ilm_system_step(h_sys, None) == abi.ERR_INVALID_HANDLE


def test_bound_distance_field_may_be_destroyed_first(ctx):
    """The system keeps the field's handle, not its address: stepping with a destroyed field is an error, not a crash."""
    from tests.test_particles_gpu import cfg1_field
    eng = native.Engine(ctx, 16, scenes.randomness_table(7))
    sysm = native.System(eng); sysm.add_chunk()
    layout, atlas, dfu = cfg1_field()
    sdf = native.DistanceFieldTexture(ctx, atlas, abi.SDF_UNORM16)
    sysm.set_distance_field(sdf)
    d = plain_desc(16, abi.UPDATE_WITH_DISTANCE_FIELD)
    d.DistanceField = dfu
    sysm.step(d)
    sdf.close()
    with pytest.raises(native.IlluminantError) as e:
        sysm.step(d)
    assert e.value.code == abi.ERR_STATE
    sysm.close(); eng.close()


@pytest.mark.parametrize("resolution", [1.0, 0.125])
@pytest.mark.parametrize("sfmt", [abi.SDF_UNORM16, abi.SDF_FP16])
def test_traces_along_the_fields_last_column_and_outside_it(ctx, oracle, sfmt, resolution):
    """The frame covers the field exactly and lights sit on its right / bottom edge, above its top slice and outside it: traces run
    along the atlas' last texel column (the right-hand tap wraps to column 0 -- the paired tap loads fall back to single ones), leave
    the volume (distance-to-volume term, general trace loop) and mix with in-volume traces in one wave.  Both tap-load kernel variants
    (texel density 1 and 1/8), both formats; integer statistics exact."""
    w, h = 128, 96
    layout = scenes.DistanceFieldLayout(w, h, 64.0, 9, resolution, 128)
    obstacles = scenes.random_obstacles(31, 10, (w, h), size_lo=6.0, size_hi=20.0, z_hi=40.0)
    atlas = scenes.build_sdf_atlas(layout, obstacles, fmt=sfmt)
    dfu = layout.uniforms(max_cone_radius=16.0, power=0.7, step_limit=48, min_step_size=1.0, long_step_factor=0.5)
    spots = [(127.9, 48.0, 30.0), (127.6, 5.0, 41.0), (64.0, 95.8, 30.0), (126.0, 94.0, 62.5), (140.0, 40.0, 20.0), (-9.0, 70.0, 12.0),
             (60.0, 40.0, 80.0), (0.2, 0.3, 25.0), (64.0, 48.0, 10.0)]
    lights = (abi.LightVertex * len(spots))(*[scenes.sphere_light(p, 6.0, 70.0 + 9.0 * i, color=(0.9, 0.8 - 0.05 * i, 0.5, 1.0)) for i, p in enumerate(spots)])
    env = scenes.environment()
    got, want, stats, ostats = render_both(ctx, oracle, lights, env, dfu, None, 0, atlas, sfmt, (0.02, 0.02, 0.02, 1.0), w, h)
    assert (stats.SdfSamples, stats.PixelLightPairs, stats.TracedPairs) == (ostats.SdfSamples, ostats.PixelLightPairs, ostats.TracedPairs)
    assert ostats.TracedPairs > 0.5 * ostats.PixelLightPairs > 0
    assert_close(got, want, "lightmap with traces on and beyond the field's edges")


def reviving_noise():
    """A Noise op whose life delta is not zero (PositionScale.w != 0): it has no life check (Noise.fx:40), so it gives dead -- and
    never-written -- slots a life."""
    return scenes.noise_params(scenes.area_none(), (0.3 * 253, 0.6 * 127), (0.8 * 253, 0.1 * 127), 0.4, cycles_per_second=10.0,
                               position=((0.5,) * 4, (0,) * 4, (1.0, 1.0, 0.0, 4.0)))


@pytest.mark.parametrize("cs", [10, 48])
def test_stride_padding_is_not_a_particle(ctx, oracle, cs):
    """Chunk sizes whose square is not a multiple of 64 / 1024 leave padding lanes behind the last slot (the plane stride is rounded
    up).  A reviving Noise runs on every lane it is given: the padding must not come to life, be updated or be counted -- the
    reference counts ChunkSize^2 pixels."""
    n = cs * cs
    rnd = scenes.randomness_table(7)
    eng = native.Engine(ctx, cs, rnd); sysm = native.System(eng)
    sysm.add_chunk()
    chunk = [np.zeros((n, 4), np.float32) for _ in range(5)]      # an empty chunk: every slot dead
    d = plain_desc(cs)
    d.OpCount = 1
    d.Ops[0].Type = abi.OP_NOISE; d.Ops[0].u.Noise = reviving_noise()
    d.Flags = abi.STEP_COUNT_LIVE
    for _ in range(2):
        sysm.step(d)
        want = oracle.step([chunk], cs, rnd, d, want_counts=True)
        assert np.array_equal(sysm.step_counts(), want)
        assert np.array_equal(sysm.live_counts(), want)
    assert 0 < int(want[0]) <= n, "the noise must have revived slots (and no more than the chunk has)"
    got = sysm.download(0, P)
    assert np.array_equal(got[:, 3] > 0, chunk[0][:, 3] > 0)
    assert_close(got, chunk[0], "revived particles", life_exact=True)
    sysm.close(); eng.close()


def test_slots_revived_past_the_high_water_mark_stay_in_the_step(ctx, oracle):
    """A spawn-target chunk with a few spawned slots is skipped past its high-water mark -- until a reviving Noise has written every
    slot.  The plain Update that follows must age, move and count the revived slots like the reference."""
    cs = 64
    n = cs * cs
    rnd = scenes.randomness_table(7)
    eng = native.Engine(ctx, cs, rnd); sysm = native.System(eng)
    sysm.add_chunk()
    chunk = [np.zeros((n, 4), np.float32) for _ in range(5)]
    d = plain_desc(cs)
    d.SpawnCount = 1
    d.Spawns[0].ChunkIndex = 0
    d.Spawns[0].Params = scenes.spawn_params(cs, 0, 99, 0, (0.1 * 253, 0.2 * 127))          # 100 of 4096 slots spawned
    d.Flags = abi.STEP_COUNT_LIVE
    sysm.step(d); oracle.step([chunk], cs, rnd, d)
    # single-pass reviving noise over the chunk (UpdateMode none): touches dead and never-written slots
    sysm.noise(0, scenes.system_uniforms(cs), reviving_noise())
    oracle.noise(chunk[0], chunk[1], cs, rnd, scenes.system_uniforms(cs), reviving_noise())
    revived = int((chunk[0][:, 3] > 0).sum())
    assert revived > 1000, revived
    # a plain step without the noise op
    d2 = plain_desc(cs)
    d2.Flags = abi.STEP_COUNT_LIVE
    for _ in range(2):
        sysm.step(d2)
        want = oracle.step([chunk], cs, rnd, d2, want_counts=True)
        assert np.array_equal(sysm.step_counts(), want), "revived slots past the high-water mark were skipped"
    for k, pl in enumerate((P, V, RC, RD)):
        assert_close(sysm.download(0, pl), chunk[(0, 1, 3, 4)[k]], "plane %d after the plain steps" % pl, life_exact=(k == 0))
    sysm.close(); eng.close()


def test_chunks_come_from_the_engines_pool_zero_filled(ctx):
    """r05: chunks are carved out of slabs and recycled through the engine's pool (the reference keeps its released buffer sets for reuse,
    ParticleEngine.cs:145-170,402-419).  A recycled chunk must be indistinguishable from a new one -- every plane zero --, chunks of one
    slab must not overlap, and systems of one engine share the pool."""
    import numpy as np
    from illuminant_amd import abi, native, scenes
    cs = 32
    n = cs * cs
    eng = native.Engine(ctx, cs, scenes.randomness_table(7))
    a, b = native.System(eng), native.System(eng)
    marks = {}
    for k in range(11):                      # more than one slab of 8
        s = a if k % 2 == 0 else b
        idx = s.add_chunk()
        for plane in range(5):
            assert not s.download(s.chunk_count() - 1, plane).any(), "a fresh chunk is not zero"
        fill = np.full((n, 4), float(k + 1), np.float32)
        for plane in (abi.PLANE_POSITION, abi.PLANE_VELOCITY, abi.PLANE_ATTRIBUTES, abi.PLANE_RENDER_COLOR, abi.PLANE_RENDER_DATA):
            s.upload(s.chunk_count() - 1, plane, fill + plane)
        marks[(id(s), s.chunk_count() - 1)] = float(k + 1)
    for (sid, c), v in marks.items():        # nobody's chunk was overwritten by a neighbour's upload
        s = a if sid == id(a) else b
        for plane in range(5):
            assert (s.download(c, plane) == v + plane).all()
    # recycle: remove three chunks of a, add five to b -- the first three come back from the pool
    for _ in range(3):
        a.remove_chunk(0)
    for _ in range(5):
        b.add_chunk()
        for plane in range(5):
            assert not b.download(b.chunk_count() - 1, plane).any(), "a recycled chunk is not zero"
    a.close()                                # its chunks go back too
    c2 = native.System(eng)
    for _ in range(4):
        c2.add_chunk()
        assert not c2.download(c2.chunk_count() - 1, abi.PLANE_POSITION).any()
    c2.close(); b.close(); eng.close()


def test_the_chunk_pool_gives_memory_back_beyond_its_spares(ctx):
    """ParticleEngine keeps at most SpareBufferCount = 20 discarded buffers (ParticleEngine.cs:402-419).  The pool carves chunks from slabs
    of up to eight: r05 only ever freed one-chunk slabs, so destroying a large system pinned all of its memory behind a live engine
    (ADVICE r05).  Remove 40 chunks of 256^2 (5 slabs of 42 MB): the device's free memory must come back but for at most the 20 spares
    and the slabs that still hold a chunk in use."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()

    def free_bytes():
        ctx.sync()
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return int(free.value)
    cs = 256
    eng = native.Engine(ctx, cs, scenes.randomness_table(3))
    sysm = native.System(eng)
    before = free_bytes()
    for _ in range(41):
        sysm.add_chunk()
    chunk_bytes = (before - free_bytes()) / 48.0          # 41 chunks come out of 6 slabs of 8
    assert 5.0e6 < chunk_bytes < 5.6e6                    # 20 planes x (65 536 + padding) floats
    for _ in range(40):
        sysm.remove_chunk(sysm.chunk_count() - 1)
    held = before - free_bytes()
    # one slab holds the chunk still in use (8 chunks), the spares are at most 20 chunks + the slab granularity
    assert held <= (8 + 20 + 8) * chunk_bytes * 1.02, (held / chunk_bytes)
    assert held >= 8 * chunk_bytes * 0.98
    # the spares are reused: the same chunks come back without a new allocation
    mid = free_bytes()
    for _ in range(16):
        sysm.add_chunk()
    assert free_bytes() >= mid - 8 * chunk_bytes * 1.02
    sysm.close(); eng.close()
