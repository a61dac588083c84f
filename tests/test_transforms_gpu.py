"""Parity of the extended step-kernel variant (MatrixMultiply, SpatialNoise, position-buffer and feedback spawners -- SURVEY 8f-2)
against the CPU oracle, through the C ABI."""
import numpy as np
import pytest

from illuminant_amd import abi, native, scenes
from tests import transforms_common as tc
from tests.util import assert_close

pytestmark = pytest.mark.gpu

P, V, A, RC, RD = abi.PLANE_POSITION, abi.PLANE_VELOCITY, abi.PLANE_ATTRIBUTES, abi.PLANE_RENDER_COLOR, abi.PLANE_RENDER_DATA


@pytest.mark.parametrize("index", range(len(tc.load_cases())))
def test_closed_form_case(ctx, index):
    tc.check_case(tc.load_cases()[index], tc.GpuBackend(ctx))


def random_matrix(seed, scale=0.2, perspective=False):
    m = np.eye(4, dtype=np.float32) + scenes.uniform(seed, (4, 4), -scale, scale)
    if not perspective:
        m[:, 3] = [0, 0, 0, 1]
    return abi.Matrix.from_rows(m.tolist())


def run_both(ctx, oracle, cs, n_chunks, desc, seed=3, dead_fraction=0.2, spawn_positions=None, spawn_pattern=None):
    rnd = scenes.randomness_table(seed)
    eng = native.Engine(ctx, cs, rnd)
    sysm = native.System(eng)
    n = cs * cs
    chunks = []
    for c in range(n_chunks):
        sysm.add_chunk()
        pos, vel, attr = scenes.make_particles(100 + c, n, dead_fraction=dead_fraction)
        vel[:, 3] = np.floor(scenes.uniform(200 + c, (n,), 0, 4))     # categories 0..3
        for plane, data in ((P, pos), (V, vel), (A, attr)):
            sysm.upload(c, plane, data)
        chunks.append([pos.copy(), vel.copy(), attr.copy(), np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)])
    if spawn_positions is not None:
        sysm.set_spawn_positions(0, spawn_positions)
    if spawn_pattern is not None:
        sysm.set_spawn_pattern(0, spawn_pattern)
    desc.Flags = abi.STEP_COUNT_LIVE
    sysm.step(desc)
    want_counts = oracle.step(chunks, cs, rnd, desc, want_counts=True, spawn_positions={0: spawn_positions} if spawn_positions is not None else None,
                              spawn_patterns={0: spawn_pattern} if spawn_pattern is not None else None)
    got_counts = sysm.step_counts()
    got = [[sysm.download(c, pl) for pl in (P, V, A, RC, RD)] for c in range(n_chunks)]
    sysm.close(); eng.close()
    return got, chunks, got_counts, want_counts


def compare(got, want, got_counts, want_counts, planes=(0, 1, 2, 3, 4)):
    assert np.array_equal(got_counts, want_counts)
    for c in range(len(got)):
        # liveness (life > 0) is bit-exact, then the floats within the north-star tolerance
        assert np.array_equal(got[c][0][:, 3] > 0, want[c][0][:, 3] > 0)
        for k in planes:
            assert_close(got[c][k], want[c][k], "chunk %d plane %d" % (c, k))


@pytest.mark.parametrize("area_type", [0, 1, 2, 3, 4, 5])
def test_matrix_multiply_matches_oracle(ctx, oracle, area_type):
    cs = 64
    d = abi.StepDesc()
    d.FirstChunk, d.ChunkCount = 0, -1
    d.System = scenes.system_uniforms(cs, friction=0.05, max_velocity=500.0, life_decay=1.0)
    d.Update = abi.UpdateParams.default()
    d.UpdateMode = abi.UPDATE_POSITIONS
    d.OpCount = 1
    d.Ops[0].Type = abi.OP_MATRIX_MULTIPLY
    ar = scenes.area_none(0.8, (1.0, 3.0)) if area_type == 0 else scenes.area(area_type, (128, 128, 8), (70, 50, 30), falloff=40.0, rotation=0.3,
                                                                                strength=0.9, category_filter=(0.0, 2.0))
    d.Ops[0].u.MatrixMultiply = scenes.matrix_multiply_params(ar, random_matrix(5, perspective=(area_type % 2 == 1)), random_matrix(6),
                                                              None if area_type == 3 else 10.0)
    got, want, gc, wc = run_both(ctx, oracle, cs, 2, d)
    compare(got, want, gc, wc)


@pytest.mark.parametrize("replace", [True, False])
def test_spatial_noise_matches_oracle(ctx, oracle, replace):
    cs = 64
    d = abi.StepDesc()
    d.FirstChunk, d.ChunkCount = 0, -1
    d.System = scenes.system_uniforms(cs, friction=0.05, max_velocity=500.0, life_decay=1.0)
    d.Update = abi.UpdateParams.default()
    d.UpdateMode = abi.UPDATE_POSITIONS
    d.OpCount = 1
    d.Ops[0].Type = abi.OP_SPATIAL_NOISE
    noise = scenes.noise_params(scenes.area(1, (128, 128, 8), (90, 70, 30), falloff=60.0, strength=0.9), (0.37 * 253, 0.81 * 127),
                                (0.12 * 253, 0.55 * 127), 0.35, 10.0, replace, position=((-0.5,) * 4, (0,) * 4, (3.0, 2.0, 1.0, 0.0)),
                                velocity=((-0.5,) * 3, (0,) * 3, (40.0, 30.0, 5.0)), speed=(-0.5, 0.0, 6.0))
    d.Ops[0].u.SpatialNoise = scenes.spatial_noise_params(noise, (13.0, 7.0))
    got, want, gc, wc = run_both(ctx, oracle, cs, 2, d)
    compare(got, want, gc, wc)


def test_spatial_noise_that_changes_life_and_single_pass_entry(ctx, oracle):
    """PositionScale.w != 0: the op can revive / kill slots (no life check, Noise.fx:86), through the single-pass entry point."""
    cs = 32
    rnd = scenes.randomness_table(4)
    eng = native.Engine(ctx, cs, rnd)
    sysm = native.System(eng)
    sysm.add_chunk()
    n = cs * cs
    pos, vel, attr = scenes.make_particles(9, n, dead_fraction=0.5)
    for plane, data in ((P, pos), (V, vel), (A, attr)):
        sysm.upload(0, plane, data)
    sysu = scenes.system_uniforms(cs)
    noise = scenes.noise_params(scenes.area_none(1.0), (10.0, 20.0), (30.0, 40.0), 0.5, 10.0, True,
                                position=((-0.5,) * 4, (0,) * 4, (1.0, 1.0, 1.0, 4.0)))
    sp = scenes.spatial_noise_params(noise, (3.0, 3.0))
    sysm.spatial_noise(0, sysu, sp)
    opos, ovel = pos.copy(), vel.copy()
    oracle.spatial_noise(opos, ovel, cs, rnd, sysu, sp)
    gpos, gvel = sysm.download(0, P), sysm.download(0, V)
    assert np.array_equal(gpos[:, 3] > 0, opos[:, 3] > 0)
    assert ((opos[:, 3] > 0) != (pos[:, 3] > 0)).any()
    assert_close(gpos, opos, "position")
    live = opos[:, 3] > 0
    assert_close(gvel[live], ovel[live], "velocity")      # dead slots with v == 0 hold normalize(0) = NaN on both sides
    # MatrixMultiply single-pass entry
    mm = scenes.matrix_multiply_params(scenes.area_none(0.5), random_matrix(1), random_matrix(2))
    sysm.matrix_multiply(0, sysu, mm)
    oracle.matrix_multiply(opos, ovel, cs, sysu, mm)
    assert_close(sysm.download(0, P), opos, "position after matrix multiply")
    sysm.close(); eng.close()


def test_all_five_transform_types_fused(ctx, oracle):
    """Gravity + SpatialNoise + MatrixMultiply + FMA in one launch, then UpdatePositions: pass order preserved per slot."""
    cs = 64
    d = abi.StepDesc()
    d.FirstChunk, d.ChunkCount = 0, -1
    d.System = scenes.system_uniforms(cs, friction=0.1, max_velocity=800.0, life_decay=1.2)
    d.Update = abi.UpdateParams.default()
    d.UpdateMode = abi.UPDATE_POSITIONS
    d.OpCount = 4
    d.Ops[0].Type = abi.OP_GRAVITY
    d.Ops[0].u.Gravity = scenes.gravity_params([((128.0, 128.0, 0.0), 150.0, 60.0, 1)], 8.0)
    d.Ops[1].Type = abi.OP_SPATIAL_NOISE
    d.Ops[1].u.SpatialNoise = scenes.spatial_noise_params(scenes.noise_params(scenes.area_none(1.0), (5.0, 6.0), (7.0, 8.0), 0.25), (4.0, 4.0))
    d.Ops[2].Type = abi.OP_MATRIX_MULTIPLY
    d.Ops[2].u.MatrixMultiply = scenes.matrix_multiply_params(scenes.area_none(0.7), random_matrix(11), random_matrix(12))
    d.Ops[3].Type = abi.OP_FMA
    d.Ops[3].u.FMA = scenes.fma_params(scenes.area_none(1.0), velocity_multiply=(0.9, 0.9, 1.0))
    got, want, gc, wc = run_both(ctx, oracle, cs, 3, d)
    compare(got, want, gc, wc)


def test_position_buffer_spawner_matches_oracle(ctx, oracle):
    cs = 64
    positions = [(40.0 + 30.0 * i, 20.0 + 11.0 * i, float(i)) for i in range(9)]
    for (rate, loop) in ((None, True), (3.0, True), (2.5, False)):
        p, buf = scenes.position_buffer_spawn_params(cs, 700, 1500, 4321, (0.42 * 253, 0.77 * 127), positions, life_constant=3.3,
                                                     position=((0, 0, 0), (9, 5, 2), (0, 0, 0), scenes.FORMULA_SPHERICAL),
                                                     velocity=((1, 2, 3), (60, 60, 60), (0, 0, 0), scenes.FORMULA_SPHERICAL),
                                                     life=(3.3, 2.7, 0.0), polygon_rate=rate, polygon_loop=loop, polygon_speed=(5.0, 2.0, 0.0))
        d = abi.StepDesc()
        d.FirstChunk, d.ChunkCount = 0, -1
        d.System = scenes.system_uniforms(cs)
        d.Update = abi.UpdateParams.default()
        d.UpdateMode = abi.UPDATE_POSITIONS
        d.SpawnCount = 1
        d.Spawns[0].ChunkIndex = 1
        d.Spawns[0].Kind = abi.SPAWN_POSITION_BUFFER
        d.Spawns[0].Params = p
        got, want, gc, wc = run_both(ctx, oracle, cs, 2, d, dead_fraction=0.6, spawn_positions=buf)
        compare(got, want, gc, wc)


@pytest.mark.parametrize("divisor,top_left,size_px,multiply", [(1, None, None, True), (2, None, None, False), (3, (5, 3), (30, 18), True),
                                                               (4, None, None, True)])
def test_pattern_spawner_matches_oracle(ctx, oracle, divisor, top_left, size_px, multiply):
    """PatternSpawner.fx:21-97 on a 37 x 23 texture with a full mip chain: whole-instance and single-row spawns, every Divisor's mip
    level, a sub-rectangle, spherical position / velocity formulas, the alpha discard."""
    cs = 64
    tw, th = 37, 23
    texels = scenes.uniform(900, (th, tw, 4), 0.0, 1.0)
    texels[::5, ::7, 3] = 0.0          # some transparent pixels: rejected by the attribute discard threshold
    levels = scenes.pattern_mip_chain(texels)
    per_row, rows = scenes.pattern_counts(tw, th, divisor, top_left, size_px)
    for current_row, count in ((0, per_row * rows), (min(2, rows - 1), per_row)):
        first = 300
        p = scenes.spawn_params(cs, first, first + count - 1, 0, (0.31 * 253, 0.58 * 127),
                                position=((500, 400, 2), (3, 3, 1), (0, 0, 0), scenes.FORMULA_SPHERICAL),
                                velocity=((1, 2, 3), (40, 40, 40), (0, 0, 0), scenes.FORMULA_SPHERICAL), life=(2.5, 1.5, 0.0),
                                color=((0.9, 0.8, 0.7, 1.0) if multiply else (0.05, 0.0, 0.1, 0.0), (0.1, 0.1, 0.1, 0.0), (0, 0, 0, 0)),
                                alpha_discard_threshold=8.0)
        d = abi.StepDesc()
        d.FirstChunk, d.ChunkCount = 0, -1
        d.System = scenes.system_uniforms(cs)
        d.Update = abi.UpdateParams.default()
        d.UpdateMode = abi.UPDATE_POSITIONS
        d.SpawnCount = 1
        d.Spawns[0].ChunkIndex = 1
        d.Spawns[0].Kind = abi.SPAWN_PATTERN
        d.Spawns[0].Params = p
        d.Spawns[0].Pattern = scenes.pattern_params(tw, th, divisor, current_row, top_left, size_px, multiply_color_constant=multiply)
        got, want, gc, wc = run_both(ctx, oracle, cs, 2, d, dead_fraction=1.0, spawn_pattern=levels)
        compare(got, want, gc, wc)
        assert 0 < wc[1] <= count      # something spawned; transparent / out-of-texture particles did not


def test_pattern_spawner_requires_a_texture(ctx):
    cs = 16
    eng = native.Engine(ctx, cs, scenes.randomness_table(7))
    sysm = native.System(eng)
    sysm.add_chunk()
    d = tc.base_desc(1 / 60)
    d.SpawnCount = 1
    d.Spawns[0].Kind = abi.SPAWN_PATTERN
    d.Spawns[0].Params = scenes.spawn_params(cs, 0, 31, 0, (1.0, 2.0))
    d.Spawns[0].Pattern = scenes.pattern_params(8, 4)
    with pytest.raises(native.IlluminantError) as e:
        sysm.step(d)
    assert e.value.code == abi.ERR_STATE
    sysm.set_spawn_pattern(0, scenes.pattern_mip_chain(np.ones((4, 8, 4), np.float32), levels=1))
    sysm.step(d)
    sysm.set_spawn_pattern(0, None)
    with pytest.raises(native.IlluminantError):
        sysm.step(d)
    sysm.close(); eng.close()


def test_feedback_spawner_matches_oracle(ctx, oracle):
    cs = 64
    n = cs * cs
    rnd = scenes.randomness_table(7)
    eng = native.Engine(ctx, cs, rnd)
    src = native.System(eng)
    dst = native.System(eng)
    src.add_chunk(); src.add_chunk(); dst.add_chunk()
    spos, svel, sattr = scenes.make_particles(50, n, dead_fraction=0.3)
    for plane, data in ((P, spos), (V, svel), (A, sattr)):
        src.upload(1, plane, data)
    dpos, dvel, dattr = scenes.make_particles(51, n, dead_fraction=0.7)
    for plane, data in ((P, dpos), (V, dvel), (A, dattr)):
        dst.upload(0, plane, data)
    p = scenes.spawn_params(cs, 256, 256 + 899, 0, (0.15 * 253, 0.66 * 127),
                            position=((1, 2, 3), (4, 4, 4), (0, 0, 0), scenes.FORMULA_SPHERICAL),
                            velocity=((0, 0, 0), (20, 20, 20), (0, 0, 0), scenes.FORMULA_TOWARDS), life=(1.5, 1.0, 0.0),
                            color=((0.5, 0.6, 0.7, 1.0), (0.1, 0.1, 0.1, 0.0), (0, 0, 0, 0)))
    d = abi.StepDesc()
    d.FirstChunk, d.ChunkCount = 0, -1
    d.System = scenes.system_uniforms(cs)
    d.Update = abi.UpdateParams.default()
    d.UpdateMode = abi.UPDATE_POSITIONS
    d.SpawnCount = 1
    d.Spawns[0].ChunkIndex = 0
    d.Spawns[0].Kind = abi.SPAWN_FEEDBACK
    d.Spawns[0].Params = p
    d.Spawns[0].Feedback = scenes.feedback_params(src.handle.value, 1, 1000, instance_multiplier=3, source_velocity_factor=0.25,
                                                  multiply_life=True, multiply_color_constant=True, source_life_range=(0.5, 5.0))
    d.Flags = abi.STEP_COUNT_LIVE
    dst.step(d)
    chunk = [dpos.copy(), dvel.copy(), dattr.copy(), np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)]
    want_counts = oracle.step([chunk], cs, rnd, d, want_counts=True, feedback_sources={0: (spos, svel, sattr)})
    got = [[dst.download(0, pl) for pl in (P, V, A, RC, RD)]]
    compare(got, [chunk], dst.step_counts(), want_counts)
    # the source system is untouched
    assert np.array_equal(src.download(1, P), spos)
    # reference error behaviour: a system cannot feed back into itself (SpecialSpawners.cs:347-349)
    d.Spawns[0].Feedback.SourceSystem = dst.handle.value
    d.Spawns[0].Feedback.SourceChunkIndex = 0
    with pytest.raises(native.IlluminantError) as e:
        dst.step(d)
    assert e.value.code == abi.ERR_INVALID_ARGUMENT
    for x in (src, dst, eng):
        x.close()


def test_position_buffer_requires_a_bound_list(ctx):
    cs = 16
    eng = native.Engine(ctx, cs, scenes.randomness_table(7))
    sysm = native.System(eng)
    sysm.add_chunk()
    p, buf = scenes.position_buffer_spawn_params(cs, 0, 9, 0, (1.0, 2.0), [(i, i, i) for i in range(6)])
    d = tc.base_desc(1 / 60)
    d.SpawnCount = 1
    d.Spawns[0].Kind = abi.SPAWN_POSITION_BUFFER
    d.Spawns[0].Params = p
    with pytest.raises(native.IlluminantError) as e:
        sysm.step(d)
    assert e.value.code == abi.ERR_STATE
    sysm.set_spawn_positions(0, buf)
    sysm.step(d)
    sysm.close(); eng.close()
