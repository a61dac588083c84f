"""ilm_system_step has two kernels (csrc/particles.hip): the one that interprets an IlmStepDesc and the one specialised for the common
shape of a step (power-of-two chunk size >= 64, UpdatePositions, Gravity / area-less Noise / area-less FMA, inline spawners).
Which one runs is a launch decision, so the same step must give the same bits through both (ilm_debug_step_interpreter forces the
interpreter) -- and both are held against the oracle.  Also covered: the curves' host-coded decisions (bezier.hpp) for every
count class / range mode / shaping mode, and the live counts the kernels publish into host memory.
"""
import numpy as np
import pytest

from illuminant_amd import abi, native, scenes
from tests.util import assert_bits_equal, assert_close

pytestmark = pytest.mark.gpu

P, V, A, RC, RD = abi.PLANE_POSITION, abi.PLANE_VELOCITY, abi.PLANE_ATTRIBUTES, abi.PLANE_RENDER_COLOR, abi.PLANE_RENDER_DATA
PLANES = (P, V, A, RC, RD)


def _bezier4(count, mode, lo=0.0, hi=3.0, negative=False):
    inv = 1.0 / (hi - lo)
    return abi.ClampedBezier4(abi.f4(lo, -inv if negative else inv, count, mode),
                              abi.f4(1.0, 0.2, 0.1, 1.0), abi.f4(0.4, 0.9, 0.3, 0.8), abi.f4(0.1, 0.5, 1.0, 0.5), abi.f4(0.9, 0.1, 0.6, 0.2))


def _bezier1(count, mode, lo=0.0, hi=50.0, negative=False):
    inv = 1.0 / (hi - lo)
    return abi.ClampedBezier1(abi.f4(lo, -inv if negative else inv, count, mode), abi.f4(0.5, 2.0, 1.25, 3.0))


def _step(cs, shape):
    d = abi.StepDesc()
    d.FirstChunk, d.ChunkCount = 0, -1
    d.System = scenes.system_uniforms(cs, friction=0.05, max_velocity=900.0, life_decay=4.0, rotation_from_velocity=shape.get("rotation", False))
    d.Update = abi.UpdateParams.default()
    if "curves" in shape:
        count, mode, negative = shape["curves"]
        d.Update.ColorFromLife = _bezier4(count, mode, 0.0, 2.5, negative)
        d.Update.ColorFromVelocity = _bezier4(max(1, 5 - count), mode, 0.0, 120.0, not negative)
        d.Update.SizeFromLife = _bezier1(count, mode, 0.0, 2.5, not negative)
        d.Update.SizeFromVelocity = _bezier1(max(1, 5 - count), mode, 0.0, 120.0, negative)
        d.Update.RotationFromLifeAndIndex[0], d.Update.RotationFromLifeAndIndex[1] = 0.7, 0.001
    ops = shape["ops"]
    d.OpCount = len(ops)
    for o, kind in enumerate(ops):
        if kind == "gravity":
            d.Ops[o].Type = abi.OP_GRAVITY
            d.Ops[o].u.Gravity = scenes.gravity_params([((60., 70., 0.), 40., 500., 0), ((200., 60., 10.), 90., 700., 1),
                                                        ((100., 210., 0.), 120., 900., 2), ((220., 200., 5.), 3., 300., 0)][:shape.get("attractors", 4)],
                                                       maximum_acceleration=shape.get("max_accel", 64.0))
        elif kind == "noise":
            d.Ops[o].Type = abi.OP_NOISE
            d.Ops[o].u.Noise = scenes.noise_params(scenes.area_none(), (0.37 * 253, 0.81 * 127), (0.12 * 253, 0.55 * 127), 0.35,
                                                   replace_old_velocity=shape.get("replace", True))
        elif kind == "fma":
            d.Ops[o].Type = abi.OP_FMA
            d.Ops[o].u.FMA = scenes.fma_params(scenes.area_none(), position_add=(0.5, -0.25, 0.0), position_multiply=(1.001, 0.999, 1.0),
                                               velocity_add=(0.0, 1.5, 0.0), velocity_multiply=(0.98, 0.97, 1.0))
    d.UpdateMode = abi.UPDATE_POSITIONS
    d.Flags = abi.STEP_COUNT_LIVE if shape.get("count", True) else 0
    for s, (chunk, first, last) in enumerate(shape.get("spawns", ())):
        d.Spawns[s].ChunkIndex = chunk
        d.Spawns[s].Params = scenes.spawn_params(cs, first, last, 17 * s, (0.3 * 253, 0.6 * 127),
                                                 position=((128, 128, 0), (100, 90, 4), (0, 0, 0), scenes.FORMULA_SPHERICAL),
                                                 velocity=((0, 0, 0), (60, 60, 10), (0, 0, 0), scenes.FORMULA_SPHERICAL), life=(1.0, 2.0, 0.0))
        d.SpawnCount = s + 1
    return d


SHAPES = {
    "gravity+noise": dict(ops=("gravity", "noise")),
    "gravity+noise+spawn": dict(ops=("gravity", "noise"), spawns=((2, 100, 1500),)),
    "two spawn records, one past the used range": dict(ops=("noise", "gravity"), spawns=((2, 0, 63), (1, 4000, 4095))),
    "update only, no counting": dict(ops=(), count=False),
    "fma+gravity, one attractor": dict(ops=("fma", "gravity"), attractors=1, max_accel=2.0),
    "noise adds to the velocity": dict(ops=("noise", "fma"), replace=False),
    "rotation from velocity": dict(ops=("gravity",), rotation=True),
    "curves linear / mirror": dict(ops=("gravity",), curves=(2, 512 + 1, False)),
    "curves three-point / repeat / square": dict(ops=("gravity", "noise"), curves=(3, 256 + 2, True)),
    "curves cubic / clamp / sine": dict(ops=("noise",), curves=(4, 1, True)),
    "curves cubic / mirror": dict(ops=("gravity",), curves=(4, 600, False)),
}


@pytest.mark.parametrize("name", list(SHAPES))
def test_both_step_kernels_agree_and_match_the_oracle(ctx, oracle, name):
    shape = SHAPES[name]
    cs, n_chunks, steps = 64, 3, 3
    n = cs * cs
    rnd = scenes.randomness_table(11)
    eng = native.Engine(ctx, cs, rnd)
    pos, vel, attr = scenes.make_particles(77, n * n_chunks, life=(0.02, 2.5), dead_fraction=0.2)
    # chunk 2 is partly used (its tail has never been written): the kernels skip untouched units
    used = [n, n, 1024]
    systems = []
    prev = native.lib().ilm_debug_step_interpreter(0)
    try:
        for interpreter in (0, 1):
            native.lib().ilm_debug_step_interpreter(interpreter)
            s = native.System(eng)
            for c in range(n_chunks):
                s.add_chunk()
                sl = slice(c * n, c * n + used[c])
                s.upload(c, P, pos[sl]); s.upload(c, V, vel[sl]); s.upload(c, A, attr[sl])
            counts = []
            for _ in range(steps):
                d = _step(cs, shape)
                s.step(d)
                if d.Flags & abi.STEP_COUNT_LIVE:
                    counts.append(s.step_counts().copy())
            systems.append((s, counts))
    finally:
        native.lib().ilm_debug_step_interpreter(prev)
    (lean, lean_counts), (interp, interp_counts) = systems
    chunks = []
    for c in range(n_chunks):
        z = [np.zeros((n, 4), np.float32) for _ in range(5)]
        z[0][:used[c]] = pos[c * n:c * n + used[c]]; z[1][:used[c]] = vel[c * n:c * n + used[c]]; z[2][:used[c]] = attr[c * n:c * n + used[c]]
        chunks.append(z)
    want_counts = []
    for _ in range(steps):
        d = _step(cs, shape)
        got = oracle.step(chunks, cs, rnd, d, want_counts=bool(d.Flags & abi.STEP_COUNT_LIVE))
        if d.Flags & abi.STEP_COUNT_LIVE:
            want_counts.append(np.asarray(got).copy())
    for a, b, w in zip(lean_counts, interp_counts, want_counts):
        assert np.array_equal(a, b) and np.array_equal(a, w)
    for c in range(n_chunks):
        for k, plane in enumerate(PLANES):
            a, b = lean.download(c, plane), interp.download(c, plane)
            assert_bits_equal(a, b, "%s: chunk %d plane %d, specialised vs interpreting kernel" % (name, c, plane))
            assert_close(a, chunks[c][k], "%s: chunk %d plane %d vs oracle" % (name, c, plane), life_exact=(plane == P))
    assert np.array_equal(lean.live_counts(), interp.live_counts())
    for s, _ in systems:
        s.close()
    eng.close()


@pytest.mark.parametrize("fmt,packed1,units,spawn,cells", [(abi.SDF_UNORM16, False, 1, False, True), (abi.SDF_UNORM16, False, 2, True, True), (abi.SDF_UNORM16, False, 4, False, True),
                                                            (abi.SDF_UNORM16, False, 2, True, False), (abi.SDF_UNORM16, False, 4, False, False),
                                                            (abi.SDF_UNORM16, True, 2, False, True), (abi.SDF_UNORM16, True, 4, True, True), (abi.SDF_FP16, False, 2, False, True),
                                                            (abi.SDF_FP16, True, 4, True, True), (abi.SDF_FP16, True, 1, True, True)])
def test_the_lean_collision_step_equals_the_interpreter_bit_for_bit(ctx, oracle, fmt, packed1, units, spawn, cells):
    """Row a10 (UpdateParticleSystemWithDistanceField.fx:29-147) has a kernel of its own since r06 (step_lean_df_kernel: the common path
    at full width, the particles that collided parked in the wave's LDS ring and taken through the reference's whole update 64 at a time).
    It must give the interpreter's bits -- planes and live counts -- for 1, 2 and 4 units per wave, both field formats, the particle path's
    uniforms (DistanceFieldPacked1 = 0: the slice-0 sampler, through the one-load cells of a UNORM16 field and through its four taps) and the
    general ones, with a spawner feeding a partly used chunk and a life penalty that kills some of the bouncing particles; both are held
    against the oracle."""
    import os
    from tests.test_particles_gpu import cfg1_field
    cs, n_chunks, steps = 64, 3, 3
    n = cs * cs
    rnd = scenes.randomness_table(11)
    eng = native.Engine(ctx, cs, rnd)
    layout, atlas, dfu = cfg1_field(fmt, packed1)
    sdf = native.DistanceFieldTexture(ctx, atlas, fmt)
    pos, vel, attr = scenes.make_particles(78, n * n_chunks, pos_lo=(-20, -20, 0), pos_hi=(276, 276, 32), life=(0.02, 2.5), dead_fraction=0.2, categories=(0.0, 2.0))
    used = [n, n, 1024]

    def step_desc():
        d = _step(cs, dict(ops=("gravity", "noise"), spawns=((2, 900, 2400),) if spawn else ()))
        d.System = scenes.system_uniforms(cs, friction=0.05, max_velocity=900.0, life_decay=4.0, collision=(128.0, 0.6, 0.33, 0.4))
        d.UpdateMode = abi.UPDATE_WITH_DISTANCE_FIELD
        d.DistanceField = dfu
        return d
    systems = []
    prev = native.lib().ilm_debug_step_interpreter(0)
    os.environ["ILM_DF_UNITS"] = str(units)
    os.environ["ILM_DF_CELLS0"] = "1" if cells else "0"
    try:
        for interpreter in (0, 1):
            native.lib().ilm_debug_step_interpreter(interpreter)
            s = native.System(eng)
            s.set_distance_field(sdf)
            for c in range(n_chunks):
                s.add_chunk()
                sl = slice(c * n, c * n + used[c])
                s.upload(c, P, pos[sl]); s.upload(c, V, vel[sl]); s.upload(c, A, attr[sl])
            counts = []
            for _ in range(steps):
                s.step(step_desc())
                counts.append(s.step_counts().copy())
            systems.append((s, counts))
    finally:
        native.lib().ilm_debug_step_interpreter(prev)
        del os.environ["ILM_DF_UNITS"], os.environ["ILM_DF_CELLS0"]
    (lean, lean_counts), (interp, interp_counts) = systems
    chunks = []
    for c in range(n_chunks):
        z = [np.zeros((n, 4), np.float32) for _ in range(5)]
        z[0][:used[c]] = pos[c * n:c * n + used[c]]; z[1][:used[c]] = vel[c * n:c * n + used[c]]; z[2][:used[c]] = attr[c * n:c * n + used[c]]
        chunks.append(z)
    otex = oracle.make_texture(atlas, fmt)
    want_counts = [np.asarray(oracle.step(chunks, cs, rnd, step_desc(), sdf=otex, want_counts=True)).copy() for _ in range(steps)]
    for a, b, w in zip(lean_counts, interp_counts, want_counts):
        assert np.array_equal(a, b) and np.array_equal(a, w), (a, b, w)
    for c in range(n_chunks):
        for k, plane in enumerate(PLANES):
            a, b = lean.download(c, plane), interp.download(c, plane)
            assert_bits_equal(a, b, "collision step: chunk %d plane %d, lean vs interpreting kernel" % (c, plane))
            assert_close(a, chunks[c][k], "collision step: chunk %d plane %d vs oracle" % (c, plane), life_exact=(plane == P))
    bounced = sum(int((chunks[c][1][:, 3] == 3.0).sum()) for c in range(n_chunks))
    assert bounced > 50, bounced                     # the scene exercises the long path (BOUNCE_DELAY in the velocity's w)
    assert np.array_equal(lean.live_counts(), interp.live_counts())
    for s, _ in systems:
        s.close()
    sdf.close(); eng.close()


def test_the_slice0_cells_follow_the_field(ctx):
    """The one-load cells of the lean collision step are derived from the atlas: an atlas uploaded anew, regenerated slices and an atlas
    whose device pointer was handed out (it may change behind the library's back: rebuilt before every use) must all be seen by the next
    step.  Lean == interpreter after every change, and the change itself moves particles."""
    import ctypes as C
    from tests.test_particles_gpu import cfg1_field
    cs = 64
    n = cs * cs
    rnd = scenes.randomness_table(11)
    eng = native.Engine(ctx, cs, rnd)
    layout, atlas, dfu = cfg1_field(abi.SDF_UNORM16, False)
    other = np.ascontiguousarray(atlas[::-1, ::-1]).copy()                  # another field of the same shape
    pos, vel, attr = scenes.make_particles(79, n, pos_lo=(-20, -20, 0), pos_hi=(276, 276, 32), life=(0.5, 2.5), categories=(0.0, 2.0))
    sdfs = [native.DistanceFieldTexture(ctx, atlas, abi.SDF_UNORM16) for _ in range(2)]
    systems = []
    prev = native.lib().ilm_debug_step_interpreter(0)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def step(s):
        d = _step(cs, dict(ops=("gravity",)))
        d.System = scenes.system_uniforms(cs, friction=0.05, max_velocity=900.0, life_decay=0.1, collision=(128.0, 0.6, 0.33, 0.0))
        d.UpdateMode = abi.UPDATE_WITH_DISTANCE_FIELD
        d.DistanceField = dfu
        s.step(d)
    try:
        for interpreter in (0, 1):
            s = native.System(eng)
            s.set_distance_field(sdfs[interpreter])
            s.add_chunk()
            s.upload(0, P, pos); s.upload(0, V, vel); s.upload(0, A, attr)
            systems.append(s)
        snapshots = []
        for phase in range(4):
            for interpreter, s in enumerate(systems):
                native.lib().ilm_debug_step_interpreter(interpreter)
                if phase == 1:
                    sdfs[interpreter].upload(other)                           # ilm_sdf_upload: the version moves
                if phase == 2:                                                # the pointer escapes, the atlas is overwritten behind the library's back
                    ptr = sdfs[interpreter].device_ptr()
                    ctx.sync()
                    assert hip.hipMemcpy(ptr, atlas.ctypes.data, atlas.nbytes, 1) == 0
                step(s); step(s)
            a, b = [systems[0].download(0, k) for k in (P, V)], [systems[1].download(0, k) for k in (P, V)]
            for k in range(2):
                assert_bits_equal(a[k], b[k], "phase %d plane %d, lean (slice-0 cells) vs interpreting kernel" % (phase, k))
            snapshots.append(a[1].copy())
        bounced = [(v[:, 3] == 3.0) for v in snapshots]
        assert bounced[0].sum() > 50 and (bounced[1] != bounced[0]).any() and (bounced[2] != bounced[1]).any()
    finally:
        native.lib().ilm_debug_step_interpreter(prev)
    for s in systems:
        s.close()
    for f in sdfs:
        f.close()
    eng.close()


def test_the_lean_collision_step_on_large_chunks_and_many_collisions(ctx):
    """1024^2 chunks (the streaming variant's shape, 4 units per wave, the counts through the bucket lines) over a field in which a third of
    the particles sit inside obstacles: long passes of 64 run in the middle of a wave's walk, the ring wraps.  Lean == interpreter."""
    import os
    from tests.test_particles_gpu import cfg1_field
    cs = 1024
    n = cs * cs
    rnd = scenes.randomness_table(5)
    eng = native.Engine(ctx, cs, rnd)
    layout, atlas, dfu = cfg1_field(abi.SDF_UNORM16, False)
    sdf = native.DistanceFieldTexture(ctx, atlas)
    pos, vel, attr = scenes.make_particles(9, n, pos_lo=(20, 20, 0), pos_hi=(236, 236, 24), life=(0.02, 1.0), dead_fraction=0.3)
    out = []
    prev = native.lib().ilm_debug_step_interpreter(0)
    try:
        for interpreter, streaming in ((0, "0"), (0, "1"), (1, "0")):
            native.lib().ilm_debug_step_interpreter(interpreter)
            os.environ["ILM_STEP_STREAMING"] = streaming          # (the non-temporal variant is a launch decision by size: forced here)
            s = native.System(eng)
            s.set_distance_field(sdf)
            s.add_chunk()
            s.upload(0, P, pos); s.upload(0, V, vel); s.upload(0, A, attr)
            for _ in range(2):
                d = _step(cs, dict(ops=("gravity", "noise")))
                d.System = scenes.system_uniforms(cs, friction=0.05, max_velocity=900.0, life_decay=4.0, collision=(128.0, 0.6, 0.33, 0.4))
                d.UpdateMode = abi.UPDATE_WITH_DISTANCE_FIELD
                d.DistanceField = dfu
                s.step(d)
            counts = s.step_counts().copy()
            planes = [s.download(0, k) for k in (P, V, RC, RD)]
            assert counts[0] == int((planes[0][:, 3] > 0).sum()) and np.array_equal(counts, s.live_counts())
            out.append((counts, planes))
            s.close()
    finally:
        native.lib().ilm_debug_step_interpreter(prev)
        os.environ.pop("ILM_STEP_STREAMING", None)
    for other in (0, 1):
        assert np.array_equal(out[other][0], out[2][0])
        for k in range(4):
            assert_bits_equal(out[other][1][k], out[2][1][k], "1024^2 collision step plane %d, lean (variant %d) vs interpreting kernel" % (k, other))
    assert (out[2][1][1][:, 3] == 3.0).mean() > 0.02
    sdf.close(); eng.close()


@pytest.mark.parametrize("cs,n_chunks", [(1024, 2), (2048, 1)])
def test_both_step_kernels_agree_on_large_chunks(ctx, cs, n_chunks):
    """Chunks larger than the randomness table take the 5 x 5 noise tables; their ~4000+ blocks publish one count through the bucket lines."""
    n = cs * cs
    rnd = scenes.randomness_table(5)
    eng = native.Engine(ctx, cs, rnd)
    pos, vel, attr = scenes.make_particles(9, n, pos_hi=(1920, 1080, 32), life=(0.02, 1.0), dead_fraction=0.3)
    out = []
    prev = native.lib().ilm_debug_step_interpreter(0)
    try:
        for interpreter in (0, 1):
            native.lib().ilm_debug_step_interpreter(interpreter)
            s = native.System(eng)
            for c in range(n_chunks):
                s.add_chunk()
                s.upload(c, P, np.roll(pos, c * 977, axis=0)); s.upload(c, V, vel); s.upload(c, A, attr)
            for _ in range(2):
                s.step(_step(cs, dict(ops=("gravity", "noise"))))
            counts = s.step_counts().copy()
            planes = [[s.download(c, k) for k in (P, V, RC, RD)] for c in range(n_chunks)]
            assert np.array_equal(counts, s.live_counts())
            for c in range(n_chunks):
                assert counts[c] == int((planes[c][0][:, 3] > 0).sum())
            out.append((counts, planes))
            s.close()
    finally:
        native.lib().ilm_debug_step_interpreter(prev)
    assert np.array_equal(out[0][0], out[1][0])
    for c in range(n_chunks):
        for k in range(4):
            assert_bits_equal(out[0][1][c][k], out[1][1][c][k], "cs %d chunk %d plane %d, specialised vs interpreting kernel" % (cs, c, k))
    eng.close()


def test_counts_of_consecutive_counting_steps_do_not_mix(ctx):
    """Each counting step publishes under its own sequence number; polling never returns a stale or half-written table."""
    cs, n_chunks = 64, 5
    n = cs * cs
    rnd = scenes.randomness_table(3)
    eng = native.Engine(ctx, cs, rnd)
    s = native.System(eng)
    pos, vel, attr = scenes.make_particles(21, n * n_chunks, life=(0.01, 0.4), dead_fraction=0.1)
    for c in range(n_chunks):
        s.add_chunk()
        sl = slice(c * n, (c + 1) * n)
        s.upload(c, P, pos[sl]); s.upload(c, V, vel[sl]); s.upload(c, A, attr[sl])
    for i in range(40):
        d = _step(cs, dict(ops=("gravity",)))
        if i % 3 == 1:
            d.FirstChunk, d.ChunkCount = 1, 3          # chunks outside a step's range count zero in that step
        s.step(d)
        counts = s.step_counts()
        life = [int((s.download(c, P)[:, 3] > 0).sum()) for c in range(n_chunks)]
        lo, hi = (1, 4) if i % 3 == 1 else (0, n_chunks)
        for c in range(n_chunks):
            assert counts[c] == (life[c] if lo <= c < hi else 0), (i, c)
    # a burst of counting steps without reading in between: the last one's counts win
    for _ in range(10):
        s.step(_step(cs, dict(ops=("gravity",))))
    counts = s.step_counts()
    assert np.array_equal(counts, s.live_counts())
    s.close()
    eng.close()


def test_two_stream_steps_equal_one_stream_steps(ctx):
    """A step over >= 8192 units runs its two chunk halves on two streams that nothing joins between steps (api.hip, run_step);
    every other entry point joins them first.  Interleave steps with uploads, downloads, standalone counts, a chunk removal and a
    chunk added while the split is in force: the planes and counts equal the one-stream run's bit for bit."""
    cs, n_chunks = 256, 16
    n = cs * cs
    rnd = scenes.randomness_table(13)
    eng = native.Engine(ctx, cs, rnd)
    pos, vel, attr = scenes.make_particles(31, n * 4, pos_hi=(1920, 1080, 32), life=(0.05, 3.0), dead_fraction=0.1)
    patch = scenes.make_particles(32, 5000, pos_hi=(1920, 1080, 32), life=(0.5, 1.0))
    shape = dict(ops=("gravity", "noise"), spawns=((16, 0, 2000),))
    results = []
    prev = native.lib().ilm_debug_step_streams(2)
    try:
        for streams in (2, 1):
            native.lib().ilm_debug_step_streams(streams)
            s = native.System(eng)
            for c in range(n_chunks + 1):
                s.add_chunk()
            for c in range(n_chunks):
                k = (c % 4) * n
                s.upload(c, P, np.roll(pos[k:k + n], c * 131, axis=0)); s.upload(c, V, vel[k:k + n]); s.upload(c, A, attr[k:k + n])
            log = []
            for i in range(12):
                d = _step(cs, shape)
                if i % 4 != 1:
                    d.Flags = 0
                s.step(d)
                if i % 4 == 1:
                    log.append(s.step_counts().copy())
                if i == 2:      # an upload into a chunk of the second half, right behind a step
                    s.upload(11, P, patch[0], first_slot=777); s.upload(11, V, patch[1], first_slot=777)
                if i == 3:      # a download of both halves right behind a step
                    log.append(s.download(2, P).copy()); log.append(s.download(14, V).copy())
                if i == 5:
                    log.append(s.live_counts().copy())
                if i == 6:      # the table shrinks: chunks 5.. move down one index, the halves are cut anew
                    s.remove_chunk(4)
                    shape = dict(ops=("gravity", "noise"), spawns=((15, 0, 2000),))
                if i == 8:      # ... and grows again
                    c_new = s.add_chunk()
                    s.upload(c_new, P, pos[:n]); s.upload(c_new, V, vel[:n]); s.upload(c_new, A, attr[:n])
            for c in range(s.chunk_count()):
                for plane in PLANES:
                    log.append(s.download(c, plane))
            log.append(s.live_counts().copy())
            results.append(log)
            shape = dict(ops=("gravity", "noise"), spawns=((16, 0, 2000),))
            s.close()
    finally:
        native.lib().ilm_debug_step_streams(prev)
    two, one = results
    assert len(two) == len(one)
    for k, (a, b) in enumerate(zip(two, one)):
        if a.dtype == np.float32:
            assert_bits_equal(a, b, "log entry %d: two streams vs one" % k)
        else:
            assert np.array_equal(a, b), k
    eng.close()


def test_two_stream_steps_race_free_under_repetition(ctx, oracle):
    """The same split step sequence many times over, reading a chunk of each half after every step: a missing join would show up as a
    plane that is one step behind.  One chunk of each half is replayed by the oracle."""
    cs, n_chunks = 256, 8
    n = cs * cs
    rnd = scenes.randomness_table(17)
    eng = native.Engine(ctx, cs, rnd)
    pos, vel, attr = scenes.make_particles(41, n, pos_hi=(1920, 1080, 32), life=(0.2, 3.0), dead_fraction=0.1)
    s = native.System(eng)
    for c in range(n_chunks):
        s.add_chunk()
        s.upload(c, P, np.roll(pos, c * 17, axis=0)); s.upload(c, V, vel); s.upload(c, A, attr)
    shape = dict(ops=("gravity", "noise"))
    replay = {c: [np.roll(pos, c * 17, axis=0).copy(), vel.copy(), attr.copy(), np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)] for c in (1, 6)}
    for i in range(6):
        d = _step(cs, shape)
        s.step(d)
        got = {c: s.download(c, P) for c in (1, 6)}
        for c in (1, 6):
            d1 = _step(cs, shape)
            d1.FirstChunk, d1.ChunkCount = 0, 1
            want_counts = oracle.step([replay[c]], cs, rnd, d1, want_counts=True)
            assert_close(got[c], replay[c][0], "step %d chunk %d position vs oracle" % (i, c), life_exact=True)
            assert s.step_counts()[c] == want_counts[0]
    s.close()
    eng.close()


def test_a_context_whose_stream_was_handed_out_steps_on_that_stream(ctx):
    """ilm_ctx_stream gives the caller a stream to queue its own readers of the particle planes on, so such a context never uses its
    second stream (api.hip, Ctx::exported): a kernel-free reader -- an async copy the TEST queues on the exported stream right behind
    a large step, without any library call in between -- must see the stepped planes."""
    import ctypes as C
    cs, n_chunks = 256, 8
    n = cs * cs
    rnd = scenes.randomness_table(23)
    own = native.Context(0)
    stream = own.stream()
    assert stream
    eng = native.Engine(own, cs, rnd)
    s = native.System(eng)
    ref_eng = native.Engine(ctx, cs, rnd)
    ref = native.System(ref_eng)
    pos, vel, attr = scenes.make_particles(51, n, pos_hi=(1920, 1080, 32), life=(0.5, 3.0))
    for sysm in (s, ref):
        for c in range(n_chunks):
            sysm.add_chunk()
            sysm.upload(c, P, np.roll(pos, c * 29, axis=0)); sysm.upload(c, V, vel); sysm.upload(c, A, attr)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    ptr, stride = s.device_ptr(n_chunks - 1, 0)          # x plane of the last chunk: the half a split would put on the second stream
    out = np.zeros(n, np.float32)
    for _ in range(3):
        d = _step(cs, dict(ops=("gravity", "noise"), count=False))
        s.step(d); ref.step(d)
        assert hip.hipMemcpyAsync(out.ctypes.data, ptr, n * 4, 2, stream) == 0         # 2 = hipMemcpyDeviceToHost
        assert hip.hipStreamSynchronize(stream) == 0
        assert_bits_equal(out, ref.download(n_chunks - 1, P)[:, 0], "x plane read on the exported stream right behind the step")
    s.close(); eng.close(); ref.close(); ref_eng.close(); own.close()


def test_consumers_of_a_split_step_see_both_halves(ctx):
    """Read-back compaction, the rasteriser and the particle lights run on the context stream right behind a two-stream step: each must
    see the planes of BOTH halves (Ctx::main joins them).  Same calls after one-stream steps give the reference bits."""
    from tests import output_common as oc
    import ctypes as C
    from tests.lights_common import no_field_uniforms, particle_light_params
    cs, n_chunks = 256, 8
    n = cs * cs
    rnd = scenes.randomness_table(29)
    eng = native.Engine(ctx, cs, rnd)
    pos, vel, attr = scenes.make_particles(61, n, pos_hi=(640, 360, 16), life=(0.5, 3.0), dead_fraction=0.97)
    results = []
    prev = native.lib().ilm_debug_step_streams(2)
    try:
        for streams in (2, 1):
            native.lib().ilm_debug_step_streams(streams)
            s = native.System(eng)
            for c in range(n_chunks):
                s.add_chunk()
                s.upload(c, P, np.roll(pos, c * 37, axis=0)); s.upload(c, V, vel); s.upload(c, A, attr)
            log = []
            target = native.Lightmap(ctx, 640, 360, abi.LIGHTMAP_FLOAT4)
            lit = native.Lightmap(ctx, 640, 360, abi.LIGHTMAP_FLOAT4)
            for i in range(3):
                s.step(_step(cs, dict(ops=("gravity", "noise"), count=False)))
                recs, count = s.readback(oc.readback_params(size=(2.0, 2.0)))
                log.append(np.frombuffer(recs, dtype=np.uint8)[:C.sizeof(abi.ReadbackDrawCall) * count].copy())
                s.step(_step(cs, dict(ops=("gravity", "noise"), count=False)))
                target.clear((0.0, 0.0, 0.0, 0.0))
                native.render_particles(s, scenes.rasterize_params(size=(1.5, 1.5)), target)
                log.append(target.download().copy())
                s.step(_step(cs, dict(ops=("gravity", "noise"), count=False)))
                lit.clear((0.0, 0.0, 0.0, 1.0))
                native.render_particle_lights(ctx, s, particle_light_params(6.0, 20.0), scenes.environment(), no_field_uniforms(), None, None, lit)
                log.append(lit.download().copy())
            results.append(log)
            target.close(); lit.close(); s.close()
    finally:
        native.lib().ilm_debug_step_streams(prev)
    two, one = results
    assert len(two) == len(one) == 9
    for k, (a, b) in enumerate(zip(two, one)):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8)), "output %d differs between two-stream and one-stream stepping" % k
    assert two[0].size > 0 and two[1].any() and two[2].any()
    eng.close()
