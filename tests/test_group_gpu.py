"""Multi-device groups (ilm_group_*) on the GPU box.

The box has ONE GPU, so the exchange paths run with several members on the same device: that executes the real strip
rendering (the kernel, not the oracle), the in-place slot layout, the hipMemcpyPeerAsync fan-out with its event ordering and
RCCL at world size 1 -- everything but the xGMI wire.  The frames must equal the single-context frame bit for bit (same kernel,
same inputs, disjoint row ranges) and the oracle within the parity tolerance.
"""
import numpy as np
import pytest

from illuminant_amd import abi, native, scenes
from tests.test_lighting_gpu import small_scene
from tests.util import assert_close

pytestmark = pytest.mark.gpu

AMBIENT = (0.05, 0.06, 0.07, 1.0)


def single_context_frame(ctx, lights, env, dfu, atlas, sfmt, w, h, lm_fmt=abi.LIGHTMAP_FLOAT4):
    sdf = native.DistanceFieldTexture(ctx, atlas, sfmt)
    lm = native.Lightmap(ctx, w, h, lm_fmt)
    stats = native.render_sphere_lights(ctx, lights, env, dfu, None, sdf, AMBIENT, lm, want_stats=True)
    frame = lm.download()
    lm.close(); sdf.close()
    return frame, stats


def group_frame(devices, gather, lights, env, dfu, atlas, sfmt, w, h, lm_fmt=abi.LIGHTMAP_FLOAT4, want_stats=True):
    g = native.Group(devices)
    sdfs = [native.DistanceFieldTexture(c, atlas, sfmt) for c in g.contexts]      # replicated input: one per member
    glm = native.GroupLightmap(g, w, h, lm_fmt)
    stats = g.render_sphere_lights(lights, env, dfu, None, sdfs, AMBIENT, glm, gather, want_stats=want_stats)
    g.sync()
    frames = [glm.download(i) for i in range(g.n_local)]
    info = dict(strips=glm.strips, slot_rows=glm.slot_rows, comm_ranks=g.comm_ranks(), world=g.world)
    glm.close()
    for s in sdfs:
        s.close()
    g.close()
    return frames, stats, info


@pytest.mark.parametrize("gather", [native.GATHER_NONE, native.GATHER_PEER, native.GATHER_RCCL])
def test_group_of_one_equals_the_single_context_frame(ctx, oracle, gather):
    layout, atlas, dfu, lights, w, h = small_scene()
    env = scenes.environment()
    want, wstats = single_context_frame(ctx, lights, env, dfu, atlas, abi.SDF_UNORM16, w, h)
    frames, stats, info = group_frame([0], gather, lights, env, dfu, atlas, abi.SDF_UNORM16, w, h)
    assert info["world"] == 1 and info["strips"] == [(0, h)] and info["slot_rows"] % 16 == 0 and info["slot_rows"] >= h
    assert np.array_equal(frames[0], want)
    assert (stats.SdfSamples, stats.PixelLightPairs, stats.TracedPairs) == (wstats.SdfSamples, wstats.PixelLightPairs, wstats.TracedPairs)
    owant, ostats = oracle.render_sphere_lights(lights, env, dfu, None, oracle.make_texture(atlas, abi.SDF_UNORM16), AMBIENT, w, h, want_stats=True)
    assert stats.SdfSamples == ostats.SdfSamples
    assert_close(frames[0], owant, "group frame vs oracle")


@pytest.mark.parametrize("members,fmt", [(2, abi.LIGHTMAP_FLOAT4), (3, abi.LIGHTMAP_HALF4), (8, abi.LIGHTMAP_FLOAT4)])
def test_peer_fan_out_composites_the_frame_on_every_member(ctx, members, fmt):
    """Several members on device 0: each renders its strip with the kernel, pushes it to the others; every member ends with
    the whole frame, equal to the single-context frame (ragged last strip: 112 rows in slots of 64 / 48 / 16)."""
    layout, atlas, dfu, lights, w, h = small_scene(abi.SDF_FP16)
    env = scenes.environment()
    want, wstats = single_context_frame(ctx, lights, env, dfu, atlas, abi.SDF_FP16, w, h, fmt)
    frames, stats, info = group_frame([0] * members, native.GATHER_PEER, lights, env, dfu, atlas, abi.SDF_FP16, w, h, fmt)
    strips = info["strips"]
    assert len(strips) == members and strips[0][0] == 0 and max(e for _, e in strips) == h
    assert all(b % 16 == 0 for b, _ in strips) and sum(e - b for b, e in strips) == h
    assert members * info["slot_rows"] >= h
    for i, f in enumerate(frames):
        assert np.array_equal(f.view(np.uint8), want.view(np.uint8)), "member %d's composited frame differs" % i
    assert (stats.SdfSamples, stats.PixelLightPairs, stats.TracedPairs) == (wstats.SdfSamples, wstats.PixelLightPairs, wstats.TracedPairs)


def test_without_a_gather_every_member_holds_only_its_strip(ctx):
    layout, atlas, dfu, lights, w, h = small_scene()
    env = scenes.environment()
    want, _ = single_context_frame(ctx, lights, env, dfu, atlas, abi.SDF_UNORM16, w, h)
    frames, _, info = group_frame([0, 0], native.GATHER_NONE, lights, env, dfu, atlas, abi.SDF_UNORM16, w, h, want_stats=False)
    for i, (b, e) in enumerate(info["strips"]):
        assert np.array_equal(frames[i][b:e], want[b:e])
        other = np.ones(h, bool); other[b:e] = False
        assert not frames[i][other].any(), "rows outside member %d's strip were written" % i


def test_rccl_refuses_members_that_share_a_device(ctx):
    layout, atlas, dfu, lights, w, h = small_scene()
    g = native.Group([0, 0])
    glm = native.GroupLightmap(g, w, h)
    with pytest.raises(native.IlluminantError) as e:
        glm.gather(native.GATHER_RCCL)
    assert e.value.code == abi.ERR_INVALID_ARGUMENT and "one device per member" in str(e.value)
    glm.close(); g.close()


def test_rank_group_of_world_size_one_runs_the_rccl_path(ctx):
    """The one-process-per-GPU shape at world size 1: ncclCommInitRank, the in-place ncclAllGather on the context stream."""
    layout, atlas, dfu, lights, w, h = small_scene()
    env = scenes.environment()
    want, _ = single_context_frame(ctx, lights, env, dfu, atlas, abi.SDF_UNORM16, w, h)
    uid = native.Group.unique_id()
    assert len(uid) == 128
    g = native.Group.rank(0, 0, 1, uid)
    assert (g.n_local, g.world, g.first_rank) == (1, 1, 0) and g.comm_ranks() == 1
    sdf = native.DistanceFieldTexture(g.contexts[0], atlas, abi.SDF_UNORM16)
    glm = native.GroupLightmap(g, w, h)
    g.render_sphere_lights(lights, env, dfu, None, [sdf], AMBIENT, glm, native.GATHER_RCCL)
    g.sync()
    assert np.array_equal(glm.download(0), want)
    glm.close(); sdf.close(); g.close()


def test_rank_group_gather_modes_at_world_one_are_no_ops(ctx):
    uid = native.Group.unique_id()
    g = native.Group.rank(0, 0, 1, uid)
    glm = native.GroupLightmap(g, 64, 48)
    glm.gather(native.GATHER_NONE)
    glm.gather(native.GATHER_RCCL)
    glm.close(); g.close()


def test_generic_all_gather_of_position_planes(ctx):
    """SURVEY 8e's optional Pos+Life all-gather for a global consumer: 3 members, each fills its slot of a plane-shaped buffer."""
    import ctypes as C
    g = native.Group([0, 0, 0])
    n = 4096
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    bufs = []
    rng = np.random.default_rng(3)
    slots = [rng.random(n, dtype=np.float32) for _ in range(3)]
    for i in range(3):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), 3 * n * 4) == 0
        host = np.zeros(3 * n, np.float32)
        host[i * n:(i + 1) * n] = slots[i]
        assert hip.hipMemcpy(p, host.ctypes.data_as(C.c_void_p), host.nbytes, 1) == 0
        bufs.append(p)
    g.all_gather([b.value for b in bufs], n * 4, native.GATHER_PEER)
    g.sync()
    want = np.concatenate(slots)
    for i in range(3):
        host = np.empty(3 * n, np.float32)
        assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), bufs[i], host.nbytes, 2) == 0
        assert np.array_equal(host, want)
        hip.hipFree(bufs[i])
    g.close()


def test_group_live_counts_follow_chunk_mod_world(ctx, oracle):
    """Chunks sharded chunk -> rank by c % world over two members; the gathered liveness table equals the single-system one
    and the oracle's (integers, bit-exact)."""
    cs, n_chunks = 32, 5
    n = cs * cs
    rnd = scenes.randomness_table(7)
    pos, vel, attr = scenes.make_particles(42, n * n_chunks, dead_fraction=0.3)
    d = abi.StepDesc()
    d.FirstChunk, d.ChunkCount = 0, -1
    d.System = scenes.system_uniforms(cs, friction=0.1, max_velocity=2048.0, life_decay=20.0)
    d.Update = abi.UpdateParams.default()
    d.OpCount = 1
    d.Ops[0].Type = abi.OP_GRAVITY
    d.Ops[0].u.Gravity = scenes.gravity_params([((128.0, 128.0, 0.0), 150.0, 60.0, 1)])
    d.UpdateMode = abi.UPDATE_POSITIONS
    d.Flags = abi.STEP_COUNT_LIVE

    def fill(system, chunks):
        for c in chunks:
            k = system.add_chunk()
            sl = slice(c * n, (c + 1) * n)
            system.upload(k, abi.PLANE_POSITION, pos[sl]); system.upload(k, abi.PLANE_VELOCITY, vel[sl]); system.upload(k, abi.PLANE_ATTRIBUTES, attr[sl])

    eng = native.Engine(ctx, cs, rnd)
    whole = native.System(eng)
    fill(whole, range(n_chunks))
    whole.step(d)
    want = whole.step_counts()
    planes = [[pos[c * n:(c + 1) * n].copy(), vel[c * n:(c + 1) * n].copy(), attr[c * n:(c + 1) * n].copy(),
               np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)] for c in range(n_chunks)]
    assert np.array_equal(want, oracle.step(planes, cs, rnd, d, want_counts=True))
    assert 0 < int(want.sum()) < n * n_chunks

    g = native.Group([0, 0])
    engines = [native.Engine(c, cs, rnd) for c in g.contexts]
    systems = [native.System(e) for e in engines]
    for r in range(2):
        fill(systems[r], range(r, n_chunks, 2))
        systems[r].step(d)
    got = g.live_counts(systems, n_chunks)
    assert got.dtype == np.uint32 and np.array_equal(got, want)
    # the sharded chunks hold the same particles as the whole system's
    for c in range(n_chunks):
        assert np.array_equal(systems[c % 2].download(c // 2, abi.PLANE_POSITION), whole.download(c, abi.PLANE_POSITION))
    # a table that does not match the sharding rule is refused
    with pytest.raises(native.IlluminantError):
        g.live_counts(systems, n_chunks + 2)
    for s in systems + [whole]:
        s.close()
    for e in engines + [eng]:
        e.close()
    # the group cannot go while member objects live -- here everything is closed
    g.close()


def test_group_destroy_refuses_while_objects_live(ctx):
    g = native.Group([0])
    lm = native.Lightmap(g.contexts[0], 32, 32)
    with pytest.raises(native.IlluminantError) as e:
        g.close()
    assert e.value.code == abi.ERR_STATE
    lm.close()
    g.close()


def test_host_all_gather_is_the_barrier_and_the_small_exchange(ctx):
    import struct
    g = native.Group([0, 0, 0])
    local = b"".join(struct.pack("<d", 1.5 * (i + 1)) for i in range(3))
    slots = g.host_all_gather(local)
    assert [struct.unpack("<d", b)[0] for b in slots] == [1.5, 3.0, 4.5]
    g.close()
    r = native.Group.rank(0, 0, 1, native.Group.unique_id())
    assert r.host_all_gather(struct.pack("<d", 7.25)) == [struct.pack("<d", 7.25)]
    r.close()


@pytest.mark.parametrize("members,fmt", [(2, abi.LIGHTMAP_FLOAT4), (3, abi.LIGHTMAP_HALF4), (5, abi.LIGHTMAP_FLOAT4)])
def test_cost_balanced_strips_are_exchanged_range_by_range(ctx, members, fmt):
    """ilm_group_lightmap_set_strips with sharding.balanced_row_strips (unequal strips of whole tile bands): every member renders its strip
    at its true rows, the ranges are pushed to the other members, and every member ends with the single-context frame bit for bit."""
    from illuminant_amd import sharding
    layout, atlas, dfu, lights, w, h = small_scene(abi.SDF_FP16)
    env = scenes.environment()
    want, wstats = single_context_frame(ctx, lights, env, dfu, atlas, abi.SDF_FP16, w, h, fmt)
    g = native.Group([0] * members)
    sdfs = [native.DistanceFieldTexture(c, atlas, abi.SDF_FP16) for c in g.contexts]
    glm = native.GroupLightmap(g, w, h, fmt)
    equal = list(glm.strips)
    strips = sharding.balanced_row_strips(h, members, lights)
    glm.set_strips(strips)
    assert glm.strips == strips and strips[0][0] == 0 and strips[-1][1] == h and all(b % 16 == 0 for b, _ in strips)
    stats = g.render_sphere_lights(lights, env, dfu, None, sdfs, AMBIENT, glm, native.GATHER_PEER, want_stats=True)
    g.sync()
    for i in range(members):
        assert np.array_equal(glm.download(i).view(np.uint8), want.view(np.uint8)), "member %d's composited frame differs" % i
    assert (stats.SdfSamples, stats.PixelLightPairs, stats.TracedPairs) == (wstats.SdfSamples, wstats.PixelLightPairs, wstats.TracedPairs)
    # a table that does not tile the frame is refused; NULL restores the equal slots
    bad = list(strips)
    bad[-1] = (bad[-1][0], h - 1)
    with pytest.raises(native.IlluminantError):
        glm.set_strips(bad)
    with pytest.raises(native.IlluminantError):
        glm.set_strips([(b + 8, e) if i == 1 else (b, e) for i, (b, e) in enumerate(strips)])
    glm.set_strips(None)
    assert glm.strips == equal
    glm.close()
    for s in sdfs:
        s.close()
    g.close()


def test_unequal_strips_over_rccl_at_world_one(ctx):
    """The send / receive exchange with a communicator of one rank: nothing to send, the frame is the single-context frame."""
    layout, atlas, dfu, lights, w, h = small_scene()
    env = scenes.environment()
    want, _ = single_context_frame(ctx, lights, env, dfu, atlas, abi.SDF_UNORM16, w, h)
    g = native.Group([0])
    sdf = native.DistanceFieldTexture(g.contexts[0], atlas, abi.SDF_UNORM16)
    glm = native.GroupLightmap(g, w, h)
    glm.set_strips([(0, h)])
    g.render_sphere_lights(lights, env, dfu, None, [sdf], AMBIENT, glm, native.GATHER_RCCL)
    g.sync()
    assert np.array_equal(glm.download(0), want)
    glm.close(); sdf.close(); g.close()


def test_a_group_members_padded_lightmap_resolves_into_a_frame_sized_back_buffer(ctx, oracle):
    """A member's lightmap is world x slot_rows rows tall (>= the frame); the host's back buffer and albedo are frame-sized.  The resolve
    takes them as they are: rows [0, height) of the member equal the resolve of a frame-sized copy."""
    layout, atlas, dfu, lights, w, h = small_scene()
    env = scenes.environment()
    g = native.Group([0, 0, 0])
    sdfs = [native.DistanceFieldTexture(c, atlas, abi.SDF_UNORM16) for c in g.contexts]
    glm = native.GroupLightmap(g, w, h, abi.LIGHTMAP_HALF4)
    assert glm.members[0].height > h                       # 112 rows in three slots of 48
    g.render_sphere_lights(lights, env, dfu, None, sdfs, AMBIENT, glm, native.GATHER_PEER)
    g.sync()
    hc = abi.HDRConfiguration()
    hc.Mode, hc.InverseScaleFactor, hc.Exposure, hc.Gamma, hc.WhitePoint = abi.HDR_TONE_MAP, 1.0, 1.2, 1.0 / 2.2, 3.0
    c0 = g.contexts[0]
    back = native.Lightmap(c0, w, h, abi.LIGHTMAP_RGBA8)
    albedo = native.Lightmap(c0, w, h, abi.LIGHTMAP_RGBA8)
    albedo.upload(np.full((h, w, 4), 200, np.uint8))
    native.resolve_lighting(glm.members[0], back, hc, albedo=albedo)
    got = back.download()
    frame = native.Lightmap(c0, w, h, abi.LIGHTMAP_HALF4)
    frame.upload(glm.download(0))
    want_lm = native.Lightmap(c0, w, h, abi.LIGHTMAP_RGBA8)
    native.resolve_lighting(frame, want_lm, hc, albedo=albedo)
    assert np.array_equal(got, want_lm.download())
    with pytest.raises(native.IlluminantError):
        native.resolve_lighting(glm.members[0], back, hc, 0, h + 1)            # rows past the back buffer
    for x in (back, albedo, frame, want_lm, glm):
        x.close()
    for s in sdfs:
        s.close()
    g.close()


@pytest.mark.parametrize("members,fmt,balanced", [(2, abi.LIGHTMAP_FLOAT4, False), (3, abi.LIGHTMAP_HALF4, True), (5, abi.LIGHTMAP_HALF4, False),
                                                   (8, abi.LIGHTMAP_RGBA8, True)])
def test_store_mode_composites_the_frame_without_a_gather(ctx, members, fmt, balanced):
    """ILM_GATHER_STORE (r05): the light kernel's final store writes each member's strip into EVERY member's copy of the frame (the other
    members' buffers through the mirror table; on this one-GPU box the members share a device, on a node the buffers are peer-mapped
    over xGMI), so no copy phase follows the strips.  Every member must end with the single-context frame bit for bit -- equal slots and
    cost-balanced strips, all three lightmap formats, the composite call and the host-driven form (armed lightmap, per-member render
    calls, gather(STORE) as the fence), a second light group accumulated on top, and particle lights through the member handles."""
    from illuminant_amd import sharding
    layout, atlas, dfu, lights, w, h = small_scene(abi.SDF_FP16, n_lights=24, width=176, height=160)
    env = scenes.environment()
    want, wstats = single_context_frame(ctx, lights, env, dfu, atlas, abi.SDF_FP16, w, h, fmt)
    g = native.Group([0] * members)
    sdfs = [native.DistanceFieldTexture(c, atlas, abi.SDF_FP16) for c in g.contexts]
    glm = native.GroupLightmap(g, w, h, fmt)
    if balanced:
        glm.set_strips(sharding.balanced_row_strips(h, members, lights))
    try:
        # 1. the composite call arms the mode for its own duration
        stats = g.render_sphere_lights(lights, env, dfu, None, sdfs, AMBIENT, glm, native.GATHER_STORE, want_stats=True)
        g.sync()
        assert (stats.SdfSamples, stats.PixelLightPairs, stats.TracedPairs) == (wstats.SdfSamples, wstats.PixelLightPairs, wstats.TracedPairs)
        for i in range(members):
            assert np.array_equal(glm.download(i).view(np.uint8), want.view(np.uint8)), "composite call: member %d's frame differs" % i
        # ... and leaves it disarmed: the next plain strip stays on its member
        for m in glm.members:
            m.clear()
        g.sync()
        b, e = glm.strips[0]
        native.render_sphere_lights(g.contexts[0], lights, env, dfu, None, sdfs[0], AMBIENT, glm.members[0], b, e)
        g.sync()
        if members > 1:
            assert not glm.download(1)[b:e].any(), "a disarmed lightmap stored into another member"
        with pytest.raises(native.IlluminantError):
            glm.gather(native.GATHER_STORE)                      # not armed
        # 2. host-driven: armed lightmap, every member renders its own strip, the fence is the gather; then a second light group on top
        glm.store_mode(True)
        few = (abi.LightVertex * 3)(*[lights[i] for i in (2, 9, 17)])
        # (the second group of member 0 goes through an ALIAS of its buffer -- a lightmap object of the host's own around
        # ilm_lightmap_device_ptr, what a renderer that wraps the group's buffer holds: the store-mode table follows the buffer)
        alias = native.Lightmap(g.contexts[0], w, h, fmt, external_ptr=glm.members[0].device_ptr())
        for group_lights, ambient in ((lights, AMBIENT), (few, None)):
            glm.gather(native.GATHER_STORE)                      # (readers of the previous pass are done before anybody overwrites)
            for i in range(members):
                b, e = glm.strips[i]
                target = alias if (i == 0 and ambient is None) else glm.members[i]
                native.render_sphere_lights(g.contexts[i], group_lights, env, dfu, None, sdfs[i], ambient, target, b, e)
            glm.gather(native.GATHER_STORE)
        g.sync()
        alias.close()
        # ... but an alias held by ANOTHER context (a sibling's renderer around the member's buffer) is refused while the mode is armed:
        # its passes would run on a stream the fence does not cover and would reach nobody else's frame
        sib = g.contexts[0].sibling()
        foreign = native.Lightmap(sib, w, h, fmt, external_ptr=glm.members[0].device_ptr())
        with pytest.raises(native.IlluminantError) as refusal:
            native.render_sphere_lights(sib, few, env, dfu, None, sdfs[0], None, foreign, *glm.strips[0])
        assert refusal.value.code == abi.ERR_STATE and "ANOTHER context" in str(refusal.value)
        foreign.close(); sib.close()
        lm = native.Lightmap(ctx, w, h, fmt)
        sdf0 = native.DistanceFieldTexture(ctx, atlas, abi.SDF_FP16)
        native.render_sphere_lights(ctx, lights, env, dfu, None, sdf0, AMBIENT, lm)
        native.render_sphere_lights(ctx, few, env, dfu, None, sdf0, None, lm)
        want2 = lm.download()
        lm.close(); sdf0.close()
        assert not np.array_equal(want2, want)
        for i in range(members):
            assert np.array_equal(glm.download(i).view(np.uint8), want2.view(np.uint8)), "host-driven store mode: member %d's frame differs" % i
        with pytest.raises(native.IlluminantError):
            glm.gather(native.GATHER_PEER)                       # an armed lightmap has nothing to copy
        with pytest.raises(native.IlluminantError):
            g.render_sphere_lights(lights, env, dfu, None, sdfs, AMBIENT, glm, native.GATHER_PEER)
        # 3. particle lights on top, through the member handles (replicated particle state; the accumulate pass is mirrored like any other)
        from tests import lights_common as lc
        cs = 16
        pos, vel, attr = scenes.make_particles(44, cs * cs, pos_lo=(0, 0, 2), pos_hi=(w, h, 30), dead_fraction=0.3)
        rc = scenes.uniform(45, (cs * cs, 4), 0.3, 1.0).astype(np.float32); rc[:, :3] *= rc[:, 3:4]
        params = lc.particle_light_params(3.0, 26.0, (1.0, 0.9, 0.8, 1.0), casts_shadows=True)
        systems = []
        for c in list(g.contexts) + [ctx]:
            eng = native.Engine(c, cs, scenes.randomness_table(7))
            sysm = native.System(eng); sysm.add_chunk()
            sysm.upload(0, abi.PLANE_POSITION, pos); sysm.upload(0, abi.PLANE_RENDER_COLOR, rc)
            systems.append((eng, sysm))
        glm.gather(native.GATHER_STORE)
        for i in range(members):
            b, e = glm.strips[i]
            native.render_particle_lights(g.contexts[i], systems[i][1], params, env, dfu, None, sdfs[i], glm.members[i], row_begin=b, row_end=e)
        glm.gather(native.GATHER_STORE)
        g.sync()
        lm = native.Lightmap(ctx, w, h, fmt)
        sdf0 = native.DistanceFieldTexture(ctx, atlas, abi.SDF_FP16)
        native.render_sphere_lights(ctx, lights, env, dfu, None, sdf0, AMBIENT, lm)
        native.render_sphere_lights(ctx, few, env, dfu, None, sdf0, None, lm)
        native.render_particle_lights(ctx, systems[-1][1], params, env, dfu, None, sdf0, lm)
        want3 = lm.download()
        lm.close(); sdf0.close()
        assert not np.array_equal(want3, want2)
        for i in range(members):
            assert np.array_equal(glm.download(i).view(np.uint8), want3.view(np.uint8)), "particle lights in store mode: member %d's frame differs" % i
        for eng, sysm in systems:
            sysm.close(); eng.close()
        glm.store_mode(False)
    finally:
        glm.close()
        for s in sdfs:
            s.close()
        g.close()


def test_store_mode_of_a_rank_group_at_world_one(ctx):
    """A group that spans processes arms the store mode through IPC handles of the ranks' buffers (a collective; world sizes 2, 3 and 8 on this
    GPU: tests/test_two_ranks_one_gpu.py); with one rank there is nobody to map and the frame is simply the rank's own."""
    layout, atlas, dfu, lights, w, h = small_scene()
    env = scenes.environment()
    want, _ = single_context_frame(ctx, lights, env, dfu, atlas, abi.SDF_UNORM16, w, h)
    g = native.Group.rank(0, 0, 1, native.Group.unique_id())
    sdf = native.DistanceFieldTexture(g.contexts[0], atlas, abi.SDF_UNORM16)
    glm = native.GroupLightmap(g, w, h)
    glm.store_mode(True)
    g.render_sphere_lights(lights, env, dfu, None, [sdf], AMBIENT, glm, native.GATHER_STORE)
    g.sync()
    assert np.array_equal(glm.download(0), want)
    glm.store_mode(False)
    glm.close(); sdf.close(); g.close()


@pytest.mark.parametrize("members,balanced", [(2, False), (3, True), (5, True)])
def test_asynchronous_exchange_over_a_ring_of_two_lightmaps(ctx, members, balanced):
    """ILM_GATHER_ASYNC (r05): the exchange of frame N runs on the members' second streams while their context streams render the strips of
    frame N + 1 into the OTHER lightmap of a ring of two (the reference's BufferRing, LightingRenderer.cs:472-485).  Six frames with six
    light sets, nothing synchronised except ilm_group_lightmap_wait in front of a lightmap's reuse: every member's copy of every frame
    equals the single-context frame bit for bit (peer copies on one device; the RCCL form of the same calls at world 1)."""
    from illuminant_amd import sharding
    layout, atlas, dfu, _, w, h = small_scene(abi.SDF_UNORM16, width=240, height=176)
    env = scenes.environment()
    frames = 6
    light_sets = [scenes.random_lights(300 + k, 10 + 3 * k, w, h, z=(8.0, 48.0), radius=10.0, ramp=(40.0, 140.0)) for k in range(frames)]
    want = []
    for k in range(frames):
        f, _ = single_context_frame(ctx, light_sets[k], env, dfu, atlas, abi.SDF_UNORM16, w, h, abi.LIGHTMAP_HALF4)
        want.append(f)
    assert not np.array_equal(want[0], want[2])
    g = native.Group([0] * members)
    sdfs = [native.DistanceFieldTexture(c, atlas, abi.SDF_UNORM16) for c in g.contexts]
    ring = [native.GroupLightmap(g, w, h, abi.LIGHTMAP_HALF4) for _ in range(2)]
    if balanced:
        for glm in ring:
            glm.set_strips(sharding.balanced_row_strips(h, members, light_sets[0]))
    try:
        def check(k):
            glm = ring[k & 1]
            glm.wait()
            for i in range(members):
                assert np.array_equal(glm.download(i).view(np.uint16), want[k].view(np.uint16)), "frame %d, member %d" % (k, i)
        for k in range(frames):
            if k >= 2:
                check(k - 2)
            g.render_sphere_lights(light_sets[k], env, dfu, None, sdfs, AMBIENT, ring[k & 1], native.GATHER_PEER | native.GATHER_ASYNC)
        check(frames - 2); check(frames - 1)
        # the flag goes with a copying mode only
        with pytest.raises(native.IlluminantError):
            ring[0].gather(native.GATHER_STORE | native.GATHER_ASYNC)
        with pytest.raises(native.IlluminantError):
            ring[0].gather(native.GATHER_NONE | native.GATHER_ASYNC)
    finally:
        g.sync()
        for glm in ring:
            glm.close()
        for s in sdfs:
            s.close()
        g.close()


def test_asynchronous_rccl_exchange_at_world_one(ctx):
    layout, atlas, dfu, lights, w, h = small_scene()
    env = scenes.environment()
    want, _ = single_context_frame(ctx, lights, env, dfu, atlas, abi.SDF_UNORM16, w, h)
    g = native.Group.rank(0, 0, 1, native.Group.unique_id())
    sdf = native.DistanceFieldTexture(g.contexts[0], atlas, abi.SDF_UNORM16)
    ring = [native.GroupLightmap(g, w, h) for _ in range(2)]
    for k in range(4):
        g.render_sphere_lights(lights, env, dfu, None, [sdf], AMBIENT, ring[k & 1], native.GATHER_RCCL | native.GATHER_ASYNC)
    for glm in ring:
        glm.wait()
        assert np.array_equal(glm.download(0), want)
    g.host_all_gather(b"\0" * 8)          # the barrier drains the exchange streams too
    for glm in ring:
        glm.close()
    sdf.close(); g.close()


@pytest.mark.parametrize("members,gather", [(2, native.GATHER_PEER), (3, native.GATHER_PEER), (1, native.GATHER_RCCL)])
def test_gather_chunks_lights_every_strip_with_every_ranks_particles(ctx, members, gather):
    """cfg5 joins P and L and chunk c lives on rank c % world: a member's strip sees only its own chunks' particles as
    lights until the table is made whole.  ilm_group_gather_chunks moves Pos+Life and RenderColor of every chunk into every member's
    gathered system (chunk order = table order), ilm_render_particle_lights over that system lights the member's strip, the strips are
    exchanged: every member's frame equals the single-context particle-light frame BIT FOR BIT (same records in the same order)."""
    from tests import lights_common as lc
    from tests.test_lights_ext_gpu import particle_scene, small_field
    w, h, cs, n_chunks = 160, 112, 16, 5
    atlas, dfu = small_field()
    env = scenes.environment()
    chunks = particle_scene(cs, n_chunks, w, h)
    params = lc.particle_light_params(3.0, 30.0, (0.9, 0.8, 0.7, 0.6), casts_shadows=True)
    rnd = scenes.randomness_table(7)

    def fill(system, which):
        for c in which:
            k = system.add_chunk()
            system.upload(k, abi.PLANE_POSITION, chunks[c][0]); system.upload(k, abi.PLANE_RENDER_COLOR, chunks[c][3])

    # the one-context frame
    eng = native.Engine(ctx, cs, rnd)
    whole = native.System(eng)
    fill(whole, range(n_chunks))
    sdf = native.DistanceFieldTexture(ctx, atlas)
    lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
    native.render_sphere_lights(ctx, None, env, dfu, None, sdf, AMBIENT, lm)
    st = native.render_particle_lights(ctx, whole, params, env, dfu, None, sdf, lm, want_stats=True)
    want = lm.download()
    assert st.PixelLightPairs > 1000
    for x in (lm, sdf, whole, eng):
        x.close()

    g = native.Group([0] * members) if gather == native.GATHER_PEER else native.Group.rank(0, 0, 1, native.Group.unique_id())
    engines = [native.Engine(c, cs, rnd) for c in g.contexts]
    sources = [native.System(e) for e in engines]
    gathered = [native.System(e) for e in engines]
    for r in range(members):
        fill(sources[r], range(r, n_chunks, members))
        for _ in range(n_chunks):
            gathered[r].add_chunk()
    sdfs = [native.DistanceFieldTexture(c, atlas) for c in g.contexts]
    glm = native.GroupLightmap(g, w, h, abi.LIGHTMAP_FLOAT4)
    try:
        g.gather_chunks(sources, gathered, n_chunks, 0, 4, gather)          # Pos+Life
        g.gather_chunks(sources, gathered, n_chunks, 12, 4, gather)         # RenderColor
        total = 0
        for i, c in enumerate(g.contexts):
            b, e = glm.strips[i]
            native.render_sphere_lights(c, None, env, dfu, None, sdfs[i], AMBIENT, glm.members[i], b, e)
            s_ = native.render_particle_lights(c, gathered[i], params, env, dfu, None, sdfs[i], glm.members[i], row_begin=b, row_end=e, want_stats=True)
            total += s_.PixelLightPairs
        glm.gather(gather)
        g.sync()
        assert total == st.PixelLightPairs
        for i in range(members):
            assert np.array_equal(glm.download(i), want), "member %d" % i
            for c in range(n_chunks):     # the gathered planes are the owners' planes
                assert np.array_equal(gathered[i].download(c, abi.PLANE_POSITION), chunks[c][0])
                assert np.array_equal(gathered[i].download(c, abi.PLANE_RENDER_COLOR), chunks[c][3])
                assert not gathered[i].download(c, abi.PLANE_VELOCITY).any()           # components outside the range stay as they were
        # a table that does not match the sharding rule, a gathered system of the wrong size and a bad component range are refused
        with pytest.raises(native.IlluminantError):
            g.gather_chunks(sources, gathered, n_chunks + 1, 0, 4, gather)
        if members > 1:
            with pytest.raises(native.IlluminantError):
                g.gather_chunks(sources, sources, n_chunks, 0, 4, gather)
        with pytest.raises(native.IlluminantError):
            g.gather_chunks(sources, gathered, n_chunks, 18, 4, gather)
    finally:
        g.sync()
        glm.close()
        for x in sdfs + sources + gathered + engines:
            x.close()
        g.close()
