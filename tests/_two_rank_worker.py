"""Worker of tests/test_two_ranks_one_gpu.py: ONE RANK of a group that spans processes (ilm_group_create_rank), all ranks on GPU 0, the
exchange through tests/fake_rccl.cpp (ILM_RCCL_LIB).  Every rank checks what it holds against the single-context frame it renders itself.
    python tests/_two_rank_worker.py <rank> <world> <id file>"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from illuminant_amd import abi, native, scenes, sharding          # noqa: E402
from tests.test_lighting_gpu import small_scene                    # noqa: E402

AMBIENT = (0.05, 0.06, 0.07, 1.0)


def unique_id(rank, path):
    if rank == 0:
        uid = native.Group.unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(path + ".tmp", path)
        return uid
    deadline = time.time() + 120
    while time.time() < deadline:
        if os.path.exists(path) and os.path.getsize(path) == 128:
            return open(path, "rb").read()
        time.sleep(0.01)
    raise SystemExit("no id")


def native_strips(glm):
    """the table as the LIBRARY holds it (ilm_group_lightmap_strip), not the binding's copy"""
    import ctypes as C
    b, e = C.c_int32(), C.c_int32()
    out = []
    for r in range(glm.group.world):
        native.check(native.lib().ilm_group_lightmap_strip(glm.handle, r, C.byref(b), C.byref(e), None))
        out.append((int(b.value), int(e.value)))
    return out


def main():
    rank, world, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    layout, atlas, dfu, lights, w, h = small_scene(abi.SDF_FP16, n_lights=20, width=208, height=176)
    env = scenes.environment()
    g = native.Group.rank(0, rank, world, unique_id(rank, path))
    assert (g.n_local, g.world, g.first_rank) == (1, world, rank) and g.comm_ranks() == world
    c = g.contexts[0]
    sdf = native.DistanceFieldTexture(c, atlas, abi.SDF_FP16)
    # the single-context frame, by this rank itself
    lm = native.Lightmap(c, w, h, abi.LIGHTMAP_HALF4)
    st = native.render_sphere_lights(c, lights, env, dfu, None, sdf, AMBIENT, lm, want_stats=True)
    want = lm.download()
    lm.close()
    # 1. equal slots: one in-place all-gather
    glm = native.GroupLightmap(g, w, h, abi.LIGHTMAP_HALF4)
    part = g.render_sphere_lights(lights, env, dfu, None, [sdf], AMBIENT, glm, native.GATHER_RCCL, want_stats=True)
    g.sync()
    assert np.array_equal(glm.download(0).view(np.uint16), want.view(np.uint16)), "rank %d: equal slots" % rank
    # every rank's strip statistics add up to the frame's
    import struct
    tot = [sum(struct.unpack("<3Q", b)[k] for b in g.host_all_gather(struct.pack("<3Q", part.SdfSamples, part.PixelLightPairs, part.TracedPairs))) for k in range(3)]
    assert tuple(tot) == (st.SdfSamples, st.PixelLightPairs, st.TracedPairs), (tot, st.SdfSamples)
    # 2. cost-balanced strips: the range exchange (ncclSend / ncclRecv), serial and asynchronous over a ring of two
    strips = sharding.balanced_row_strips(h, world, lights)
    glm.set_strips(strips)
    assert glm.strips == [tuple(s) for s in strips]
    for m in glm.members:
        m.clear()
    g.render_sphere_lights(lights, env, dfu, None, [sdf], AMBIENT, glm, native.GATHER_RCCL)
    g.sync()
    assert np.array_equal(glm.download(0).view(np.uint16), want.view(np.uint16)), "rank %d: balanced strips" % rank
    glm_b = native.GroupLightmap(g, w, h, abi.LIGHTMAP_HALF4)
    glm_b.set_strips(strips)
    ring = (glm, glm_b)
    for m in glm.members + glm_b.members:
        m.clear()
    for k in range(5):
        g.render_sphere_lights(lights, env, dfu, None, [sdf], AMBIENT, ring[k & 1], native.GATHER_RCCL | native.GATHER_ASYNC)
    for x in ring:
        x.wait()
    g.sync()
    for x in ring:
        assert np.array_equal(x.download(0).view(np.uint16), want.view(np.uint16)), "rank %d: asynchronous exchange" % rank
    # 2b. the store mode across processes: the other ranks' buffers mapped through IPC handles, the kernel's final store writes this
    #     rank's strip into every rank's copy, the fence is a tiny collective; composite call and host-driven form, then disarmed
    for m in glm.members:
        m.clear()
    g.sync(); g.host_all_gather(b"\0" * 8)
    g.render_sphere_lights(lights, env, dfu, None, [sdf], AMBIENT, glm, native.GATHER_STORE)
    g.sync(); g.host_all_gather(b"\0" * 8)
    assert np.array_equal(glm.download(0).view(np.uint16), want.view(np.uint16)), "rank %d: store mode, composite call" % rank
    glm.store_mode(True)
    for m in glm.members:
        m.clear()
    g.sync(); g.host_all_gather(b"\0" * 8)
    b, e = glm.strips[rank]
    for _ in range(3):
        glm.gather(native.GATHER_STORE)
        native.render_sphere_lights(c, lights, env, dfu, None, sdf, AMBIENT, glm.members[0], b, e)
        glm.gather(native.GATHER_STORE)
    g.sync(); g.host_all_gather(b"\0" * 8)
    assert np.array_equal(glm.download(0).view(np.uint16), want.view(np.uint16)), "rank %d: store mode, host-driven" % rank
    glm.store_mode(False)
    # 2c. arming PROVES the mappings (every rank finds every other rank's stamp in its own buffer) -- negative control: rank 0 of the
    #     library is told to address memory of its own instead of rank 1's buffer; EVERY rank must refuse, nobody arms, the buffer is
    #     left as it was; a lightmap made afterwards arms, and the first one re-arms on the mappings it already has
    glm2 = native.GroupLightmap(g, w, h, abi.LIGHTMAP_HALF4)
    glm2.members[0].upload(want)
    os.environ["ILM_EXP_IPC_BOGUS_MAPPING"] = "1"
    try:
        glm2.store_mode(True)
        raise SystemExit("rank %d: armed over a mapping that addresses other memory" % rank)
    except native.IlluminantError as refusal:
        assert refusal.code == abi.ERR_STATE and "stamp" in str(refusal), refusal
    del os.environ["ILM_EXP_IPC_BOGUS_MAPPING"]
    assert np.array_equal(glm2.download(0).view(np.uint16), want.view(np.uint16)), "rank %d: the proof left its stamps in the frame" % rank
    with_error = None
    try:
        glm2.gather(native.GATHER_STORE)
    except native.IlluminantError as refusal:
        with_error = refusal
    assert with_error is not None and with_error.code == abi.ERR_STATE              # not armed
    glm2.store_mode(True)                                                            # the same lightmap, mapped afresh: fine now
    assert np.array_equal(glm2.download(0).view(np.uint16), want.view(np.uint16))
    glm2.store_mode(False)
    glm2.close()
    glm.store_mode(True)
    for m in glm.members:
        m.clear()
    g.sync(); g.host_all_gather(b"\0" * 8)
    glm.gather(native.GATHER_STORE)
    native.render_sphere_lights(c, lights, env, dfu, None, sdf, AMBIENT, glm.members[0], b, e)
    glm.gather(native.GATHER_STORE)
    g.sync(); g.host_all_gather(b"\0" * 8)
    assert np.array_equal(glm.download(0).view(np.uint16), want.view(np.uint16)), "rank %d: store mode, re-armed" % rank
    glm.store_mode(False)
    # 3. ilm_group_lightmap_set_strips is ALWAYS a collective: ranks that disagree, a rank with a malformed table, a rank that resets while the
    #    others install -- every rank fails (nobody hangs, nobody installs), the table stays what it was
    before = native_strips(glm)
    other = [(0, 16), (16, h)] if world == 2 else [(0, 16)] + [(16 * i, 16 * (i + 1)) for i in range(1, world - 1)] + [(16 * (world - 1), h)]
    for mine in ((other if rank == 0 else strips), ([(0, 8), (8, h)] + [(h, h)] * (world - 2) if rank == world - 1 else strips), (None if rank == 0 else strips)):
        try:
            glm.set_strips(mine)
            raise SystemExit("rank %d: a table the ranks disagree on was installed" % rank)
        except native.IlluminantError as e:
            assert e.code in (abi.ERR_STATE, abi.ERR_INVALID_ARGUMENT), e
        assert native_strips(glm) == before, "a failed installation changed the table"
    glm.set_strips(strips)           # and agreement still works afterwards
    # 4. the liveness table of a sharded system: chunk c on rank c mod world
    cs, total_chunks = 16, 5
    eng = native.Engine(c, cs, scenes.randomness_table(7))
    sysm = native.System(eng)
    d = abi.StepDesc(); d.FirstChunk, d.ChunkCount = 0, -1
    d.System = scenes.system_uniforms(cs, life_decay=20.0); d.Update = abi.UpdateParams.default(); d.UpdateMode = abi.UPDATE_POSITIONS; d.Flags = abi.STEP_COUNT_LIVE
    expected = []
    for ch in range(total_chunks):
        pos, vel, attr = scenes.make_particles(70 + ch, cs * cs, dead_fraction=0.1 * ch, life=(0.01, 0.8))
        if ch % world == rank:
            k = sysm.add_chunk()
            sysm.upload(k, abi.PLANE_POSITION, pos); sysm.upload(k, abi.PLANE_VELOCITY, vel); sysm.upload(k, abi.PLANE_ATTRIBUTES, attr)
    sysm.step(d)
    counts = g.live_counts([sysm], total_chunks)
    mine = sysm.step_counts()
    for i, ch in enumerate(range(rank, total_chunks, world)):
        assert counts[ch] == mine[i]
    everyone = g.host_all_gather(counts.astype("<u4").tobytes())
    assert all(b == everyone[0] for b in everyone), "the ranks hold different liveness tables"
    assert 0 < int(counts.sum()) < total_chunks * cs * cs
    # 5. particle lights across ranks (SURVEY 8f-3 + 8e row P): chunk ch lives on rank ch % world; ilm_group_gather_chunks makes Pos+Life and
    #    RenderColor whole in every rank's gathered system (one group of ncclSend / ncclRecv per call), the rank lights ITS strip with EVERY
    #    rank's particles, the strips are exchanged: every rank holds the frame one context renders from the whole table, bit for bit
    from tests import lights_common as lc
    from tests.test_lights_ext_gpu import particle_scene, small_field
    pw, ph, pcs, pchunks = 160, 112, 16, 5
    patlas, pdfu = small_field()
    penv = scenes.environment()
    chunks = particle_scene(pcs, pchunks, pw, ph)
    pparams = lc.particle_light_params(3.0, 30.0, (0.9, 0.8, 0.7, 0.6), casts_shadows=True)
    peng = native.Engine(c, pcs, scenes.randomness_table(7))
    whole, mine_, gathered = native.System(peng), native.System(peng), native.System(peng)
    for ch in range(pchunks):
        for target in ([whole] + ([mine_] if ch % world == rank else [])):
            k = target.add_chunk()
            target.upload(k, abi.PLANE_POSITION, chunks[ch][0]); target.upload(k, abi.PLANE_RENDER_COLOR, chunks[ch][3])
        gathered.add_chunk()
    psdf = native.DistanceFieldTexture(c, patlas)
    plm = native.Lightmap(c, pw, ph, abi.LIGHTMAP_FLOAT4)
    native.render_sphere_lights(c, None, penv, pdfu, None, psdf, AMBIENT, plm)
    pst = native.render_particle_lights(c, whole, pparams, penv, pdfu, None, psdf, plm, want_stats=True)
    pwant = plm.download()
    plm.close()
    pglm = native.GroupLightmap(g, pw, ph, abi.LIGHTMAP_FLOAT4)
    g.gather_chunks([mine_], [gathered], pchunks, 0, 4, native.GATHER_RCCL)
    g.gather_chunks([mine_], [gathered], pchunks, 12, 4, native.GATHER_RCCL)
    pb, pe = pglm.strips[rank]
    native.render_sphere_lights(c, None, penv, pdfu, None, psdf, AMBIENT, pglm.members[0], pb, pe)
    ps_ = native.render_particle_lights(c, gathered, pparams, penv, pdfu, None, psdf, pglm.members[0], row_begin=pb, row_end=pe, want_stats=True)
    pglm.gather(native.GATHER_RCCL)
    g.sync()
    assert np.array_equal(pglm.download(0), pwant), "rank %d: particle lights across ranks" % rank
    for ch in range(pchunks):
        assert np.array_equal(gathered.download(ch, abi.PLANE_POSITION), chunks[ch][0]) and np.array_equal(gathered.download(ch, abi.PLANE_RENDER_COLOR), chunks[ch][3])
    pairs = sum(struct.unpack("<Q", b)[0] for b in g.host_all_gather(struct.pack("<Q", ps_.PixelLightPairs)))
    assert pairs == pst.PixelLightPairs and pairs > 1000, (pairs, pst.PixelLightPairs)
    pglm.close(); psdf.close(); whole.close(); mine_.close(); gathered.close(); peng.close()
    sysm.close(); eng.close()
    glm_b.close(); glm.close(); sdf.close(); g.close()
    print("rank %d of %d ok" % (rank, world))


if __name__ == "__main__":
    main()
