"""Randomised parity sweep (not part of the test suite): many small lighting scenes and particle steps with fresh seeds, HIP path vs the
CPU oracle.  A seed FAILS (and the exit code is 1) when integer statistics, live counts or the liveness mask differ, when a life value is not
bit-identical to the oracle's, or when a float is outside the criterion of tests/util.py: |got - want| <= 1e-4 |want| + 1e-5 * (largest |want|
of the same component of the same plane).  The report also says how many elements needed the absolute floor at all (pure 1e-4 relative).
    python tools/fuzz_parity.py [first_seed] [count]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from illuminant_amd import abi, native, scenes
from oracle import oracle
from tests import fuzz_scenes

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = native.Context(0)
P, V, A, RC, RD = abi.PLANE_POSITION, abi.PLANE_VELOCITY, abi.PLANE_ATTRIBUTES, abi.PLANE_RENDER_COLOR, abi.PLANE_RENDER_DATA


def rel_err(got, want):
    got = np.asarray(got, np.float64); want = np.asarray(want, np.float64)
    ok = np.isfinite(want) & np.isfinite(got)
    if not ok.any():
        return 0.0
    scale = max(1.0, float(np.abs(want[ok]).max()))
    return float((np.abs(got[ok] - want[ok]) / np.maximum(np.abs(want[ok]), 1e-4 * scale * 1e4 * 1e-4 + 1e-30 + 1e-4 * scale)).max())


bad_light, bad_step, worst_l, worst_s = [], [], 0.0, 0.0
explained = []          # (seed, slot, components): float misses that ARE the reference's own cancellation (see cancellation_explains)


def cancellation_explains(seed, cs, rnd, d, got, failing):
    """r06: is every float of this seed's step that fell outside the criterion the known family -- velocity / position of a particle BORN in
    this step, beside a physical attractor of the step's Gravity whose radius all but cancels the particle's squared distance
    (Gravity.fx:44-47: strength / max(d^2 - radius, 0.001)), with the device's velocity INSIDE the envelope of the oracle's own answers
    for +-3 ulp of its own post-spawn position (the spawn formula's sin / cos / acos differ by that much between OCML and glibc)?
    The same facts tests/test_fuzz_regressions_gpu.py asserts of the named seeds.  failing: {(chunk, plane, slot, component)}."""
    import copy, itertools
    from tests import _variant_worker as vw
    if not failing or d.SpawnCount != 1 or d.OpCount < 1 or d.Ops[0].Type != abi.OP_GRAVITY:
        return None
    sp = d.Spawns[0].Params
    slots = {f[2] for f in failing}
    if len(slots) != 1 or any(f[0] != 1 or f[1] not in (0, 1, 4) or f[3] > 2 for f in failing):      # one slot of chunk 1: position, velocity (and the render data made of it)
        return None
    slot = slots.pop()
    if not (sp.ChunkSizeAndIndices[1] <= slot <= sp.ChunkSizeAndIndices[2]):
        return None
    cs_, rnd_, spawned, d0 = vw.post_spawn_state(seed)
    p0 = spawned[1][0][slot, :3].astype(np.float64)
    g = d.Ops[0].u.Gravity
    near = False
    for k_ in range(int(g.AttractorCount)):
        ar = g.AttractorRadiusesAndStrengths[k_]
        if ar[2] < 0.5:
            d2 = float(((np.array(list(g.AttractorPositions[k_])[:3], np.float64) - p0) ** 2).sum())
            near = near or (0.0 < d2 - float(ar[0]) < 0.12 * d2)
    if not near:
        return None
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    for signs in itertools.product(range(-3, 4), repeat=3):
        trial = [[a.copy() for a in c] for c in spawned]
        for axis, sg in enumerate(signs):
            trial[1][0][slot, axis] += np.float32(sg) * np.spacing(np.abs(trial[1][0][slot, axis]))
        one = [trial[1]]
        d1 = copy.copy(d0); d1.FirstChunk, d1.ChunkCount = 0, -1
        oracle.step(one, cs, rnd, d1)
        v = one[0][1][slot, :3].astype(np.float64)
        lo, hi = np.minimum(lo, v), np.maximum(hi, v)
    gv = got[1][1][slot, :3].astype(np.float64)
    slack = 1e-4 * np.maximum(np.abs(lo), np.abs(hi)) + 1e-6
    if not ((gv >= lo - slack) & (gv <= hi + slack)).all():
        return None
    return (seed, int(slot), sorted({(f[1], f[3]) for f in failing}))

group_scenes = 0
bad_float, floor_needed, elements_compared = [], 0, 0
bad_gbuffer, gbuffer_texels = [], 0
worst_where = None
bad_collision, collision_steps, collision_elements, collided_total = [], 0, 0, 0
bad_plight, particle_light_scenes, particle_light_pairs, worst_pl = [], 0, 0, 0.0
import time as _time
_t0, _budget = _time.time(), float(os.environ.get("ILM_FUZZ_SECONDS", "0") or 0)      # (a sweep under a time limit: stop drawing seeds, report what was done)
for seed in range(first, first + count):
    if _budget > 0 and _time.time() - _t0 > _budget:
        count = seed - first
        break
    rng = np.random.default_rng(seed)
    # ---- lighting: random field, lights, G-buffer normals, both SDF formats -------------------------------------------
    L_ = fuzz_scenes.draw_lighting(rng, seed)          # (tests/fuzz_scenes.py: the seed's draws, shared with tests/test_fuzz_regressions_gpu.py)
    w, h, fmt, layout, obstacles, dfu, lights = L_["w"], L_["h"], L_["fmt"], L_["layout"], L_["obstacles"], L_["dfu"], L_["lights"]
    atlas = scenes.build_sdf_atlas(layout, obstacles, fmt=fmt)
    env = scenes.environment()
    sdf = native.DistanceFieldTexture(ctx, atlas, fmt)
    lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
    # (r04) the light split: chosen per launch / forced to 1, 2, 4, 8 workgroups per tile, and the frame in one launch or in three strips that
    # start on arbitrary rows -- none of it may change a statistic, and the strips must reproduce the one-launch frame bit for bit
    split = (0, 1, 2, 4, 8)[seed % 5]
    ctx.set_light_split(split)
    st = native.render_sphere_lights(ctx, lights, env, dfu, None, sdf, (0.05, 0.05, 0.05, 1.0), lm, want_stats=True)
    got = lm.download()
    if seed % 3 == 0:
        cuts = sorted(set([0, h] + [int(v) for v in np.random.default_rng(seed + 99).integers(1, h, 2)]))
        ctx.set_light_split((8, 0, 2, 4, 1)[seed % 5])
        lm.clear()
        for b0, b1 in zip(cuts[:-1], cuts[1:]):
            native.render_sphere_lights(ctx, lights, env, dfu, None, sdf, (0.05, 0.05, 0.05, 1.0), lm, b0, b1)
        if not np.array_equal(lm.download(), got):
            bad_light.append((seed, "strips under another split differ from the one-launch frame", cuts))
    ctx.set_light_split(0)
    # (r05) every 6th scene also through a GROUP of 2-4 members on this device under one of the exchange modes -- peer copies, the
    # store mode (the kernel's final store into every member's copy), the asynchronous exchange over a ring of two lightmaps, and a
    # sibling context reading the first one's field: every copy of the frame must equal the one-launch frame bit for bit
    if seed % 6 == 0:
        grng = np.random.default_rng(seed + 31337)
        members = int(grng.integers(2, 5))
        mode = int(grng.integers(0, 4))
        group_scenes += 1
        if mode == 3:
            sib = ctx.sibling()
            lm2 = native.Lightmap(sib, w, h, abi.LIGHTMAP_FLOAT4)
            native.render_sphere_lights(sib, lights, env, dfu, None, sdf, (0.05, 0.05, 0.05, 1.0), lm2)
            if not np.array_equal(lm2.download(), got):
                bad_light.append((seed, "a sibling context's frame through the borrowed field differs"))
            lm2.close(); sib.close()
        else:
            g = native.Group([0] * members)
            gsdfs = [native.DistanceFieldTexture(c, atlas, fmt) for c in g.contexts]
            ring = [native.GroupLightmap(g, w, h, abi.LIGHTMAP_FLOAT4) for _ in range(2 if mode == 2 else 1)]
            gather = (native.GATHER_PEER, native.GATHER_STORE, native.GATHER_PEER | native.GATHER_ASYNC)[mode]
            for k in range(len(ring) * 2):
                g.render_sphere_lights(lights, env, dfu, None, gsdfs, (0.05, 0.05, 0.05, 1.0), ring[k % len(ring)], gather)
            for glm in ring:
                glm.wait()
            g.sync()
            for glm in ring:
                for i in range(members):
                    if not np.array_equal(glm.download(i), got):
                        bad_light.append((seed, "group of %d, exchange mode %d: member %d's frame differs" % (members, mode, i)))
            for glm in ring:
                glm.close()
            for x in gsdfs:
                x.close()
            g.close()
    want, ost = oracle.render_sphere_lights(lights, env, dfu, None, oracle.make_texture(atlas, fmt), (0.05, 0.05, 0.05, 1.0), w, h, 0, h, want_stats=True)
    # (r06) every 4th scene: PARTICLE lights over the same field (ParticleLight.fx:16-118) -- 1-3 chunks of 16^2 .. 64^2 slots, some of them
    # crowded onto one spot so that a tile's list overflows a batch (lighting.hip, BIG), random template (radius, ramp, falloff, AO,
    # specular, shadows), quad counts that cut chunks short.  The plain launch (cull, batches of 4 096) must equal the instrumented one
    # (no cull, batches of 1 024) bit for bit; counts and floats against the oracle.
    if seed % 4 == 1:
        from tests import lights_common as lc
        prng = np.random.default_rng(seed + 777)
        pcs = int(prng.choice([16, 32, 64])); pn = pcs * pcs
        pchunks, pquads = [], []
        for c in range(int(prng.integers(1, 4))):
            crowd = bool(prng.integers(0, 2))
            cx_, cy_ = float(prng.uniform(0, w)), float(prng.uniform(0, h))
            lo = (cx_ - 20, cy_ - 15, 1) if crowd else (-10, -10, 1)
            hi = (cx_ + 20, cy_ + 15, 25) if crowd else (w + 10, h + 10, 40)
            ppos, pvel, pattr = scenes.make_particles(seed * 7 + c, pn, pos_lo=lo, pos_hi=hi, dead_fraction=float(prng.uniform(0, 0.5)))
            prc = scenes.uniform(seed * 11 + c, (pn, 4), 0.0, 1.0).astype(np.float32)
            prc[:, :3] *= prc[:, 3:4]
            prc[scenes.uniform(seed * 13 + c, (pn,)) < 0.1, 3] = 0.0
            pchunks.append([ppos, pvel, pattr, prc, np.zeros((pn, 4), np.float32)])
            pquads.append(int(prng.integers(1, pn + 1)) if prng.integers(0, 3) == 0 else pn)
        pparams = lc.particle_light_params(float(prng.uniform(1, 6)), float(prng.uniform(4, 40)), (1.0, 0.9, 0.8, float(prng.uniform(0.02, 1.0))),
                                           casts_shadows=bool(prng.integers(0, 4) != 0), ao_radius=float(prng.choice([0.0, 5.0])), ao_opacity=float(prng.uniform(0, 1)),
                                           falloff_y=float(prng.choice([1.0, 0.6, 2.0])), spec=tuple(prng.uniform(0, 0.4, 3)) if prng.integers(0, 3) == 0 else (0, 0, 0),
                                           spec_power=float(prng.uniform(1, 4)), ramp_mode=int(prng.integers(0, 3)))
        peng = native.Engine(ctx, pcs, scenes.randomness_table(7)); psys = native.System(peng)
        for c, planes in enumerate(pchunks):
            psys.add_chunk()
            psys.upload(c, P, planes[0]); psys.upload(c, RC, planes[3])
        pframes, pst = [], None
        for want_stats in (True, False):
            plm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
            native.render_sphere_lights(ctx, None, env, dfu, None, sdf, (0.05, 0.06, 0.07, 1.0), plm)
            r_ = native.render_particle_lights(ctx, psys, pparams, env, dfu, None, sdf, plm, quad_counts=pquads, want_stats=want_stats)
            if want_stats: pst = r_
            pframes.append(plm.download()); plm.close()
        pwant = np.empty((h, w, 4), np.float32); pwant[...] = np.asarray((0.05, 0.06, 0.07, 1.0), np.float32)
        post = oracle.render_particle_lights(pchunks, pquads, pparams, env, dfu, None, oracle.make_texture(atlas, fmt), pwant, want_stats=True)
        particle_light_scenes += 1
        particle_light_pairs += int(post.PixelLightPairs)
        pe = float((np.abs(pframes[0] - pwant) / np.maximum(np.abs(pwant), 1.0)).max())
        worst_pl = max(worst_pl, pe)
        if not np.array_equal(pframes[0].view(np.uint32), pframes[1].view(np.uint32)):
            bad_plight.append((seed, "the plain frame differs from the instrumented one"))
        if (pst.SdfSamples, pst.PixelLightPairs, pst.TracedPairs) != (post.SdfSamples, post.PixelLightPairs, post.TracedPairs) or pe > 1e-4:
            bad_plight.append((seed, (pst.SdfSamples, pst.PixelLightPairs, pst.TracedPairs), (post.SdfSamples, post.PixelLightPairs, post.TracedPairs), pe))
        psys.close(); peng.close()
    lm.close(); sdf.close()
    e = float((np.abs(got - want) / np.maximum(np.abs(want), 1.0)).max())
    worst_l = max(worst_l, e)
    if (st.SdfSamples, st.PixelLightPairs, st.TracedPairs) != (ost.SdfSamples, ost.PixelLightPairs, ost.TracedPairs) or e > 1e-4:
        bad_light.append((seed, (st.SdfSamples, st.PixelLightPairs, st.TracedPairs), (ost.SdfSamples, ost.PixelLightPairs, ost.TracedPairs), e))
    # ---- the collision update through this seed's field, every other seed: both formats, the uniforms as the lighting path binds
    #      them and as the particle path does (DistanceFieldPacked1 = 0: the slice-0 form of the sampler) ------------------------------
    if seed % 2 == 0:
        crng = np.random.default_rng(seed + 424242)
        ccs = 64
        cn = ccs * ccs
        cpos, cvel, cattr = scenes.make_particles(seed + 5, cn, pos_lo=(-16, -16, 0), pos_hi=(272, 208, 60), dead_fraction=float(crng.uniform(0, 0.3)),
                                                  life=(0.01, 6.0), categories=(0.0, 2.0))
        csu = scenes.system_uniforms(ccs, friction=float(crng.uniform(0, 0.3)), max_velocity=float(crng.choice([90.0, 2048.0])), life_decay=float(crng.uniform(0, 2)),
                                     collision=(float(crng.uniform(30, 200)), float(crng.choice([0.0, 0.6, 1.2])), float(crng.uniform(0.1, 2.0)), float(crng.uniform(0, 0.2))))
        cfmt = abi.SDF_FP16 if (seed // 2) % 2 else abi.SDF_UNORM16
        catlas = atlas if cfmt == fmt else scenes.build_sdf_atlas(layout, obstacles, fmt=cfmt)
        cdfu = layout.uniforms(packed1=bool((seed // 4) % 2))
        cup = abi.UpdateParams.default()
        ceng = native.Engine(ctx, ccs, scenes.randomness_table(3)); csys = native.System(ceng); csys.add_chunk()
        csdf = native.DistanceFieldTexture(ctx, catlas, cfmt)
        csys.set_distance_field(csdf)
        for pl, data in ((P, cpos), (V, cvel), (A, cattr)):
            csys.upload(0, pl, data)
        csys.update(0, csu, cup, df=cdfu)
        cwant = [cpos.copy(), cvel.copy(), cattr.copy(), np.zeros((cn, 4), np.float32), np.zeros((cn, 4), np.float32)]
        oracle.update(cwant[0], cwant[1], cwant[2], cwant[3], cwant[4], ccs, csu, cup, df=cdfu, sdf=oracle.make_texture(catlas, cfmt))
        collision_steps += 1
        for kk, pl in enumerate((P, V, A, RC, RD)):
            g = csys.download(0, pl).astype(np.float64); wv = cwant[kk].astype(np.float64)
            if kk == 0:
                if not np.array_equal(g[:, 3] > 0, wv[:, 3] > 0) or not np.array_equal(g[:, 3].astype(np.float32).view(np.uint32), wv[:, 3].astype(np.float32).view(np.uint32)):
                    bad_collision.append((seed, "liveness / life not identical"))
            comp_scale = np.where(np.isfinite(wv), np.abs(wv), 0.0).max(axis=0)
            both_nan = np.isnan(g) & np.isnan(wv)
            outside = ~(np.abs(g - wv) <= 1e-5 * comp_scale[None, :] + 1e-4 * np.abs(wv)) & ~both_nan
            collision_elements += g.size
            if outside.any():
                i, j = np.argwhere(outside)[0]
                bad_collision.append((seed, "plane %d slot %d component %d: got %.9g want %.9g; %d element(s)" % (kk, i, j, g[i, j], wv[i, j], int(outside.sum()))))
        collided_total += int((cwant[1][:, 3] == 3.0).sum())
        csdf.close(); csys.close(); ceng.close()
    # ---- particles: random op list, spawner, chunk size ---------------------------------------------------------------
    cs, rnd, chunks, d = fuzz_scenes.draw_particle_step(rng, seed)
    n = cs * cs
    k = int(d.OpCount)
    eng = native.Engine(ctx, cs, rnd); sysm = native.System(eng)
    for c in range(2):
        sysm.add_chunk()
        for pl, a in ((P, chunks[c][0]), (V, chunks[c][1]), (A, chunks[c][2])):
            sysm.upload(c, pl, a)
    inputs = [(c[0].copy(), c[1].copy()) for c in chunks]
    initial = [[a.copy() for a in c] for c in chunks]
    sysm.step(d)
    want_counts = oracle.step(chunks, cs, rnd, d, want_counts=True)
    got_counts = sysm.step_counts()
    problem = not np.array_equal(got_counts, want_counts)
    seed_failing, seed_bad_float, seed_got = set(), [], [[None] * 5, [None] * 5]
    for c in range(2):
        gp = sysm.download(c, P)
        problem = problem or not np.array_equal(gp[:, 3] > 0, chunks[c][0][:, 3] > 0)
        if not np.array_equal(gp[:, 3].view(np.uint32), chunks[c][0][:, 3].view(np.uint32)):
            nan_both = np.isnan(gp[:, 3]) & np.isnan(chunks[c][0][:, 3])
            if not ((gp[:, 3].view(np.uint32) == chunks[c][0][:, 3].view(np.uint32)) | nan_both).all():
                bad_float.append((seed, c, "life not bit-identical"))
        for kk, pl in enumerate((P, V, A, RC, RD)):
            g32 = sysm.download(c, pl); seed_got[c][kk] = g32
            g = g32.astype(np.float64); wv = chunks[c][kk].astype(np.float64)
            # the suite's criterion (tests/util.py assert_close): per-component scale
            comp_scale = np.where(np.isfinite(wv), np.abs(wv), 0.0).max(axis=0)
            both_nan = np.isnan(g) & np.isnan(wv)
            err_abs = np.abs(g - wv)
            outside = ~(err_abs <= 1e-5 * comp_scale[None, :] + 1e-4 * np.abs(wv)) & ~both_nan
            elements_compared += g.size
            floor_needed += int((~(err_abs <= 1e-4 * np.abs(wv)) & ~both_nan).sum())
            if outside.any():
                i, j = np.argwhere(outside)[0]
                seed_bad_float.append((seed, c, "plane %d slot %d component %d: got %.9g want %.9g (component scale %.4g); %d element(s)"
                                       % (kk, i, j, g[i, j], wv[i, j], comp_scale[j], int(outside.sum()))))
                for (ii, jj) in np.argwhere(outside)[:64]:
                    seed_failing.add((c, kk, int(ii), int(jj)))
            ok = np.isfinite(wv) & np.isfinite(g)
            if ok.any():
                scale = max(1.0, float(np.abs(wv[ok]).max()))
                r = np.where(ok, np.abs(g - wv) / (np.abs(wv) + 1e-4 * scale), 0.0)
                if float(r.max()) > worst_s:
                    worst_s = float(r.max())
                    i, j = np.unravel_index(int(np.argmax(r)), r.shape)
                    worst_where = (seed, c, kk, int(i), int(j), float(g[i, j]), float(wv[i, j]), scale, cs, k, int(d.SpawnCount))
                    if os.environ.get("ILM_FUZZ_DUMP"):
                        print("worst so far:", worst_where)
                        print("  input position/life", inputs[c][0][i], "velocity", inputs[c][1][i])
                        print("  got  P", sysm.download(c, P)[i], "V", sysm.download(c, V)[i])
                        print("  want P", chunks[c][0][i], "V", chunks[c][1][i])
                        for o in range(k):
                            if d.Ops[o].Type == abi.OP_GRAVITY:
                                gp_ = d.Ops[o].u.Gravity
                                print("  gravity: count", gp_.AttractorCount, "max accel", gp_.MaximumAcceleration)
                                for a_ in range(int(gp_.AttractorCount)):
                                    ap = gp_.AttractorPositions[a_]; ar = gp_.AttractorRadiusesAndStrengths[a_]
                                    dist = float(np.linalg.norm(np.array(list(ap)) - inputs[c][0][i][:3]))
                                    print("    attractor", list(ap), "radius/strength/type", list(ar), "distance", dist)
                            else:
                                print("  op type", d.Ops[o].Type)
                        # the same step without its ops (spawn + update only): is the difference there before the ops run?
                        import copy
                        d0 = copy.copy(d); d0.OpCount = 0
                        sys0 = native.System(eng)
                        ref0 = [[a.copy() for a in cc] for cc in initial]
                        for cc in range(2):
                            sys0.add_chunk()
                            for pl_, a_ in ((P, initial[cc][0]), (V, initial[cc][1]), (A, initial[cc][2])):
                                sys0.upload(cc, pl_, a_)
                        sys0.step(d0)
                        oracle.step(ref0, cs, rnd, d0)
                        print("  without the ops: got  P", sys0.download(c, P)[i], "V", sys0.download(c, V)[i])
                        print("  without the ops: want P", ref0[c][0][i], "V", ref0[c][1][i])
                        sys0.close()
    if problem:
        bad_step.append((seed, list(got_counts), list(want_counts)))
    if seed_bad_float:
        why = None
        if not problem and len(seed_failing) < 64 and not any("life" in str(b[2]) for b in bad_float if b[0] == seed):
            try:
                why = cancellation_explains(seed, cs, rnd, d, [[np.asarray(a) for a in cc] for cc in seed_got], seed_failing)
            except Exception as e_:      # noqa: BLE001 -- an analysis that cannot run explains nothing
                why = None
        if why is not None:
            explained.append(why)
        else:
            bad_float.extend(seed_bad_float)
    sysm.close(); eng.close()
# ---- distance-field generation (exact culling!) and the particle rasteriser, every 4th seed (the oracle side is slower) ------------------
bad_field, bad_raster, field_texels, raster_worst, field_volume_scenes = [], [], 0, 0, 0
for seed in range(first, first + count, 4):
    rng = np.random.default_rng(seed + 77777)
    ext = (int(rng.integers(64, 300)), int(rng.integers(64, 240)))
    layout = scenes.DistanceFieldLayout(ext[0], ext[1], float(rng.uniform(32, 128)), int(rng.integers(3, 16)), float(rng.choice([1.0, 0.5, 0.3, 0.25])), 128)
    obs = scenes.random_obstructions(seed, int(rng.integers(1, 40)), ext, size_lo=float(rng.uniform(2, 10)), size_hi=float(rng.uniform(12, 90)), z_hi=float(rng.uniform(8, 80)),
                                     rotate=bool((seed // 4) % 2))      # every other scene unrotated: the kernel skips identity rotations
    arr = scenes.obstruction_array(obs)
    fmt = abi.SDF_FP16 if seed % 8 < 4 else abi.SDF_UNORM16
    d = scenes.render_desc(layout)
    triplets = list(range(0, layout.slice_count, 3))
    # every other scene also carries height volumes (DistanceToPolygon; the kernel culls a volume on the circle around its polygon) and a
    # random MaximumEncodedDistance: short reaches cull most texel / volume pairs, long ones none
    fvols, fpoly = None, None
    if (seed // 4) % 2 == 1:
        layout.maximum_encoded_distance = float(rng.choice([24.0, 64.0, 128.0, 300.0]))
        d = scenes.render_desc(layout)
        vlist = []
        for _ in range(int(rng.integers(1, 30))):
            cx, cy, rad = float(rng.uniform(0, ext[0])), float(rng.uniform(0, ext[1])), float(rng.uniform(2, 40))
            nv = int(rng.integers(1, 8))
            ang = np.sort(rng.uniform(0, 2 * np.pi, nv)); rr = rad * rng.uniform(0.3, 1.0, nv)
            vlist.append(([(float(cx + rr[k] * np.cos(ang[k])), float(cy + rr[k] * np.sin(ang[k]))) for k in range(nv)],
                          float(rng.uniform(0, 20)), float(rng.uniform(1, 60)), bool(rng.integers(0, 2))))
        fvols, fpoly = scenes.height_volume_arrays(vlist)
        field_volume_scenes += 1
    sdf = native.DistanceFieldTexture(ctx, None, fmt, size=(layout.atlas_width, layout.atlas_height))
    sdf.render_slices(d, triplets, arr, fvols, fpoly)
    got = sdf.download(); sdf.close()
    want = oracle.render_distance_field_slices(np.zeros((layout.atlas_height, layout.atlas_width, 4), np.uint16), fmt, d, triplets, arr, fvols, fpoly)
    field_texels += got.size
    if not np.array_equal(got, want):
        bad_field.append((seed, int((got != want).sum())))
    # rasteriser: random sprites, sometimes a bitmap
    cs, w, h = 32, int(rng.integers(40, 200)), int(rng.integers(30, 150))
    n = cs * cs
    pos = np.zeros((n, 4), np.float32)
    pos[:, 0] = rng.uniform(-10, w + 10, n); pos[:, 1] = rng.uniform(-10, h + 10, n); pos[:, 2] = rng.uniform(0, 6, n)
    pos[:, 3] = np.where(rng.random(n) < 0.3, 0.0, rng.uniform(0.1, 3.0, n))
    a = rng.uniform(0, 1, n); rgb = rng.uniform(0, 1, (n, 3))
    col = np.concatenate([rgb * a[:, None], a[:, None]], axis=1).astype(np.float32)
    rd = np.zeros((n, 4), np.float32); rd[:, 0] = rng.uniform(0, 9, n); rd[:, 1] = rng.uniform(-7, 20, n); rd[:, 3] = np.floor(rng.uniform(-1, 3, n))
    chunk = [pos, np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32), col, rd]
    textured = bool(rng.integers(0, 2))
    sheet = rng.uniform(0, 1, (8, 16, 4)).astype(np.float32) if textured else None
    params = scenes.rasterize_params(size=(float(rng.uniform(0.3, 2)), float(rng.uniform(0.3, 2))), rounded=bool(rng.integers(0, 2)),
                                     blend=int(rng.integers(0, 2)), z_to_y=float(rng.uniform(0, 1)), size_from_z=float(rng.uniform(0, 0.1)),
                                     texture_size=(16, 8) if textured else None, size_px=(4.0, 4.0) if textured else None,
                                     bilinear=bool(rng.integers(0, 2)), animation_rate=(float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2))),
                                     column_from_velocity=bool(rng.integers(0, 2)), row_from_velocity=bool(rng.integers(0, 2)),
                                     dithered_opacity=bool(rng.integers(0, 3) == 0))
    eng = native.Engine(ctx, cs, scenes.randomness_table(1)); sysm = native.System(eng); sysm.add_chunk()
    sysm.upload(0, P, pos); sysm.upload(0, RC, col); sysm.upload(0, RD, rd)
    if textured: sysm.set_bitmap(sheet)
    lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4); lm.clear((0.1, 0.2, 0.3, 1.0))
    live, pairs, shaded = native.render_particles(sysm, params, lm, want_stats=True)
    gimg = lm.download(); lm.close(); sysm.close(); eng.close()
    wimg = np.zeros((h, w, 4), np.float32); wimg[:] = (0.1, 0.2, 0.3, 1.0)
    wimg, (olive, oshaded) = oracle.render_particles([chunk], params, w, h, image=wimg, bitmap=sheet)
    err = np.abs(gimg.astype(np.float64) - wimg).max(axis=-1)
    outliers = int((err > 1e-4 * np.maximum(1.0, np.abs(wimg).max(axis=-1))).sum())
    raster_worst = max(raster_worst, outliers)
    if live != olive or outliers > (40 if textured else 8) or abs(shaded - oshaded) > 8:
        bad_raster.append((seed, live, olive, shaded, oshaded, outliers, textured))

    # ---- G-buffer from meshes: 2.5D volumes + billboards (every 2nd seed) ----------------------------------------------
    if seed % 2 == 0:
        gw, gh = int(rng.integers(40, 200)), int(rng.integers(30, 140))
        k = float(rng.choice([0.0, 0.4, 0.6, 1.0]))
        two = bool(rng.integers(0, 4) != 0)
        vols = []
        for v in range(int(rng.integers(1, 12))):
            nv = int(rng.integers(3, 8))
            ang = np.sort(rng.uniform(0, 2 * np.pi, nv)); rad = rng.uniform(4, 40, nv)
            cx0, cy0 = rng.uniform(0, gw), rng.uniform(0, gh)
            vols.append(([(float(np.float32(cx0 + rad[q] * np.cos(ang[q]))), float(np.float32(cy0 + rad[q] * np.sin(ang[q])))) for q in range(nv)],
                         float(rng.uniform(-4, 10)), float(rng.uniform(1, 70)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))))
        vols.sort(key=lambda t: (t[1] + t[2]) * (-1 if two else 1))
        top = np.concatenate([scenes.top_face_mesh(t[0], t[1], t[2], t[3]) for t in vols])
        front = np.concatenate([scenes.front_face_mesh(t[0], t[1], t[2], t[4]) for t in vols] + [np.zeros((0, 9), np.float32)]) if two else None
        boards, kinds, texs, handles = [], [], [], []
        for q in range(int(rng.integers(0, 7))):
            kind = int(rng.integers(0, 2))
            x, y = rng.uniform(-10, gw), rng.uniform(-10, gh)
            bd = dict(screen_bounds=((x, y), (x + rng.uniform(2, 60), y + rng.uniform(2, 60))), type=kind, normal=tuple(rng.uniform(-1, 1, 3)),
                      cylinder_factor=float(rng.choice([0.0, 0.5, 1.0])), data_scale=(None if rng.integers(0, 2) else float(rng.uniform(0.1, 6))),
                      static_lighting_only=bool(rng.integers(0, 2)), world_offset=tuple(rng.uniform(-2, 2, 3)))
            if kind == abi.BILLBOARD_GBUFFER_DATA: bd["world_elevation"] = float(rng.uniform(0, 30))
            boards.append(bd); kinds.append(kind)
            which = int(rng.integers(0, 4))
            t = rng.uniform(0, 1, (int(rng.integers(1, 12)), int(rng.integers(1, 12)), 4))
            t[..., :2] = 0.5 + (t[..., :2] - 0.5) * 0.7          # keep most tangent normals inside the unit disc; some NaNs stay
            tex = None if which == 3 else (np.round(t * 255).astype(np.uint8) if which == 0 else t.astype(np.float16) if which == 1 else t.astype(np.float32))
            texs.append(tex)
            if tex is None:
                handles.append(None)
            else:
                fm = {np.dtype(np.uint8): abi.LIGHTMAP_RGBA8, np.dtype(np.float16): abi.LIGHTMAP_HALF4, np.dtype(np.float32): abi.LIGHTMAP_FLOAT4}[tex.dtype]
                hl = native.Lightmap(ctx, tex.shape[1], tex.shape[0], fm); hl.upload(tex); handles.append(hl)
        bbv = scenes.billboard_vertices(boards, 0.0, k)
        gd = scenes.gbuffer_mesh_desc(ground_z=float(rng.uniform(-2, 6)), viewport_position=tuple(rng.uniform(-8, 8, 2)), viewport_scale=tuple(rng.uniform(0.5, 2.0, 2)),
                                      z_to_y=k, render_scale=tuple(rng.uniform(0.5, 2, 2)), extent_z=float(rng.uniform(16, 128)),
                                      self_occlusion_hack=float(rng.uniform(0, 1)), z_self_occlusion_hack=float(rng.uniform(0, 3)), two_point_five_d=two,
                                      render_ground_plane=bool(rng.integers(0, 5) != 0), enable_ground_shadows=bool(rng.integers(0, 2)))
        gfmt = abi.GBUFFER_HALF4 if (seed // 2) % 2 else abi.GBUFFER_FLOAT4
        gbt = native.GBufferTexture(ctx, None, gfmt, size=(gw, gh))
        gbt.render_meshes(gd, top, front, bbv, [(handles[q], q, 1, kinds[q]) for q in range(len(kinds))])
        ggot = gbt.download(); gbt.close()
        for hl in handles:
            if hl is not None: hl.close()
        gwant = oracle.render_gbuffer_meshes(gw, gh, gd, top, front, bbv, [(q, 1, kinds[q]) for q in range(len(kinds))], texs)
        if gfmt == abi.GBUFFER_HALF4:
            ggot = ggot.view(np.float16).astype(np.float32); gwant = gwant.astype(np.float16).astype(np.float32); gtol = 2.0 ** -10
        else:
            gtol = 1e-6
        gbuffer_texels += gw * gh
        exact = np.array_equal(ggot[..., 1:], gwant[..., 1:], equal_nan=True) and np.array_equal(np.isnan(ggot[..., 0]), np.isnan(gwant[..., 0]))
        d0 = np.abs(ggot[..., 0] - gwant[..., 0])
        if not exact or (np.isfinite(d0).any() and np.nanmax(d0) > gtol):
            bad_gbuffer.append((seed, gw, gh, two, len(vols), len(kinds), exact, float(np.nanmax(d0)) if np.isfinite(d0).any() else 0.0))

print("seeds %d..%d" % (first, first + count - 1))
print("G-buffer meshes: %d scenes with a differing texel of %d (%.2f M texels; channels 1-3 bit-equal, channel 0 within atan2's last bits)"
      % (len(bad_gbuffer), len(range(first + (first % 2), first + count, 2)), gbuffer_texels / 1e6))
for b in bad_gbuffer[:10]: print("   ", b)
print("field generation: %d scenes with differing codes of %d, %d of them with height volumes (%.1f M texel channels compared)" % (
    len(bad_field), len(range(first, first + count, 4)), field_volume_scenes, field_texels / 1e6))
for b in bad_field[:10]: print("   ", b)
print("rasteriser: %d scenes out of bounds; most edge pixels that flipped in one frame: %d" % (len(bad_raster), raster_worst))
for b in bad_raster[:10]: print("   ", b)
print("lighting: %d scenes with differing statistics or > 1e-4 error; worst relative error %.3g (%d of the scenes also through a group / a sibling context: peer, store, asynchronous exchange)" % (len(bad_light), worst_l, group_scenes))
for b in bad_light[:10]: print("   ", b)
print("particle lights: %d scenes with differing statistics, > 1e-4 error or a plain frame that is not the instrumented one, of %d (%.1f M pixel.light pairs); worst relative error %.3g"
      % (len(bad_plight), particle_light_scenes, particle_light_pairs / 1e6, worst_pl))
for b in bad_plight[:10]: print("   ", b)
print("collision update: %d problems in %d steps (%.1f M elements; %d particles bounced or were redirected); liveness and life bit-identical, floats by the suite's criterion"
      % (len(bad_collision), collision_steps, collision_elements / 1e6, collided_total))
for b in bad_collision[:10]: print("   ", b)
print("particles: %d steps with differing live counts / liveness; worst error relative to (|want| + 1e-4 scale) %.3g" % (len(bad_step), worst_s))
for b in bad_step[:10]: print("   ", b)
print("worst particle element (seed, chunk, plane, slot, component, got, want, plane scale, chunk size, ops, spawns):", worst_where)
print("particle floats: %d failures of the suite's criterion (1e-4 relative + 1e-5 x component scale, life bit-identical) in %.1f M elements; "
      "%d elements (%.2g of all) are outside a PURE 1e-4 relative bound, i.e. needed the absolute floor" %
      (len(bad_float), elements_compared / 1e6, floor_needed, floor_needed / max(elements_compared, 1)))
for b in bad_float[:10]: print("   ", b)
print("explained by the reference's own cancellation (a newborn particle beside a physical attractor with 0 < d^2 - radius < 0.12 d^2, the device's velocity inside "
      "the oracle's envelope for +-3 ulp of its own post-spawn position; not failures): %d" % len(explained))
for b in explained[:10]: print("   ", b)
failed = bool(bad_field or bad_raster or bad_light or bad_step or bad_float or bad_gbuffer or bad_collision or bad_plight)
print("FUZZ %s" % ("FAILED" if failed else "PASSED"))
sys.exit(1 if failed else 0)
