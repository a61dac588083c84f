"""Quick on-GPU timing probe of the two hot kernels (development aid, not the contract bench)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from illuminant_amd import abi, native, scenes

def particles(ctx, cs=256, n_chunks=16, steps=50, spawn=True, ops="gn", update=True):
    rnd = scenes.randomness_table(7)
    eng = native.Engine(ctx, cs, rnd); sysm = native.System(eng)
    n = cs * cs
    for c in range(n_chunks):
        sysm.add_chunk()
        pos, vel, attr = scenes.make_particles(10 + c, n, pos_lo=(0, 0, 0), pos_hi=(1920, 1080, 32), life=(50.0, 90.0))
        sysm.upload(c, abi.PLANE_POSITION, pos); sysm.upload(c, abi.PLANE_VELOCITY, vel); sysm.upload(c, abi.PLANE_ATTRIBUTES, attr)
    tgt = sysm.add_chunk() if spawn else -1
    d = abi.StepDesc(); d.FirstChunk, d.ChunkCount = 0, -1
    d.System = scenes.system_uniforms(cs, friction=0.02, max_velocity=2048.0, life_decay=0.01)
    d.Update = abi.UpdateParams.default(); d.UpdateMode = abi.UPDATE_POSITIONS if update else abi.UPDATE_NONE
    k = 0
    for o in ops:
        if o == "g":
            d.Ops[k].Type = abi.OP_GRAVITY
            d.Ops[k].u.Gravity = scenes.gravity_params([((400., 300., 0.), 70., 600., 1), ((1500., 300., 0.), 150., 900., 1),
                                                        ((400., 800., 0.), 200., 1200., 1), ((1500., 800., 0.), 100., 1500., 1)], 1024.0)
        elif o == "n":
            d.Ops[k].Type = abi.OP_NOISE
            d.Ops[k].u.Noise = scenes.noise_params(scenes.area_none(), (0.37 * 253, 0.81 * 127), (0.12 * 253, 0.55 * 127), 0.35)
        elif o == "f":
            d.Ops[k].Type = abi.OP_FMA
            d.Ops[k].u.FMA = scenes.fma_params(scenes.area_none(), velocity_multiply=(0.99, 0.99, 1.0))
        k += 1
    d.OpCount = k
    if os.environ.get("ILM_PROBE_COUNT"):        # what ParticleSystem.Update asks for every frame: the live count of every chunk
        d.Flags = abi.STEP_COUNT_LIVE
    import ctypes
    descs = []
    first = 0
    for i in range(steps + 5):
        dd = abi.StepDesc(); ctypes.memmove(ctypes.byref(dd), ctypes.byref(d), ctypes.sizeof(d))
        if spawn:
            dd.SpawnCount = 1; dd.Spawns[0].ChunkIndex = tgt
            dd.Spawns[0].Params = scenes.spawn_params(cs, first, first + 1091, first, (0.42 * 253, 0.77 * 127),
                position=((960, 540, 0), (900, 450, 0), (0, 0, 0), 1), velocity=((0, 0, 0), (60, 60, 60), (0, 0, 0), 1), life=(50.0, 2.7, 0))
            first = (first + 1092) % (n - 1092)
        descs.append(dd)
    it = iter(descs)
    def step():
        sysm.step(next(it))
    for _ in range(5): step()
    ctx.sync()
    t0 = time.perf_counter(); ctx.timer_start()
    for _ in range(steps): step()
    ms = ctx.timer_stop(); wall = (time.perf_counter() - t0) * 1e3
    slots = n * (n_chunks + (1 if spawn else 0))
    live = n * n_chunks
    print("particles cs=%d chunks=%d ops=%r update=%d spawn=%d: %.4f ms/step gpu (%.3f wall)  %.1f Mslot-steps/s  live-bytes %.2f TB/s" %
          (cs, n_chunks, ops, update, spawn, ms / steps, wall / steps, live * steps / ms / 1e3, live * 112 * steps / ms / 1e9))
    sysm.close(); eng.close()

def lighting(ctx, w=1920, h=1080, n_lights=64, res=0.25, frames=5, fmt=abi.SDF_UNORM16, world=2048):
    layout = scenes.DistanceFieldLayout(world, world, 128.0, 32, res, 128)
    t = time.time()
    atlas = scenes.build_sdf_atlas(layout, scenes.random_obstacles(11, 256, (world, world)), fmt=fmt)
    print("atlas %dx%d built in %.1fs" % (layout.atlas_width, layout.atlas_height, time.time() - t))
    dfu = layout.uniforms(max_cone_radius=24.0, power=0.7, step_limit=64, min_step_size=1.0, long_step_factor=0.5)
    sc = w / 1920.0
    lights = scenes.random_lights(12, n_lights, w, h, z=(8.0, 64.0), radius=24.0, ramp=(200.0 * sc, 550.0 * sc))
    env = scenes.environment()
    sdf = native.DistanceFieldTexture(ctx, atlas, fmt)
    lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_HALF4)
    st = native.render_sphere_lights(ctx, lights, env, dfu, None, sdf, (0.05, 0.05, 0.05, 1), lm, want_stats=True)
    ctx.sync()
    ctx.timer_start()
    for _ in range(frames):
        native.render_sphere_lights(ctx, lights, env, dfu, None, sdf, (0.05, 0.05, 0.05, 1), lm)
    ms = ctx.timer_stop() / frames
    print("lighting %dx%d %d lights: %.3f ms/frame  %.1f Mpx/s  samples %.3g (%.1f/px) pairs %.3g traced %.3g  algorithmic %.2f TB/s" %
          (w, h, n_lights, ms, w * h / ms / 1e3, st.SdfSamples, st.SdfSamples / (w * h), st.PixelLightPairs, st.TracedPairs,
           (st.SdfSamples * 32 + w * h * 8) / ms / 1e9))
    lm.close(); sdf.close()

def pcie(ctx):
    """Host <-> device legs of the boundary (AoS float4 staging + SoA conversion included), for DESIGN.md."""
    cs, n_chunks = 256, 16
    n = cs * cs
    rnd = scenes.randomness_table(7)
    eng = native.Engine(ctx, cs, rnd); sysm = native.System(eng)
    pos, vel, attr = scenes.make_particles(10, n * n_chunks)
    for c in range(n_chunks):
        sysm.add_chunk()
    ctx.sync()
    t0 = time.perf_counter()
    for c in range(n_chunks):
        sl = slice(c * n, (c + 1) * n)
        sysm.upload(c, abi.PLANE_POSITION, pos[sl]); sysm.upload(c, abi.PLANE_VELOCITY, vel[sl]); sysm.upload(c, abi.PLANE_ATTRIBUTES, attr[sl])
    up = time.perf_counter() - t0
    t0 = time.perf_counter()
    for c in range(n_chunks):
        for k in (abi.PLANE_POSITION, abi.PLANE_VELOCITY, abi.PLANE_RENDER_COLOR, abi.PLANE_RENDER_DATA):
            sysm.download(c, k)
    down = time.perf_counter() - t0
    print("pcie particles: upload 3 planes of %d slots %.2f ms (%.1f GB/s), download 4 planes %.2f ms (%.1f GB/s)" %
          (n * n_chunks, up * 1e3, n * n_chunks * 48 / up / 1e9, down * 1e3, n * n_chunks * 64 / down / 1e9))
    sysm.close(); eng.close()
    lm = native.Lightmap(ctx, 3840, 2160, abi.LIGHTMAP_HALF4)
    lm.download()
    t0 = time.perf_counter()
    lm.download()
    d = time.perf_counter() - t0
    print("pcie lightmap: 4K half4 download %.2f ms (%.1f GB/s)" % (d * 1e3, 3840 * 2160 * 8 / d / 1e9))
    lm.close()


if __name__ == "__main__":
    if os.environ.get("ILM_HIP_LIB"):       # experiment builds of the library (block-size variants ...)
        native.LIB_PATH = os.path.abspath(os.environ["ILM_HIP_LIB"])
    ctx = native.Context(0)
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "pcie":
        pcie(ctx)
    if what == "ablate":
        for ops, upd in (("gn", True), ("g", True), ("n", True), ("", True), ("gn", False), ("", False), ("f", True)):
            particles(ctx, 256, 16, 200, False, ops, upd)
    if what in ("all", "p"):
        particles(ctx, 256, 16, 100, True)
        particles(ctx, 256, 16, 100, False)
        particles(ctx, 1024, 8, 20, False)
    if what in ("all", "l"):
        lighting(ctx, 1920, 1080, 64, 0.25)
        lighting(ctx, 3840, 2160, 256, 0.125, frames=2, fmt=abi.SDF_FP16, world=4096)
    ctx.close()
