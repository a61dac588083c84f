"""Run ON THE GPU BOX.  The particle-light frame of bench.py (1080p, cfg3's field, Radius 4 / RampLength 60) against the number of live
particles that light it: 4 096 (the bench row), 16 384, 65 536.     python tools/particle_lights_scaling_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from illuminant_amd import abi, scenes  # noqa: E402
from illuminant_amd import _host as H  # noqa: E402

ctx = H.DeviceContext(0)
L = bench.build_lighting(H, ctx, scenes, abi, 1920, 1080, 0, 0.25, 2048, abi.SDF_UNORM16)
r = L["renderer"]
for n, cs in ((4096, 64), (16384, 128), (65536, 256)):
    ecfg = H.ParticleEngineConfiguration(cs)
    eng = H.ParticleEngine(ctx, ecfg, scenes.randomness_table(7))
    pcfg = H.ParticleSystemConfiguration()
    pcfg.LifeDecayPerSecond = 0.01
    lsys = H.ParticleSystem(eng, pcfg)
    pos, vel, attr = scenes.make_particles(91, n, pos_lo=(0, 0, 4), pos_hi=(1920, 1080, 48), life=(50.0, 90.0))
    lsys.Spawn(n, pos, vel, attr)
    lsys.Update(0)
    pls = H.ParticleLightSource()
    tmpl = H.SphereLightSource()
    tmpl.Radius = 4.0; tmpl.RampLength = 60.0; tmpl.Color = [1.0, 0.9, 0.8, 1.0]
    pls.Template = tmpl
    pls.System = lsys
    L["env"].ParticleLights = [pls]
    stats = r.RenderLighting(1.0, 0, -1, True)
    ctx.Sync()
    ctx.TimerStart()
    frames = 3
    for _ in range(frames):
        r.RenderLighting(1.0, 0, -1, False)
    ms = ctx.TimerStop() / frames
    print("%6d particle lights: %.3f ms per frame, %.1f M pixel-light pairs, %.1f M SDF samples -> %.2f ns per pair" % (
        n, ms, stats[1] / 1e6, stats[0] / 1e6, ms * 1e6 / max(stats[1], 1)))
    L["env"].ParticleLights = []
    del pls, lsys, eng
