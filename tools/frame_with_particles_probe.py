"""Run ON THE GPU BOX.  cfg5's lit frame alone and with a particle step (2 chunks of 1024^2, cfg2's transforms) in front of every frame, both
on the device's clock and on the host's: where the combined frame's extra time sits.     python tools/frame_with_particles_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from illuminant_amd import abi, scenes  # noqa: E402
from illuminant_amd import _host as H  # noqa: E402

ctx = H.DeviceContext(0)
L = bench.build_lighting(H, ctx, scenes, abi, 3840, 2160, 256, 0.125, 4096, abi.SDF_FP16)
r = L["renderer"]
Q = bench.build_particle_system(H, ctx, scenes, abi, 1024, 2, 0, with_spawner=False)
ps, tp = Q["ps"], Q["tp"]
f = 0
for _ in range(3):
    tp.Advance(1 / 60); ps.Update(f); f += 1
    r.RenderLighting(1.0, 0, -1, False)
for what in ("lit frame alone", "particle step alone", "particle step + lit frame", "lit frame alone", "particle step + lit frame"):
    ctx.Sync()
    ctx.TimerStart()
    t0 = time.perf_counter()
    n = 16
    for _ in range(n):
        if "step" in what:
            tp.Advance(1 / 60); ps.Update(f); f += 1
        if "lit" in what:
            r.RenderLighting(1.0, 0, -1, False)
    t1 = time.perf_counter()
    ms = ctx.TimerStop() / n
    t2 = time.perf_counter()
    print("%-28s %.4f ms per frame on the device's clock, %.4f on the host's (enqueue %.1f us per frame)" % (what, ms, (t2 - t0) / n * 1e3, (t1 - t0) / n * 1e6))
