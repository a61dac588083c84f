"""Run ON THE GPU BOX.  ilm_gbuffer_render (the non-2.5D G-buffer: ground plane + height-volume tops decided per pixel against the polygons) at
1080p with 0 / 16 / 256 volumes.     python tools/gbuffer_polygon_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from illuminant_amd import abi, native, scenes  # noqa: E402

ctx = native.Context(0)
w, h = 1920, 1080
for n in (0, 16, 256):
    r = scenes.uniform(77, (max(n, 1), 8))
    volumes = []
    for v in range(n):
        cx, cy, rad = 40 + r[v, 0] * 1840, 80 + r[v, 1] * 960, 12 + r[v, 2] * 50
        nv = 4 + int(r[v, 3] * 4)
        ang = np.sort(scenes.uniform(770 + v, (nv,)) * 2 * np.pi)
        volumes.append(([(float(cx + rad * np.cos(a)), float(cy + rad * np.sin(a))) for a in ang], float(r[v, 4] * 8), float(6 + r[v, 5] * 70), True, True))
    vols, poly = scenes.height_volume_arrays(volumes)
    gb = native.GBufferTexture(ctx, None, abi.GBUFFER_FLOAT4, size=(w, h))
    desc = scenes.gbuffer_render_desc(0.0)
    for _ in range(3):
        gb.render(desc, vols, poly)
    ctx.sync()
    ctx.timer_start()
    t0 = time.perf_counter()
    frames = 100
    for _ in range(frames):
        gb.render(desc, vols, poly)
    host = (time.perf_counter() - t0) / frames
    ms = ctx.timer_stop() / frames
    print("%3d volumes: %.4f ms per frame on the device's clock, host %.1f us per call" % (n, ms, host * 1e6))
    gb.close()
