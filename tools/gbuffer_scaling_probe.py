"""Run ON THE GPU BOX.  The mesh G-buffer frame of bench.py with its volumes replicated 1 / 10 / 40 times (2.5 k, 24 k, 95 k triangles):
does the block-list scheme stay linear?     python tools/gbuffer_scaling_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from illuminant_amd import abi, native  # noqa: E402

ctx = native.Context(0)
gd, top, front, bbv = bench.gbuffer_meshes_scene()
runs = [(None, 0, 64, abi.BILLBOARD_MASK)]
for copies in (1, 10, 40):
    tops, fronts = [], []
    for k in range(copies):
        dx, dy = (k * 37) % 200 - 100, (k * 91) % 160 - 80
        t = top.copy(); t[:, 0] += dx; t[:, 1] += dy; tops.append(t)
        f = front.copy(); f[:, 0] += dx; f[:, 1] += dy; fronts.append(f)
    T, F = np.ascontiguousarray(np.concatenate(tops)), np.ascontiguousarray(np.concatenate(fronts))
    gbt = native.GBufferTexture(ctx, None, abi.GBUFFER_FLOAT4, size=(1920, 1080))
    for _ in range(3):
        gbt.render_meshes(gd, T, F, bbv, runs)
    ctx.sync()
    ctx.timer_start()
    n = 50
    for _ in range(n):
        gbt.render_meshes(gd, T, F, bbv, runs)
    print("%6d triangles (%5.1f MB of vertices): %.4f ms per frame" % (2 + len(T) // 3 + len(F) // 3 + 128, (T.nbytes + F.nbytes + bbv.nbytes) / 1e6, ctx.timer_stop() / n))
    gbt.close()
