"""Run ON THE GPU BOX.  Field generation from height volumes (DistanceToPolygon) instead of analytic obstructions: cfg3's atlas (33 slices of
512 x 512) with 0 / 16 / 256 volumes.     python tools/field_volume_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from illuminant_amd import abi, native, scenes  # noqa: E402

ctx = native.Context(0)
layout = scenes.DistanceFieldLayout(2048, 2048, 128.0, 32, 0.25)
sdf = native.DistanceFieldTexture(ctx, None, abi.SDF_UNORM16, size=(layout.atlas_width, layout.atlas_height))
slices = list(range(0, layout.slice_count, 3))
for n in (0, 16, 256):
    r = scenes.uniform(77, (max(n, 1), 8))
    volumes = []
    for v in range(n):
        cx, cy, rad = 60 + r[v, 0] * 1900, 60 + r[v, 1] * 1900, 12 + r[v, 2] * 50
        nv = 4 + int(r[v, 3] * 4)
        ang = np.sort(scenes.uniform(770 + v, (nv,)) * 2 * np.pi)
        volumes.append(([(float(cx + rad * np.cos(a)), float(cy + rad * np.sin(a))) for a in ang], float(r[v, 4] * 8), float(6 + r[v, 5] * 70), True, True))
    vols, poly = scenes.height_volume_arrays(volumes)
    fv = vols
    d = scenes.render_desc(layout)
    for _ in range(2):
        sdf.render_slices(d, slices, None, fv if n else None, poly if n else None)
    ctx.sync()
    ctx.timer_start()
    for _ in range(10):
        sdf.render_slices(d, slices, None, fv if n else None, poly if n else None)
    print("%3d volumes: %.4f ms per whole field" % (n, ctx.timer_stop() / 10))
