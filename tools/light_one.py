"""Run ON THE GPU BOX (under rocprofv3 by tools/pmc_dispatches.sh): a few launches of one light-pass scenario.
    python tools/light_one.py <cfg3|cfg5> <all|far|none> <split> [frames] [row_begin row_end]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from illuminant_amd import abi, native, scenes  # noqa: E402
from tools.strip_probe import build  # noqa: E402

name, kind, split = sys.argv[1], sys.argv[2], int(sys.argv[3])
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ctx = native.Context(0)
w, h, dfu, lights, sdf = build(ctx, name)
rb, re_ = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (0, h)
n = len(lights)
if kind == "far":
    lights = (abi.LightVertex * n)(*[scenes.sphere_light((-50000.0 - 10.0 * i, -50000.0, 16.0), 24.0, 300.0) for i in range(n)])
elif kind == "none":
    lights = (abi.LightVertex * 0)()
lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_HALF4)
ctx.set_light_split(split)
for _ in range(frames):
    native.render_sphere_lights(ctx, lights, scenes.environment(), dfu, None, sdf, (0.05, 0.05, 0.05, 1.0), lm, rb, re_)
ctx.sync()
