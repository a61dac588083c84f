"""Run ON THE GPU BOX.  What a workgroup of the light pass costs before and after its lights: whole frames of cfg3 / cfg5 size with
(a) no lights at all (prologue + store), (b) the config's number of lights, all far outside the frame (prologue + binning of every
light + store: every tile list is empty), under each light-split setting.    python tools/light_overhead_probe.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from illuminant_amd import abi, native, scenes  # noqa: E402
from tools.strip_probe import build, timed  # noqa: E402

ctx = native.Context(0)
env = scenes.environment()
ambient = (0.05, 0.05, 0.05, 1.0)
for name in ("cfg3", "cfg5"):
    w, h, dfu, lights, sdf = build(ctx, name)
    lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_HALF4)
    n = len(lights)
    far = (abi.LightVertex * n)(*[scenes.sphere_light((-50000.0 - 10.0 * i, -50000.0, 16.0), 24.0, 300.0) for i in range(n)])
    none = (abi.LightVertex * 0)()
    for split in (1, 2, 4, 8):
        ctx.set_light_split(split)
        t_none = timed(ctx, lambda: native.render_sphere_lights(ctx, none, env, dfu, None, sdf, ambient, lm), 20)
        t_far = timed(ctx, lambda: native.render_sphere_lights(ctx, far, env, dfu, None, sdf, ambient, lm), 20)
        t_all = timed(ctx, lambda: native.render_sphere_lights(ctx, lights, env, dfu, None, sdf, ambient, lm), 8)
        print("%s split=%d  no lights %.4f ms   %d lights outside the frame %.4f ms   the frame %.4f ms" % (name, split, t_none, n, t_far, t_all), flush=True)
    lm.close(); sdf.close()
ctx.close()
