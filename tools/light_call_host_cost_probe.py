"""Run ON THE GPU BOX: what ONE ilm_render_sphere_lights call costs the HOST (no synchronisation inside the timed loop; the device is
kept far behind by making the frames tiny): the ctypes binding's share is measured by timing a trivially cheap ABI call the same way.
    python tools/light_call_host_cost_probe.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from illuminant_amd import abi, native, scenes  # noqa: E402
from tools.strip_probe import build  # noqa: E402

ctx = native.Context(0)
for name in ("cfg3", "cfg5"):
    w, h, dfu, lights, sdf = build(ctx, name)
    env = scenes.environment()
    lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_HALF4)
    n = len(lights)
    # lights that touch no pixel: the device side of a call is three near-empty launches, so the loop below runs at the HOST's pace
    lights = (abi.LightVertex * n)(*[scenes.sphere_light((-50000.0 - 10.0 * i, -50000.0, 16.0), 24.0, 300.0) for i in range(n)])
    amb = (C.c_float * 4)(0.05, 0.05, 0.05, 1.0)
    lib = native.lib()
    args = (ctx.handle, C.cast(lights, C.c_void_p), n, C.byref(env), C.byref(dfu), abi.Handle(0), sdf.handle, C.cast(amb, C.c_void_p), lm.handle)
    for rows in (16, 128):
        for _ in range(20):
            lib.ilm_render_sphere_lights(*args, 0, rows, None)
        ctx.sync()
        reps = 400
        t0 = time.perf_counter()
        for _ in range(reps):
            lib.ilm_render_sphere_lights(*args, 0, rows, None)
        host = (time.perf_counter() - t0) / reps * 1e6
        ctx.sync()
        total = (time.perf_counter() - t0) / reps * 1e6
        print("%s, %d lights, rows [0, %d): %.1f us of host time per call (raw ctypes call, arguments prebuilt), %.1f us per call with the device drained at the end" % (name, n, rows, host, total))
    t0 = time.perf_counter()
    for _ in range(2000):
        lib.ilm_device_count()
    print("   (a trivial ABI call through ctypes: %.2f us)" % ((time.perf_counter() - t0) / 2000 * 1e6))
    lm.close(); sdf.close()
