"""Run ON THE GPU BOX.  The reference's default cadence for a dynamic field: MaximumFieldUpdatesPerFrame = 1 slice triplet per UpdateFields
(LightingRenderer.Configuration.cs:91), every frame, in front of the lit frame -- what one such update costs on the device (slices + the
cells of the slices around them) and on the host, for cfg3's and cfg5's fields.     python tools/field_update_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from illuminant_amd import abi, scenes  # noqa: E402
from illuminant_amd import _host as H  # noqa: E402

ctx = H.DeviceContext(0)
for name, (w, h, nl, res, world, fmt) in {"cfg3": (1920, 1080, 64, 0.25, 2048, abi.SDF_UNORM16), "cfg5": (3840, 2160, 256, 0.125, 4096, abi.SDF_FP16)}.items():
    L = bench.build_lighting(H, ctx, scenes, abi, w, h, nl, res, world, fmt)
    r = L["renderer"]
    r.Configuration.MaximumFieldUpdatesPerFrame = 1
    r.Configuration.EnableGBuffer = False
    frames = 44                                     # four passes over the field's eleven triplets
    for update, lit in ((True, False), (False, True), (True, True)):
        r.Configuration.MaximumFieldUpdatesPerFrame = 9999
        r.InvalidateFields(); r.UpdateFields()
        r.Configuration.MaximumFieldUpdatesPerFrame = 1
        if lit:
            r.RenderLighting(1.0, 0, -1, False)
        ctx.Sync()
        ctx.TimerStart()
        t0 = time.perf_counter()
        for k in range(frames):
            if update:
                if k % 11 == 0:
                    r.InvalidateFields()
                r.UpdateFields()
            if lit:
                r.RenderLighting(1.0, 0, -1, False)
        host = (time.perf_counter() - t0) / frames
        ms = ctx.TimerStop() / frames
        print("%s: %s%s: %.4f ms per frame on the device's clock, host %.1f us per frame" % (
            name, "one slice triplet per frame" if update else "static field", " + the lit frame" if lit else "", ms, host * 1e6))
    del L, r
