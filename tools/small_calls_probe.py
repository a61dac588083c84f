"""Run ON THE GPU BOX.  The small synchronous entry points a host may call every frame, end to end on the host's clock: live counts, the
ordered live-slot list of a chunk, SDF point queries, a lightmap clear.     python tools/small_calls_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from illuminant_amd import abi, native, scenes  # noqa: E402
from tools.strip_probe import build  # noqa: E402

ctx = native.Context(0)
cs = 256
eng = native.Engine(ctx, cs, scenes.randomness_table(7))
sysm = native.System(eng)
for c in range(16):
    sysm.add_chunk()
    pos, vel, attr = scenes.make_particles(100 + c, cs * cs, pos_lo=(0, 0, 0), pos_hi=(1920, 1080, 32), life=(0.5, 60.0), dead_fraction=0.3)
    for plane, data in ((abi.PLANE_POSITION, pos), (abi.PLANE_VELOCITY, vel), (abi.PLANE_ATTRIBUTES, attr)):
        sysm.upload(c, plane, data)
w, h, dfu, lights, sdf = build(ctx, "cfg3")
lm = native.Lightmap(ctx, 3840, 2160, abi.LIGHTMAP_HALF4)
pts = np.stack([scenes.uniform(1, (1000,), 0, 2048), scenes.uniform(2, (1000,), 0, 2048), scenes.uniform(3, (1000,), 0, 100)], axis=1).astype(np.float32)


def timed(name, fn, reps=100):
    for _ in range(3):
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    print("%-52s %8.1f us per call" % (name, (time.perf_counter() - t0) / reps * 1e6))


timed("ilm_system_live_counts (16 chunks of 256^2)", lambda: sysm.live_counts())
timed("ilm_chunk_live_slots (one chunk, ~46 000 live)", lambda: sysm.live_slots(3))
timed("ilm_sdf_sample (1 000 positions)", lambda: sdf.sample(dfu, pts))
timed("ilm_lightmap_clear (4K half4)", lambda: lm.clear((0.1, 0.2, 0.3, 1.0)))
