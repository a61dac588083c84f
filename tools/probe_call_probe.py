"""Run ON THE GPU BOX.  What one ilm_render_light_probes call costs end to end (it returns the values: the call is synchronous): 256 probes
under cfg3's 64 lights and field.     python tools/probe_call_probe.py [probes]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from illuminant_amd import native, scenes  # noqa: E402
from tools.strip_probe import build  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = native.Context(0)
w, h, dfu, lights, sdf = build(ctx, "cfg3")
pp = np.zeros((n, 4), np.float32)
pp[:, 3] = 1.0                                      # opacity: a used probe slot
pp[:, 0] = scenes.uniform(5, (n,), 0.0, w); pp[:, 1] = scenes.uniform(6, (n,), 0.0, h); pp[:, 2] = scenes.uniform(7, (n,), 0.0, 32.0)
pn = np.zeros((n, 4), np.float32); pn[:, 2] = 1.0; pn[:, 3] = 1.0      # up, shadows enabled
env = scenes.environment()
for _ in range(5):
    v = native.render_light_probes(ctx, lights, pp, pn, env, dfu, sdf)
t0 = time.perf_counter()
reps = 200
for _ in range(reps):
    v = native.render_light_probes(ctx, lights, pp, pn, env, dfu, sdf)
print("%d probes x %d lights: %.1f us per call (values checksum %.6g)" % (n, len(lights), (time.perf_counter() - t0) / reps * 1e6, float(v.sum())))
