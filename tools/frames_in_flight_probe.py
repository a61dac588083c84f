"""Run ON THE GPU BOX.  Frames in flight: the reference keeps a ring of lightmaps (BufferRing, LightingRenderer.cs:472-485) because frame
N + 1 is built while frame N renders.  Here: the same lit frame rendered back to back (a) on ONE context / stream into one lightmap -- what
bench.py times: every launch waits for the previous one's last wave -- and (b) alternately on TWO sibling contexts of the same device (ilm_ctx_create_sibling: own
stream, own lightmap; ONE field and G-buffer, owned by the first and read by both), so that the next frame's first waves fill the slots the previous frame's drain leaves empty.
Wall clock over N frames, both contexts drained at the end; whole frames and one rank's strip of an 8-rank frame.
    python tools/frames_in_flight_probe.py [frames]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from illuminant_amd import abi, native, scenes, sharding  # noqa: E402
from tools.strip_probe import build  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ambient = (0.05, 0.05, 0.05, 1.0)
first = native.Context(0)
ctxs = [first, first.sibling(), first.sibling(), first.sibling()]     # ilm_ctx_create_sibling: the others READ the first one's field and G-buffer
for name in ("cfg3", "cfg5"):
    one = build(first, name)
    S = [one, one]
    w, h, dfu, lights = one[:4]
    env = scenes.environment(gbuffer_size=(w, h))
    gb = native.GBufferTexture(first, scenes.ground_plane_gbuffer(w, h, abi.GBUFFER_FLOAT4), abi.GBUFFER_FLOAT4)
    gbs = [gb] * len(ctxs)
    S = [one] * len(ctxs)
    lms = [native.Lightmap(c, w, h, abi.LIGHTMAP_HALF4) for c in ctxs]
    strips = sharding.balanced_row_strips(h, 8, lights)
    n = frames if name == "cfg3" else max(20, frames // 8)
    for label, (b, e) in (("whole frame", (0, h)), ("strip 4 of 8", strips[4]), ("strip 0 of 8", strips[0])):
        def render(k):
            native.render_sphere_lights(ctxs[k], lights, env, dfu, gbs[k], S[k][4], ambient, lms[k], b, e)
        out = {}
        for in_flight in (1, 2, 3, 4):
            for k in range(len(ctxs)):
                render(k); render(k)
            for c in ctxs:
                c.sync()
            t0 = time.perf_counter()
            for i in range(n):
                render(i % in_flight)
            for c in ctxs:
                c.sync()
            out[in_flight] = (time.perf_counter() - t0) / n * 1e3
        same = all(np.array_equal(lms[0].download(b, e - b), lm.download(b, e - b)) for lm in lms[1:])
        print("%s %-13s rows [%4d, %4d): ms per frame with 1 / 2 / 3 / 4 frames in flight: %.4f  %.4f (%+.1f %%)  %.4f (%+.1f %%)  %.4f (%+.1f %%); the contexts' frames are %s"
              % (name, label, b, e, out[1], out[2], 100.0 * (out[2] / out[1] - 1.0), out[3], 100.0 * (out[3] / out[1] - 1.0), out[4], 100.0 * (out[4] / out[1] - 1.0),
                 "bit-equal" if same else "DIFFERENT"), flush=True)
    for x in lms + [gb, one[4]]:
        x.close()
