"""Run ON THE GPU BOX with the -DILM_LIGHT_TRACE variant (tools/ab_build.sh ltrace lighting.hip -DILM_LIGHT_TRACE): where does one rank's
strip of an 8-rank cfg5 / cfg3 frame lose its time?  Every wave of the strip's launch records start / end (100 MHz clock), tile and XCC.
Prints, per light-split setting: the launch span against the share (whole frame / 8), how full the chip is over the span, and per XCC
its last wave's end and its busy wave-time -- imbalance between the XCDs shows as XCCs that end early, a drain as occupancy falling.
    ILM_HIP_LIB=tools/ab/ltrace/libilluminant_hip.so python tools/strip_trace_probe.py [cfg5|cfg3] [strip index] [ranks]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from illuminant_amd import abi, native, scenes, sharding  # noqa: E402
from tools.strip_probe import build  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
which = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ranks = int(sys.argv[3]) if len(sys.argv) > 3 else 8
h = C.CDLL(native.LIB_PATH)
ctx = native.Context(0)
w, hh, dfu, lights, sdf = build(ctx, name)
env = scenes.environment(gbuffer_size=(w, hh))
gb = native.GBufferTexture(ctx, scenes.ground_plane_gbuffer(w, hh, abi.GBUFFER_FLOAT4), abi.GBUFFER_FLOAT4)
lm = native.Lightmap(ctx, w, hh, abi.LIGHTMAP_HALF4)
b, e = sharding.balanced_row_strips(hh, ranks, lights)[which]
print("%s strip %d of %d: rows [%d, %d)" % (name, which, ranks, b, e))
for split in (1, 0):
    ctx.set_light_split(split)
    for _ in range(3):
        native.render_sphere_lights(ctx, lights, env, dfu, gb, sdf, (0.05, 0.05, 0.05, 1.0), lm, b, e)
    ctx.sync()
    wg, sp, macro = ctx.last_light_launch()
    n_waves = wg * 4
    buf = np.zeros(4 * 262144, np.uint64)
    assert h.ilm_experiment_light_trace(buf.ctypes.data_as(C.c_void_p), C.c_int(4 * 262144)) == 0
    t = buf.reshape(-1, 4)[:n_waves]
    t = t[t[:, 0] > 0]
    t0 = int(t[:, 0].min())
    start, end = (t[:, 0].astype(np.int64) - t0) * 0.01, (t[:, 1].astype(np.int64) - t0) * 0.01      # us
    life = end - start
    span = float(end.max())
    print("split %s: %d workgroups (largest split %d, tile groups of %d): %d waves recorded; span %.1f us; wave life us median %.1f p90 %.1f max %.1f; "
          "mean occupancy %.0f of 8192 wave slots" % (split if split else "auto", wg, sp, macro, len(t), span, np.median(life), np.percentile(life, 90), life.max(), life.sum() / span))
    edges = np.linspace(0.0, span, 21)
    print("   waves in flight at 0 %, 5 %, ... 100 % of the span:", [int(((start <= x) & (end > x)).sum()) for x in edges])
    xcc = (t[:, 3] >> np.uint64(32)).astype(np.int64) & 0xF
    ends = []
    for x in range(8):
        m = xcc == x
        if m.any():
            ends.append(end[m].max())
            print("   XCC %d: %5d waves, last end %7.1f us, busy wave-us %9.0f, mean waves in flight until its end %.0f" % (x, int(m.sum()), end[m].max(), life[m].sum(), life[m].sum() / end[m].max()))
    print("   XCC ends: first %.1f us, last %.1f us: the chip waits %.1f %% of the span for its slowest XCD; total busy wave-us %.0f = %.1f us at 8192 slots" % (
        min(ends), max(ends), 100.0 * (max(ends) - np.mean(ends)) / span, life.sum(), life.sum() / 8192.0))
for x in (lm, gb, sdf):
    x.close()
