"""Experiment: how are the waves of one light-pass launch spread over time?  Needs the variant library built with -DILM_LIGHT_TRACE
(tools/ab_build.sh ltrace lighting.hip -DILM_LIGHT_TRACE): every wave records when it started and ended (100 MHz clock), its tile and
where it ran.   ILM_HIP_LIB=tools/ab/ltrace/libilluminant_hip.so python tools/light_trace_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from illuminant_amd import abi, native, scenes
h = C.CDLL(native.LIB_PATH)
ctx = native.Context(0)
for name, (w, hh, n_lights, res, virt, fmt) in (("cfg3", (1920, 1080, 64, 0.25, 2048, abi.SDF_UNORM16)), ("cfg5", (3840, 2160, 256, 0.125, 4096, abi.SDF_FP16))):
    layout = scenes.DistanceFieldLayout(virt, virt, 128.0, 32, res, 128)
    obs = scenes.obstruction_array(scenes.random_obstructions(11, 256, (virt, virt)))
    field = native.DistanceFieldTexture(ctx, None, fmt, size=(layout.atlas_width, layout.atlas_height))
    field.render_slices(scenes.render_desc(layout), list(range(0, layout.slice_count, 3)), obs)
    dfu = layout.uniforms(max_cone_radius=24.0, power=0.7, step_limit=64, min_step_size=1.0, long_step_factor=0.5)
    scale = w / 1920.0
    lights = scenes.random_lights(5, n_lights, w, hh, z=(8.0, 64.0), radius=24.0, ramp=(200.0 * scale, 550.0 * scale))
    env = scenes.environment()
    lm = native.Lightmap(ctx, w, hh, abi.LIGHTMAP_HALF4)
    for _ in range(3):
        native.render_sphere_lights(ctx, lights, env, dfu, None, field, (0.05, 0.05, 0.05, 1.0), lm)
    ctx.sync()
    n_waves = ((w + 15) // 16) * ((hh + 15) // 16) * 4
    buf = np.zeros(4 * n_waves, np.uint64)
    assert h.ilm_experiment_light_trace(buf.ctypes.data_as(C.c_void_p), C.c_int(4 * n_waves)) == 0
    t = buf.reshape(-1, 4)
    t = t[t[:, 0] > 0]
    t0 = int(t[:, 0].min())
    start, end = (t[:, 0].astype(np.int64) - t0) * 0.01, (t[:, 1].astype(np.int64) - t0) * 0.01      # us
    life = end - start
    span = end.max()
    print("%s: %d waves; launch span %.1f us; wave lifetime us: median %.1f p10 %.1f p90 %.1f max %.1f; sum of lifetimes / (span x 1024 SIMDs) = %.2f waves per SIMD on average"
          % (name, len(t), span, np.median(life), np.percentile(life, 10), np.percentile(life, 90), life.max(), life.sum() / (span * 1024)))
    edges = np.linspace(0.0, span, 21)
    print("   waves in flight at 0 %, 5 %, ... of the span:", [int(((start <= x) & (end > x)).sum()) for x in edges])
    xcc = (t[:, 3] >> np.uint64(32)).astype(np.int64) & 0xF
    for x in range(8):
        m = xcc == x
        if m.any():
            print("   XCC %d: %d waves, last end %.1f us, busy wave-us %.0f" % (x, int(m.sum()), end[m].max(), life[m].sum()))
    lm.close(); field.close()
