"""Experiment: how are the workgroups of the rasteriser's tile pass spread over time?  Needs the variant library built with -DILM_RASTER_TRACE
(tools/ab_build.sh rtrace raster.hip -DILM_RASTER_TRACE): every work item records when it started and ended (100 MHz clock), how many
sprites its segment held and which segment of its tile it was.  The scene is bench.py's cfg2 system after `steps` steps.
   LD_LIBRARY_PATH=tools/ab/rtrace ILM_HIP_LIB=tools/ab/rtrace/libilluminant_hip.so python tools/raster_trace_probe.py [steps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from illuminant_amd import abi, native, scenes
from illuminant_amd import _host as H
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1800
h = C.CDLL(native.LIB_PATH)
ctx = H.DeviceContext(0)
P = bench.build_particle_system(H, ctx, scenes, abi, 256, 16, 0)
ps, tp = P["ps"], P["tp"]
for f in range(steps):
    tp.Advance(1.0 / 60.0); ps.Update(f)
target = H.RenderTarget(ctx, 1920, 1080, abi.LIGHTMAP_RGBA8)
target.Clear([0.0, 0.0, 0.0, 1.0])
stats = ps.Render(target, abi.BLEND_ALPHA, [0.0, 0.0], [1.0, 1.0], [1.0, 1.0], [0.0, 0.0], True)
for _ in range(3):
    ps.Render(target, abi.BLEND_ALPHA, [0.0, 0.0], [1.0, 1.0], [1.0, 1.0], [0.0, 0.0], False)
ctx.Sync()
buf = np.zeros(4 * 131072, np.uint64)
assert h.ilm_experiment_raster_trace(buf.ctypes.data_as(C.c_void_p), C.c_int(len(buf))) == 0
t = buf.reshape(-1, 4)
t = t[t[:, 0] > 0]
t0 = int(t[:, 0].min())
start, end = (t[:, 0].astype(np.int64) - t0) * 0.01, (t[:, 1].astype(np.int64) - t0) * 0.01
life, sprites = end - start, t[:, 2].astype(np.int64)
span = end.max()
print("live quads %d, pairs %d; %d work items; span %.1f us; lifetime us: median %.1f p90 %.1f max %.1f; average workgroups in flight %.0f (of 256 CUs x 9)"
      % (int(stats[0]), int(stats[1]), len(t), span, np.median(life), np.percentile(life, 90), life.max(), life.sum() / span))
edges = np.linspace(0.0, span, 21)
print("   workgroups in flight at 0 %, 5 %, ...:", [int(((start <= x) & (end > x)).sum()) for x in edges])
for lo, hi in ((0, 1), (1, 64), (64, 256), (256, 512), (512, 1024), (1024, 2048), (2048, 2049)):
    m = (sprites >= lo) & (sprites < hi)
    if m.any():
        print("   segments of %4d..%4d sprites: %6d items, median lifetime %.1f us, %.3f us per sprite" % (lo, hi - 1, int(m.sum()), np.median(life[m]), life[m].sum() / max(sprites[m].sum(), 1)))
