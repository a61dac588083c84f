"""Experiment: per-workgroup timeline of one distance-field generation launch (a -DILM_FIELD_TRACE build of the library):
    tools/ab_build.sh ftrace fields.hip -DILM_FIELD_TRACE;  ILM_HIP_LIB=tools/ab/ftrace/libilluminant_hip.so LD_LIBRARY_PATH=tools/ab/ftrace python tools/field_trace_probe.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from illuminant_amd import abi, native, scenes
from illuminant_amd import _host as H

ctx = H.DeviceContext(0)
for name, (w, h, nl, res, wsize, fmt) in (("cfg3", (1920, 1080, 64, 0.25, 2048, abi.SDF_UNORM16)), ("cfg5", (3840, 2160, 256, 0.125, 4096, abi.SDF_FP16))):
    L = bench.build_lighting(H, ctx, scenes, abi, w, h, nl, res, wsize, fmt)
    ctx.Sync()
    f = L["field"]
    blocks = f.PhysicalSliceCount * ((f.SliceWidth + 31) // 32) * ((f.SliceHeight + 7) // 8)
    n = min(blocks, 65536)
    buf = np.zeros(4 * n, np.uint64)
    rc = native.lib().ilm_experiment_field_trace(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), ctypes.c_int(4 * n))
    assert rc == 0
    t = buf.reshape(n, 4).astype(np.int64)
    t0, t1, listed, evaluated = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    life = (t1 - t0) / 100.0                      # us (100 MHz)
    span = (t1.max() - t0.min()) / 100.0
    order = np.argsort(life)
    print("%s: %d workgroups, launch span %.1f us; workgroup life mean %.2f us, median %.2f, p90 %.2f, p99 %.2f, max %.2f; sum of lives / span = %.0f workgroups in flight"
          % (name, n, span, life.mean(), np.median(life), np.percentile(life, 90), np.percentile(life, 99), life.max(), life.sum() / span))
    print("   list length mean %.1f max %d; evaluated (passed the cull) mean %.1f max %d" % (listed.mean(), listed.max(), evaluated.mean(), evaluated.max()))
    late = np.argsort(t1)[-5:]
    for b in late:
        print("   one of the last to finish: block %d started %.1f us into the launch, lived %.1f us, list %d, evaluated %d" % (b, (t0[b] - t0.min()) / 100.0, life[b], listed[b], evaluated[b]))
    started = (t0 - t0.min()) / 100.0
    print("   last workgroup started %.1f us into the launch" % started.max())
    for edge in (1, 2, 5, 10, 20, 50):
        print("   started within the first %2d us: %5d   (in flight at that time: %d)" % (edge, int((started <= edge).sum()), int(((started <= edge) & ((t1 - t0.min()) / 100.0 > edge)).sum())))
