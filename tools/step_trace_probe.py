"""Experiment: where does a step launch spend its time?  Needs the variant library built with -DILM_STEP_TRACE (tools/ab_build.sh trace
particles.hip -DILM_STEP_TRACE); every wave of the lean step kernel records when it started, when its 12 loads had arrived and when
it ended (100 MHz clock).  Prints the launch's occupancy over time.   ILM_HIP_LIB=tools/ab/trace/libilluminant_hip.so python tools/step_trace_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("tsl", os.path.join(os.path.dirname(os.path.abspath(__file__)), "two_stream_lib.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
from illuminant_amd import native
native.lib().ilm_debug_step_streams(1)
h = C.CDLL(native.LIB_PATH)
a = native.Context(0)
for (cs, chunks, spawn) in ((256, 16, False), (256, 16, True), (256, 1, False)):
    e, s, d = m.make(a, cs, chunks, 10, spawn)
    for _ in range(30): s.step(d)
    a.sync()
    n_waves = (chunks + (1 if spawn else 0)) * (cs * cs // 64)
    buf = np.zeros(5 * n_waves, np.uint64)
    assert h.ilm_experiment_step_trace(buf.ctypes.data_as(C.c_void_p), C.c_int(5 * n_waves)) == 0
    t = buf.reshape(-1, 5).astype(np.int64)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    start, loaded, end = (t[:, 0] - t0) * 0.01, (t[:, 1] - t0) * 0.01, (t[:, 2] - t0) * 0.01       # us
    print("cs=%d chunks=%d spawn=%d: %d waves; launch span %.2f us (first start -> last end)" % (cs, chunks, spawn, len(t), end.max()))
    life = end - start
    print("   wave lifetime us: median %.2f  p10 %.2f  p90 %.2f  max %.2f ; loads arrive after median %.2f us (p90 %.2f)" %
          (np.median(life), np.percentile(life, 10), np.percentile(life, 90), life.max(), np.median(loaded - start), np.percentile(loaded - start, 90)))
    ok = t[:, 3] > t[:, 0]
    print("   phases (median us): entry -> loads %.2f | transforms %.2f | update + render data %.2f | stores + exit %.2f" %
          (np.median(loaded - start), np.median((t[ok, 3] - t[ok, 1]) * 0.01), np.median((t[ok, 4] - t[ok, 3]) * 0.01), np.median((t[ok, 2] - t[ok, 4]) * 0.01)))
    print("   starts: p1 %.2f p25 %.2f p50 %.2f p75 %.2f p99 %.2f last %.2f" % tuple(np.percentile(start, [1, 25, 50, 75, 99, 100])))
    print("   ends:   p1 %.2f p25 %.2f p50 %.2f p75 %.2f p99 %.2f last %.2f" % tuple(np.percentile(end, [1, 25, 50, 75, 99, 100])))
    edges = np.arange(0.0, end.max() + 1.0, 1.0)
    running = [int(((start <= x) & (end > x)).sum()) for x in edges]
    print("   waves in flight at t = 0, 1, 2 ... us:", running)
    life_by_start = [round(float(np.median(life[(start >= x) & (start < x + 2.0)])), 2) if ((start >= x) & (start < x + 2.0)).any() else None for x in edges[::2]]
    print("   median lifetime of waves starting in [t, t+2):", life_by_start)
    s.close(); e.close()
