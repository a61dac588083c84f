"""Run ON THE GPU BOX.  What the contract's timed block (barrier, K steps, barrier: bench.py) costs beyond its K steps: the host clock around
the block against the HIP events around the same K steps, for several K, and where the difference sits (first launch after an idle device,
the wait for the stop event, the closing synchronisation).      python tools/block_overhead_probe.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bench  # noqa: E402
from illuminant_amd import abi, scenes  # noqa: E402
from illuminant_amd import _host as H  # noqa: E402

ctx = H.DeviceContext(0)
P = bench.build_particle_system(H, ctx, scenes, abi, 256, 16, 0, with_spawner=("--spawner" in sys.argv))      # without: a constant 1 048 576 particles
ps, tp = P["ps"], P["tp"]
f = 0
for _ in range(25):
    tp.Advance(1 / 60); ps.Update(f); f += 1
ctx.Sync()
for K in (1, 5, 20, 80, 200, 800):
    rows = []
    for _ in range(9 if K > 100 else 15):
        ctx.Sync()
        if "--idle" in sys.argv:
            time.sleep(0.002)                      # a device that has been idle for 2 ms in front of every block
        ctx.TimerStart()
        t0 = time.perf_counter()
        for _ in range(K):
            tp.Advance(1 / 60); ps.Update(f); f += 1
        t1 = time.perf_counter()
        gpu_ms = ctx.TimerStop()
        t2 = time.perf_counter()
        ctx.Sync()
        t3 = time.perf_counter()
        rows.append(((t3 - t0) * 1e6, gpu_ms * 1e3, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6))
    r = np.median(np.array(rows), axis=0)
    print("K = %3d: block %.1f us on the host clock = %.2f us/step; events %.1f us = %.2f us/step; enqueue loop %.1f us, wait for the stop event %.1f us, "
          "closing sync %.1f us; block - events = %.1f us" % (K, r[0], r[0] / K, r[1], r[1] / K, r[2], r[3], r[4], r[0] - r[1]))
