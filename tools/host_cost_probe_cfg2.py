"""Host-side cost of bench.py's cfg2 step (host mirror -> ilm_system_step): enqueue time per step against the time until the GPU is idle,
and what a step costs that has to add a chunk (the Spawner fills a 256^2 chunk every 60 steps)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from illuminant_amd import abi, native, scenes
from illuminant_amd import _host as H
ctx = H.DeviceContext(0)
P = bench.build_particle_system(H, ctx, scenes, abi, 256, 16, 0)
ps, tp = P["ps"], P["tp"]
f = 0
for _ in range(25):
    tp.Advance(1 / 60); ps.Update(f); f += 1
ctx.Sync()
for n in (20, 40, 80):
    ctx.Sync()
    t0 = time.perf_counter()
    for _ in range(n):
        tp.Advance(1 / 60); ps.Update(f); f += 1
    t1 = time.perf_counter()
    ctx.Sync()
    t2 = time.perf_counter()
    print("%d steps: enqueue %.2f us/step, until idle %.2f us/step (chunks %d)" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6, len(ps.Chunks)))
# every step on its own: enqueue and completion, with the chunk count before / after
rows = []
for _ in range(200):
    before = len(ps.Chunks)
    ctx.Sync()
    t0 = time.perf_counter()
    tp.Advance(1 / 60); ps.Update(f); f += 1
    t1 = time.perf_counter()
    ctx.Sync()
    t2 = time.perf_counter()
    rows.append((len(ps.Chunks) - before, (t1 - t0) * 1e6, (t2 - t0) * 1e6))
rows = np.array(rows)
grow = rows[:, 0] > 0
print("single synchronised steps: enqueue median %.1f us, complete median %.1f us; the %d steps that added a chunk: enqueue %s us, complete %s us"
      % (np.median(rows[~grow, 1]), np.median(rows[~grow, 2]), int(grow.sum()), np.round(rows[grow, 1], 1).tolist(), np.round(rows[grow, 2], 1).tolist()))
