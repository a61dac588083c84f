"""Host-side cost of one ParticleSystem.Update (host mirror -> ilm_system_step -> hipLaunchKernel): steps of a tiny system, where the GPU
work (a few microseconds) hides behind the CPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from illuminant_amd import abi, native, scenes
from illuminant_amd import _host as H

ctx = H.DeviceContext(0)
for (cs, chunks) in ((64, 1), (256, 16)):
    P = bench.build_particle_system(H, ctx, scenes, abi, cs, chunks, 0)
    ps, tp = P["ps"], P["tp"]
    f = 0
    for _ in range(50):
        tp.Advance(1 / 60); ps.Update(f); f += 1
    ctx.Sync()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        tp.Advance(1 / 60); ps.Update(f); f += 1
    t1 = time.perf_counter()
    ctx.Sync()
    t2 = time.perf_counter()
    print("chunk size %d x %d chunks: enqueue %.2f us/step, with the final sync %.2f us/step" % (cs, chunks, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
