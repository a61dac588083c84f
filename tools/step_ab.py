"""A/B of the particle step on the bench's own cfg2 system (host mirror, Spawner + Gravity x4 + Noise + UpdatePositions): every variant
library (tools/ab/<tag>/libilluminant_hip.so; LD_LIBRARY_PATH is set per child process) steps a fresh system in blocks of 20 steps and
prints the median / minimum time per step of the HIP-event timer; variants are interleaved so that box-to-box spread cancels.
  python tools/step_ab.py base v1 ...        (run ON THE GPU BOX)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import bench
from illuminant_amd import abi, scenes
from illuminant_amd import _host as H
ctx = H.DeviceContext(0)
for (cs, chunks, spawner, blocks) in ((256, 16, True, 12), (256, 16, False, 12), (1024, 8, False, 4)):
    P = bench.build_particle_system(H, ctx, scenes, abi, cs, chunks, 0, with_spawner=spawner)
    ps, tp = P["ps"], P["tp"]
    f = 0
    for _ in range(5):
        tp.Advance(1 / 60); ps.Update(f); f += 1
    import time
    ts, hs = [], []
    K = int(os.environ.get("STEP_AB_K", "20"))
    for b in range(blocks):
        ctx.TimerStart()
        t0 = time.perf_counter()
        for _ in range(K):
            tp.Advance(1 / 60); ps.Update(f); f += 1
        hs.append((time.perf_counter() - t0) / K * 1e6)
        ts.append(ctx.TimerStop() / K * 1e3)
    ts.sort(); hs.sort()
    print("cs=%%d chunks=%%d spawner=%%d: median %%.2f min %%.2f us/step   (host enqueue median %%.2f us/step, K=%%d)" %% (cs, chunks, spawner, ts[len(ts) // 2], ts[0], hs[len(hs) // 2], K))
    del ps, P
''' % ROOT

for rnd in range(2):
    for tag in sys.argv[1:]:
        env = dict(os.environ)
        lib_tag = tag
        if "@" in tag:      # <library tag>@VAR=value[,VAR=value]: the same library under different environment switches
            lib_tag, settings = tag.split("@", 1)
            for kv in settings.split(","):
                k, v = kv.split("=", 1)
                env[k] = v
        env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "tools", "ab", lib_tag) + ":" + env.get("LD_LIBRARY_PATH", "")
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        for line in out.stdout.splitlines():
            if line.startswith("cs="):
                print("%-24s %s" % (tag, line))
        if out.returncode != 0:
            print(tag, "FAILED", out.stderr[-400:])
