"""Per-block step times of one configuration of tools/step_ab.py's child (every block printed): python tools/step_blocks.py <cs> <chunks> <spawner 0|1> [blocks]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cs, chunks, spawner = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3]))
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 12
sys.argv = ["x"]
import bench
from illuminant_amd import abi, scenes
from illuminant_amd import _host as H
ctx = H.DeviceContext(0)
P = bench.build_particle_system(H, ctx, scenes, abi, cs, chunks, 0, with_spawner=spawner)
ps, tp = P["ps"], P["tp"]
f = 0
for _ in range(5):
    tp.Advance(1 / 60); ps.Update(f); f += 1
out = []
for b in range(blocks):
    ctx.TimerStart()
    for _ in range(20):
        tp.Advance(1 / 60); ps.Update(f); f += 1
    out.append(ctx.TimerStop() / 20 * 1e3)
print("cs=%d chunks=%d spawner=%d us/step per block of 20:" % (cs, chunks, spawner), " ".join("%.1f" % t for t in out))
