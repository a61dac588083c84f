"""Where the particle step leaves the Infinity Cache (run ON THE GPU BOX):  python tools/working_set_sweep.py [> profiles/rNN_step_working_set_sweep.txt]

cfg2's transform list without the spawner (Gravity x4 + Noise + UpdatePositions, bench.build_particle_system) over systems of growing
size: 256^2-slot chunks up to ParticleSystem.MaxChunkCount (64), then 512^2 and 1024^2 chunks.  One 256^2 chunk = 65 536 slots x 80 B of
state = 5.2 MB, of which a step streams 112 B / slot (48 read, 64 written).  For every point: median and minimum time per step of
HIP-event-timed blocks, and the algorithmic rate 112 B x live slots / time.  The MI355X carries 256 MiB of Infinity Cache: working sets
below it are served from there, above it from HBM -- the knee is what this table puts on file.  ILM_STEP_LEAN / ILM_STEP_STREAMS and the
STREAM (non-temporal) variant switch are the library's own (api.hip chooses STREAM when the launch's working set exceeds the cache)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from illuminant_amd import abi, scenes  # noqa: E402
from illuminant_amd import _host as H  # noqa: E402

POINTS = [(256, c) for c in (4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48, 52, 56, 60, 64)] + \
         [(512, c) for c in (16, 20, 24, 32, 48)] + [(1024, c) for c in (8, 12, 16)]


def main():
    ctx = H.DeviceContext(0)
    print("# particle step working-set sweep: Gravity x4 + Noise + UpdatePositions, no spawner, fp32; 112 algorithmic B per live slot-step")
    print("# %-10s %-7s %-12s %-14s %-13s %-13s %-10s %-10s" % ("chunk", "chunks", "particles", "state_MB(80B)", "us/step med", "us/step min", "GB/s med", "of 8 TB/s"))
    for cs, chunks in POINTS:
        P = bench.build_particle_system(H, ctx, scenes, abi, cs, chunks, 0, with_spawner=False)
        ps, tp = P["ps"], P["tp"]
        live = P["live"]
        f = 0
        warm = 60 if live > 4e6 else 30
        for _ in range(warm):
            tp.Advance(1 / 60); ps.Update(f); f += 1
        ctx.Sync()
        k = 20
        blocks = 9 if live <= 4e6 else 5
        ts = []
        for _ in range(blocks):
            ctx.TimerStart()
            for _ in range(k):
                tp.Advance(1 / 60); ps.Update(f); f += 1
            ts.append(ctx.TimerStop() / k * 1e3)
        ts.sort()
        med, mn = ts[len(ts) // 2], ts[0]
        gbs = live * 112 / (med * 1e-6) / 1e9
        print("  %-10s %-7d %-12d %-14.1f %-13.2f %-13.2f %-10.1f %-10.4f" % ("%d^2" % cs, chunks, live, live * 80 / 1e6, med, mn, gbs, gbs / 8000.0), flush=True)
        del ps, P
        time.sleep(0.05)


if __name__ == "__main__":
    main()
