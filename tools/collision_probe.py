#!/usr/bin/env python3
"""Run ON THE GPU BOX: the collision update (row a10) timed through the interpreter and through step_lean_df_kernel with 1 / 2 / 4 units per
wave -- cfg2's 1 M particles (16 chunks of 256^2, cache-resident) and cfg4's per-GPU share (8 chunks of 1024^2, 0.67 GB, HBM-resident), the
demo's field (SimpleParticles.cs:210-284).  One process; the variants are launch decisions (ILM_DF_LEAN, ILM_DF_UNITS are read per launch).
    python tools/collision_probe.py [--sizes 1m,8m] [--reps 7]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                             # noqa: E402
from illuminant_amd import abi, native, scenes          # noqa: E402
from illuminant_amd import _host as H                   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1m,8m")
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--variants", default="", help="';'-separated environment settings, e.g. 'ILM_DF_LEAN=0;ILM_DF_UNITS=4,ILM_STEP_STREAMING=0'")
    args = ap.parse_args()
    ctx = H.DeviceContext(0)
    scene = bench.build_collision_scene(H, ctx, scenes, abi)
    dt = 1.0 / 60.0
    variants = [("interpreter", {"ILM_DF_LEAN": "0"}), ("lean K=1", {"ILM_DF_UNITS": "1"}), ("lean K=2", {"ILM_DF_UNITS": "2"}), ("lean K=4", {"ILM_DF_UNITS": "4"}),
                ("lean default", {})]
    if args.variants:
        variants = [(v, dict(kv.split("=") for kv in v.split(",") if kv)) for v in args.variants.split(";")]
    for size in args.sizes.split(","):
        cs, nch = (256, 16) if size == "1m" else ((1024, 8) if size == "8m" else (1024, 64))
        for label, env in [("plain UpdatePositions", None)] + variants:
            for k in ("ILM_DF_LEAN", "ILM_DF_UNITS", "ILM_STEP_STREAMING", "ILM_DF_CELLS0"):
                os.environ.pop(k, None)
            os.environ.update(env or {})
            S = bench.build_particle_system(H, ctx, scenes, abi, cs, nch, 0, with_spawner=False, replicate_cfg4_images=(size == "64m"))
            ps, tp = S["ps"], S["tp"]
            if env is not None:
                col = H.ParticleCollision()
                col.DistanceField = scene["field"]
                col.DistanceFieldMaximumZ = 256.0
                col.LifePenalty = 1.0
                cfgc = ps.Configuration
                cfgc.Collision = col
                ps.Configuration = cfgc
            f = 0
            for _ in range(12):
                tp.Advance(dt); ps.Update(f); f += 1
            samples = 0
            if env is not None:
                out_n = C.c_uint64(0)
                native.check(native.lib().ilm_debug_step_sdf_samples(ctx.Handle, 1, None))
                tp.Advance(dt); ps.Update(f); f += 1
                native.check(native.lib().ilm_debug_step_sdf_samples(ctx.Handle, 0, C.byref(out_n)))
                samples = int(out_n.value)
            times = []
            for _ in range(args.reps):
                ctx.Sync()
                ctx.TimerStart()
                for _ in range(args.steps):
                    tp.Advance(dt); ps.Update(f); f += 1
                times.append(ctx.TimerStop() / args.steps)
            times.sort()
            us = times[len(times) // 2] * 1e3
            alg = S["live"] * 112 + samples * 32
            print("%-4s %-22s %8.2f us (min %8.2f)  samples/particle %.3f  algorithmic %.0f GB/s" % (size, label, us, times[0] * 1e3, samples / S["live"], alg / (us * 1e-6) / 1e9), flush=True)
            del S, ps
    for k in ("ILM_DF_LEAN", "ILM_DF_UNITS", "ILM_STEP_STREAMING", "ILM_DF_CELLS0"):
        os.environ.pop(k, None)


if __name__ == "__main__":
    main()
