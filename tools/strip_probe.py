"""Run ON THE GPU BOX.  What one rank of an n-GPU lit frame launches, measured on ONE GPU (a stand-in: no multi-GPU node is involved):
every strip of cfg3 / cfg5 rendered alone -- equal bands and the cost-balanced strips bench.py's N > 1 path cuts -- under each light-split
setting (ilm_ctx_set_light_split: workgroups per tile), next to the whole frame; and the bit-equality the split promises: every strip
and every setting must reproduce the one-workgroup whole frame exactly.

    python tools/strip_probe.py [--ranks 8] [--splits 1,0,2,4,8] [--frames 30] [--scenes cfg3,cfg5]        (0 = the library's own choice)

The scenes are tests/test_properties_gpu.py's cfg3 / cfg5 frames (ground plane, no G-buffer bound unless --gbuffer), fields generated on the device.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from illuminant_amd import abi, native, scenes, sharding  # noqa: E402


def build(ctx, name):
    if name == "cfg3":
        w, h, nl = 1920, 1080, 64
        layout = scenes.DistanceFieldLayout(2048, 2048, 128.0, 32, 0.25)
        fmt = abi.SDF_UNORM16
        obstacles = scenes.random_obstacles(11, 256, (2048, 2048))
        dfu = layout.uniforms(power=0.7, min_step_size=1.0, long_step_factor=0.5)
        lights = scenes.random_lights(12, nl, w, h)
    else:
        w, h, nl = 3840, 2160, 256
        layout = scenes.DistanceFieldLayout(4096, 4096, 128.0, 32, 0.125, 128)
        fmt = abi.SDF_FP16
        obstacles = scenes.random_obstacles(11, 256, (4096, 4096))
        dfu = layout.uniforms(max_cone_radius=24.0, power=0.7, step_limit=64, min_step_size=1.0, long_step_factor=0.5)
        lights = scenes.random_lights(13, nl, w, h, z=(8.0, 64.0), radius=24.0, ramp=(400.0, 1100.0))
    sdf = native.DistanceFieldTexture(ctx, None, fmt, size=(layout.atlas_width, layout.atlas_height))
    sdf.render_slices(scenes.render_desc(layout), list(range(0, layout.slice_count, 3)), scenes.obstruction_array([(t - 1, c, s) for (t, c, s) in obstacles]))
    return w, h, dfu, lights, sdf


def timed(ctx, fn, frames):
    fn(); fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(frames):
        fn()
    return ctx.timer_stop() / frames


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--splits", default="1,0,2,4,8")
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--scenes", default="cfg3,cfg5")
    ap.add_argument("--gbuffer", action="store_true", help="bind the ground-plane G-buffer (half4) as the bench's configured frames do")
    args = ap.parse_args()
    splits = [int(v) for v in args.splits.split(",")]
    ctx = native.Context(0)
    env = scenes.environment()
    ambient = (0.05, 0.05, 0.05, 1.0)
    print("# one GPU standing in for each rank of a %d-rank frame; times in ms per launch (HIP events, %d launches each)" % (args.ranks, args.frames))
    for name in args.scenes.split(","):
        w, h, dfu, lights, sdf = build(ctx, name)
        gb = None
        if args.gbuffer:
            gb = native.GBufferTexture(ctx, scenes.ground_plane_gbuffer(w, h, abi.GBUFFER_HALF4), abi.GBUFFER_HALF4)
            env = scenes.environment(gbuffer_size=(w, h))
        lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_HALF4)
        ref = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
        ctx.set_light_split(1)
        native.render_sphere_lights(ctx, lights, env, dfu, gb, sdf, ambient, ref)
        whole_bits = ref.download()
        tables = {"equal": sharding.row_strips(h, args.ranks), "balanced": sharding.balanced_row_strips(h, args.ranks, lights)}
        for split in splits:
            ctx.set_light_split(split)
            t_whole = timed(ctx, lambda: native.render_sphere_lights(ctx, lights, env, dfu, gb, sdf, ambient, lm), max(4, args.frames // (4 if name == "cfg5" else 1)))
            # the bits: whole frame and the balanced strips under this setting against the one-workgroup frame
            native.render_sphere_lights(ctx, lights, env, dfu, gb, sdf, ambient, ref)
            same_whole = bool(np.array_equal(ref.download(), whole_bits))
            ref.clear()
            for (b, e) in tables["balanced"]:
                native.render_sphere_lights(ctx, lights, env, dfu, gb, sdf, ambient, ref, b, e)
            same_strips = bool(np.array_equal(ref.download(), whole_bits))
            print("%s split=%s whole %.4f ms   bits: whole %s, strips %s" % (name, split if split else "auto", t_whole, "equal" if same_whole else "DIFFER",
                                                                         "equal" if same_strips else "DIFFER"), flush=True)
            def measure(table):
                return [timed(ctx, lambda b=b, e=e: native.render_sphere_lights(ctx, lights, env, dfu, gb, sdf, ambient, lm, b, e), args.frames) for (b, e) in table]

            def show(kind, table, ts):
                print("%s split=%s %-9s strips: %s   max %.4f  sum %.4f  (whole / %d = %.4f)  rows %s" % (
                    name, split if split else "auto", kind, " ".join("%.4f" % t for t in ts), max(ts), sum(ts), args.ranks, t_whole / args.ranks,
                    " ".join(str(e - b) for (b, e) in table)), flush=True)
            for kind, table in tables.items():
                ts = measure(table)
                show(kind, table, ts)
            # cost-balanced by MEASUREMENT: the footprint model's strips re-cut from the times just measured (what an N-rank run does
            # with the ranks' own strip times, sharding.rebalance_row_strips), twice
            table, ts = tables["balanced"], measure(tables["balanced"])
            for round_ in (1, 2):
                table = sharding.rebalance_row_strips(table, ts, h)
                ts = measure(table)
                show("measured%d" % round_, table, ts)
            ref.clear()
            for (b, e) in table:
                native.render_sphere_lights(ctx, lights, env, dfu, gb, sdf, ambient, ref, b, e)
            assert np.array_equal(ref.download(), whole_bits), "re-cut strips changed the frame"
        for x in (lm, ref, sdf):
            x.close()
        if gb is not None:
            gb.close()
    ctx.close()


if __name__ == "__main__":
    t0 = time.time()
    main()
    print("# %.0f s" % (time.time() - t0))
