"""Run ON THE GPU BOX.  bench.py's 2.5D G-buffer frame (1080p, 2 507 triangles) alone: device time per frame (HIP events around a
run of frames) and the host's time per ilm_gbuffer_render_meshes call (the frame is whichever is longer); under
`rocprofv3 --kernel-trace --stats` the same command shows the three kernels' durations.

    python tools/gbuffer_probe.py [--frames 200]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from illuminant_amd import abi, native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    args = ap.parse_args()
    ctx = native.Context(0)
    gd, top, front, bbv = bench.gbuffer_meshes_scene()
    runs = [(None, 0, 64, abi.BILLBOARD_MASK)]
    for fmt, name in ((abi.GBUFFER_FLOAT4, "Vector4"), (abi.GBUFFER_HALF4, "HalfVector4")):
        gbt = native.GBufferTexture(ctx, None, fmt, size=(1920, 1080))
        for _ in range(3):
            gbt.render_meshes(gd, top, front, bbv, runs)
        ctx.sync()
        ctx.timer_start()
        t0 = time.perf_counter()
        for _ in range(args.frames):
            gbt.render_meshes(gd, top, front, bbv, runs)
        host = (time.perf_counter() - t0) / args.frames
        ms = ctx.timer_stop() / args.frames
        print("%s: %.4f ms per frame on the device's clock; host %.1f us per call (%d triangles, %d KB of vertices)" % (
            name, ms, host * 1e6, 2 + len(top) // 3 + len(front) // 3 + 128, (top.nbytes + front.nbytes + bbv.nbytes) // 1024))
        gbt.close()
    ctx.close()


if __name__ == "__main__":
    main()
