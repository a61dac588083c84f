"""Run ON THE GPU BOX (alone or under tools/pmc_kernels.sh): cfg2's particle system stepped `steps` times (the attractors cluster it), then
rasterised onto a 1920 x 1080 RGBA8 target `frames` times.     python tools/raster_one.py [steps] [frames]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from illuminant_amd import abi, native, scenes  # noqa: E402
from tests.test_properties_gpu import cfg2_step  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 220
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cs, n_chunks = 256, 16
ctx = native.Context(0)
eng = native.Engine(ctx, cs, scenes.randomness_table(7))
sysm = native.System(eng)
for c in range(n_chunks):
    sysm.add_chunk()
    pos, vel, attr = scenes.make_particles(100 + c, cs * cs, pos_lo=(0, 0, 0), pos_hi=(1920, 1080, 32), life=(50.0, 60.0))
    for plane, data in ((abi.PLANE_POSITION, pos), (abi.PLANE_VELOCITY, vel), (abi.PLANE_ATTRIBUTES, attr)):
        sysm.upload(c, plane, data)
desc = cfg2_step(cs)
desc.System = scenes.system_uniforms(cs, friction=0.02, max_velocity=2048.0, life_decay=0.01, size=(4.0, 4.0))
for _ in range(steps):
    sysm.step(desc)
target = native.Lightmap(ctx, 1920, 1080, abi.LIGHTMAP_RGBA8)
params = scenes.rasterize_params(size=(4.0, 4.0))
st = native.render_particles(sysm, params, target, want_stats=True)
ctx.sync()
ctx.timer_start()
for _ in range(frames):
    native.render_particles(sysm, params, target)
ms = ctx.timer_stop() / frames
print("after %d steps: %d quads, %d (quad, tile) pairs, %d shaded pixels: %.4f ms per frame" % (steps, st[0], st[1], st[2], ms))
