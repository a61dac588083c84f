"""Run ON THE GPU BOX: bench.py's particle-light frame (4 096 particle lights, 1080p, cfg3's field) through the lane-queue walk and through
the entry-by-entry walk (ILM_PL_QUEUE=0, read per launch) of ONE library -- the two lightmaps compared bit for bit, then timed in
alternating blocks.   python tools/particle_lights_ab.py [blocks] [frames_per_block] [lights] [fp16]
    ILM_HIP_LIB=tools/ab/<tag>/libilluminant_hip.so LD_LIBRARY_PATH=tools/ab/<tag> python tools/particle_lights_ab.py   (a variant build)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from illuminant_amd import abi, native, scenes  # noqa: E402
from illuminant_amd import _host as H  # noqa: E402

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 5
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n_lights = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
fmt = abi.SDF_FP16 if (len(sys.argv) > 4 and sys.argv[4] == "fp16") else abi.SDF_UNORM16

ctx = H.DeviceContext(0)
L = bench.build_lighting(H, ctx, scenes, abi, 1920, 1080, 0, 0.25, 2048, fmt)
chunk = 64
while chunk * chunk < n_lights:
    chunk *= 2
eng = H.ParticleEngine(ctx, H.ParticleEngineConfiguration(chunk), scenes.randomness_table(7))
pcfg = H.ParticleSystemConfiguration()
pcfg.LifeDecayPerSecond = 0.01
lsys = H.ParticleSystem(eng, pcfg)
pos, vel, attr = scenes.make_particles(91, n_lights, pos_lo=(0, 0, 4), pos_hi=(1920, 1080, 48), life=(50.0, 90.0))
lsys.Spawn(n_lights, pos, vel, attr)
lsys.Update(0)
pls = H.ParticleLightSource()
tmpl = H.SphereLightSource()
tmpl.Radius = 4.0; tmpl.RampLength = 60.0; tmpl.Color = [1.0, 0.9, 0.8, 1.0]
pls.Template = tmpl
pls.System = lsys
L["env"].ParticleLights = [pls]
r = L["renderer"]


def frame(queue, stats=False):
    os.environ["ILM_PL_QUEUE"] = "1" if queue else "0"
    return r.RenderLighting(1.0, 0, -1, stats)


def texels():
    ctx.Sync()
    return np.array(r.ReadLightmap(), copy=True)


st = frame(False, True)
ref_stats = texels()
frame(False); old = texels()
frame(True); new = texels()
print("stats (samples, pairs, traced):", [int(x) for x in st[:3]])
print("entry-by-entry walk == statistics variant:", bool(np.array_equal(old.view(np.uint8), ref_stats.view(np.uint8))))
eq = bool(np.array_equal(old.view(np.uint8), new.view(np.uint8)))
print("lane queue == entry-by-entry walk, every bit:", eq)
if not eq:
    a, b = old.reshape(-1), new.reshape(-1)
    d = np.flatnonzero(a != b)
    print("  differing elements: %d of %d; first: %s" % (d.size, a.size, [(int(i), int(a[i]), int(b[i])) for i in d[:8]]))
for _ in range(30):
    frame(True)
ctx.Sync()
res = {True: [], False: []}
for _ in range(blocks):
    for q in (False, True):
        ctx.TimerStart()
        for _ in range(frames):
            frame(q)
        res[q].append(ctx.TimerStop() / frames)
for q in (False, True):
    v = sorted(res[q])
    print("%-22s ms per frame: median %.4f  min %.4f  max %.4f" % ("lane queue" if q else "entry-by-entry walk", v[len(v) // 2], v[0], v[-1]))
