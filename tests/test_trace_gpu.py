"""ILM_TRACE=1: every data-path entry point of the C ABI runs inside a named roctx range (the reference's
RenderTrace.Marker calls, Illuminant/Particles/ParticleSystem.cs:464-469, Illuminant/Lighting/LightingRenderer.cs:1123-1124).  The ranges
are for a profiler to see (tools/marker_trace.sh -> profiles/); here: the roctx library binds, push and pop balance (roctxRangePop returns
the depth that is left), and results do not change."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import ctypes, sys
sys.path.insert(0, %r)
import numpy as np
from illuminant_amd import abi, native, scenes
roctx = None
for name in ("librocprofiler-sdk-roctx.so.1", "libroctx64.so.4"):
    try:
        roctx = ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL); break
    except OSError:
        pass
assert roctx is not None, "no roctx library on this box"
ctx = native.Context(0)
cs = 32
eng = native.Engine(ctx, cs, scenes.randomness_table(7))
sysm = native.System(eng); sysm.add_chunk()
pos, vel, attr = scenes.make_particles(3, cs * cs, dead_fraction=0.2)
for plane, a in ((abi.PLANE_POSITION, pos), (abi.PLANE_VELOCITY, vel), (abi.PLANE_ATTRIBUTES, attr)):
    sysm.upload(0, plane, a)
d = abi.StepDesc(); d.FirstChunk, d.ChunkCount = 0, -1
d.System = scenes.system_uniforms(cs); d.Update = abi.UpdateParams.default(); d.UpdateMode = abi.UPDATE_POSITIONS; d.Flags = abi.STEP_COUNT_LIVE
sysm.step(d)
counts = sysm.step_counts()
lm = native.Lightmap(ctx, 64, 48)
lights = scenes.random_lights(4, 3, 64, 48, z=(8.0, 32.0), radius=6.0, ramp=(30.0, 60.0))
native.render_sphere_lights(ctx, lights, scenes.environment(), scenes.DistanceFieldLayout(128, 128, 64.0, 6, 0.5).uniforms(), None, None, (0.1, 0.1, 0.1, 1.0), lm)
img = lm.download()
# every range the calls above pushed has been popped: a push of our own sits at depth 0, its pop leaves 0
roctx.roctxRangePushA.argtypes = [ctypes.c_char_p]
depth = roctx.roctxRangePushA(b"test")
left = roctx.roctxRangePop()
print("RESULT", int(counts[0]), float(img.sum()), depth, left)
'''


def run(trace):
    env = dict(os.environ)
    env.pop("ILM_TRACE", None)
    if trace:
        env["ILM_TRACE"] = "1"
    p = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "no roctx library could be bound" not in p.stderr
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][-1].split()
    return int(line[1]), float(line[2]), int(line[3]), int(line[4])


def test_trace_ranges_balance_and_change_nothing():
    off = run(False)
    on = run(True)
    assert off[:2] == on[:2] and off[0] > 0
    assert on[2] == on[3] == 0 or (on[2] >= 0 and on[3] == on[2]), on      # roctxRangePushA returns the level it opened, roctxRangePop the level left
