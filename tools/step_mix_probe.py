"""Instruction mix per phase of the step kernel: the same 16 chunks of 256^2 stepped with different op lists (5 launches each), in a fixed
order, so that the per-dispatch rows of a rocprofv3 --pmc run can be told apart by position.
  order: [update only] [gravity] [noise] [gravity+noise] [transforms only: gravity+noise, UpdateMode NONE]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from illuminant_amd import abi, native, scenes

cs, n_chunks = 256, 16
ctx = native.Context(0)
eng = native.Engine(ctx, cs, scenes.randomness_table(7)); sysm = native.System(eng)
for c in range(n_chunks):
    sysm.add_chunk()
    pos, vel, attr = scenes.make_particles(10 + c, cs * cs, pos_lo=(0, 0, 0), pos_hi=(1920, 1080, 32), life=(50.0, 90.0))
    sysm.upload(c, abi.PLANE_POSITION, pos); sysm.upload(c, abi.PLANE_VELOCITY, vel); sysm.upload(c, abi.PLANE_ATTRIBUTES, attr)
grav = scenes.gravity_params([((400., 300., 0.), 70., 600., 1), ((1500., 300., 0.), 150., 900., 1),
                              ((400., 800., 0.), 200., 1200., 1), ((1500., 800., 0.), 100., 1500., 1)], 1024.0)
noise = scenes.noise_params(scenes.area_none(), (0.37 * 253, 0.81 * 127), (0.12 * 253, 0.55 * 127), 0.35)


def desc(ops, mode=abi.UPDATE_POSITIONS):
    d = abi.StepDesc(); d.FirstChunk, d.ChunkCount = 0, -1
    d.System = scenes.system_uniforms(cs, friction=0.02, max_velocity=2048.0, life_decay=0.01)
    d.Update = abi.UpdateParams.default(); d.UpdateMode = mode
    d.OpCount = len(ops)
    for i, (t, p) in enumerate(ops):
        d.Ops[i].Type = t
        if t == abi.OP_GRAVITY: d.Ops[i].u.Gravity = p
        else: d.Ops[i].u.Noise = p
    return d


G, N = (abi.OP_GRAVITY, grav), (abi.OP_NOISE, noise)
for d in (desc([]), desc([G]), desc([N]), desc([G, N]), desc([G, N], abi.UPDATE_NONE)):
    for _ in range(5):
        sysm.step(d)
    ctx.sync()
print("done")
