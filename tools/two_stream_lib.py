"""Experiment: does stepping two halves of the chunk table on two streams (no cross-stream sync: chunks never interact) hide the
ramp / tail of the small cfg2 launch?  Two contexts = two HIP streams on the same GPU."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from illuminant_amd import abi, native, scenes


def make(ctx, cs, n_chunks, seed, spawn):
    rnd = scenes.randomness_table(7)
    eng = native.Engine(ctx, cs, rnd); sysm = native.System(eng)
    n = cs * cs
    for c in range(n_chunks):
        sysm.add_chunk()
        pos, vel, attr = scenes.make_particles(seed + c, n, pos_lo=(0, 0, 0), pos_hi=(1920, 1080, 32), life=(50.0, 90.0))
        sysm.upload(c, abi.PLANE_POSITION, pos); sysm.upload(c, abi.PLANE_VELOCITY, vel); sysm.upload(c, abi.PLANE_ATTRIBUTES, attr)
    tgt = sysm.add_chunk() if spawn else -1
    d = abi.StepDesc(); d.FirstChunk, d.ChunkCount = 0, -1
    d.System = scenes.system_uniforms(cs, friction=0.02, max_velocity=2048.0, life_decay=0.01)
    d.Update = abi.UpdateParams.default(); d.UpdateMode = abi.UPDATE_POSITIONS
    d.OpCount = 2
    d.Ops[0].Type = abi.OP_GRAVITY
    d.Ops[0].u.Gravity = scenes.gravity_params([((400., 300., 0.), 70., 600., 1), ((1500., 300., 0.), 150., 900., 1),
                                                ((400., 800., 0.), 200., 1200., 1), ((1500., 800., 0.), 100., 1500., 1)], 1024.0)
    d.Ops[1].Type = abi.OP_NOISE
    d.Ops[1].u.Noise = scenes.noise_params(scenes.area_none(), (0.37 * 253, 0.81 * 127), (0.12 * 253, 0.55 * 127), 0.35)
    if spawn:
        d.SpawnCount = 1; d.Spawns[0].ChunkIndex = tgt
        d.Spawns[0].Params = scenes.spawn_params(cs, 0, 1091, 0, (0.42 * 253, 0.77 * 127),
            position=((960, 540, 0), (900, 450, 0), (0, 0, 0), 1), velocity=((0, 0, 0), (60, 60, 60), (0, 0, 0), 1), life=(50.0, 2.7, 0))
    return eng, sysm, d


def run(parts, steps=300):
    for _ in range(20):
        for (c, s, d) in parts: s.step(d)
    for (c, s, d) in parts: c.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for (c, s, d) in parts: s.step(d)
    for (c, s, d) in parts: c.sync()
    return (time.perf_counter() - t0) / steps * 1e6


