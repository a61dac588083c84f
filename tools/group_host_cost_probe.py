"""Run ON THE GPU BOX.  What driving the members of ilm_group_create(device_ids, 8) costs the HOST: one process, eight members (all on
device 0 here -- the host side is what is timed; members sharing a device share its execution units, so the GPU-side times are NOT
those of eight devices), each with cfg2's particle system (16 chunks of 256^2, Gravity x 4 + Noise + UpdatePositions).

  * one thread steps the eight members one after the other (what a C# host that loops over its systems does, ParticleSystem.cs:743-745);
  * eight threads, one per member (contexts are independent: each has its own streams and may be driven from its own thread).

    python tools/group_host_cost_probe.py [members]         -> profiles/r04_group_host_cost.txt
"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from illuminant_amd import abi, native, scenes  # noqa: E402
from tests.test_properties_gpu import cfg2_step  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cs, n_chunks = 256, 16
group = native.Group([0] * N)
rnd = scenes.randomness_table(7)
members = []
for ctx in group.contexts:
    eng = native.Engine(ctx, cs, rnd)
    sysm = native.System(eng)
    for c in range(n_chunks):
        sysm.add_chunk()
        pos, vel, attr = scenes.make_particles(100 + c, cs * cs, pos_lo=(0, 0, 0), pos_hi=(1920, 1080, 32), life=(50.0, 60.0))
        for plane, data in ((abi.PLANE_POSITION, pos), (abi.PLANE_VELOCITY, vel), (abi.PLANE_ATTRIBUTES, attr)):
            sysm.upload(c, plane, data)
    members.append((ctx, eng, sysm))
desc = cfg2_step(cs)
desc.System = scenes.system_uniforms(cs, friction=0.02, max_velocity=2048.0, life_decay=0.01)
for (_, _, s) in members:
    for _ in range(10):
        s.step(desc)
group.sync()


def one_thread(frames):
    group.sync()
    t0 = time.perf_counter()
    for _ in range(frames):
        for (_, _, s) in members:
            s.step(desc)
    t1 = time.perf_counter()
    group.sync()
    t2 = time.perf_counter()
    return (t1 - t0) / frames * 1e6, (t2 - t0) / frames * 1e6


def thread_per_member(frames):
    group.sync()
    barrier = threading.Barrier(N + 1)
    done = [0.0] * N

    def run(i):
        s = members[i][2]
        barrier.wait()
        for _ in range(frames):
            s.step(desc)
        done[i] = time.perf_counter()
    ts = [threading.Thread(target=run, args=(i,)) for i in range(N)]
    for t in ts:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    for t in ts:
        t.join()
    t1 = max(done)
    group.sync()
    t2 = time.perf_counter()
    return (t1 - t0) / frames * 1e6, (t2 - t0) / frames * 1e6


# one member alone: what its step costs the host and the GPU when nothing else runs
ctx0, _, s0 = members[0]
ctx0.sync()
t0 = time.perf_counter()
for _ in range(200):
    s0.step(desc)
t1 = time.perf_counter()
ctx0.sync()
t2 = time.perf_counter()
print("# %d members on device 0, cfg2 systems (1 048 576 particles each); times per FRAME (= one step of every member), microseconds" % N)
print("one member alone:          enqueue %.1f us per step, until idle %.1f us per step" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
for frames in (50, 200):
    e, d = one_thread(frames)
    print("one thread, %d members:     enqueue %.1f us per frame = %.1f us per member, until idle %.1f us per frame (%d frames)" % (N, e, e / N, d, frames))
for frames in (50, 200):
    e, d = thread_per_member(frames)
    print("one thread per member:     enqueue %.1f us per frame (slowest thread), until idle %.1f us per frame (%d frames)" % (e, d, frames))
print("# the members share ONE device here: 'until idle' is %d steps' worth of GPU work on one GPU; on %d devices a frame's GPU work is one step's" % (N, N))
for (_, e, s) in members:
    s.close(); e.close()
group.close()
