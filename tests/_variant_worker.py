"""Worker of tests/test_fuzz_regressions_gpu.py: runs in its own process so that a VARIANT build of the library can be loaded
(ILM_HIP_LIB, here illuminant_amd/lib/libilluminant_hip_gravity_exact.so = particles.hip compiled with -DILM_GRAVITY_EXACT: Gravity's IEEE
sqrt / division form).  Replays the TRANSFORMS of a fuzz seed's particle step (its Gravity op; UpdateMode = ILM_UPDATE_NONE -- the Update
pass has divisions of its own and is not what the variant changes) from the ORACLE's post-spawn state, so the device and the oracle start
from the same bits, and prints one JSON line about one slot of chunk 1 and about the chunk.
    ILM_HIP_LIB=... python tests/_variant_worker.py <seed> <slot>"""
import copy
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from illuminant_amd import abi, native          # noqa: E402
from oracle import oracle as orc                # noqa: E402
from tests import fuzz_scenes                   # noqa: E402


def post_spawn_state(seed):
    """(cs, rnd, chunks after the ORACLE's spawn pass, the step descriptor without its spawner)"""
    cs, rnd, chunks, d = fuzz_scenes.particle_step_of_seed(seed)
    assert d.SpawnCount == 1 and d.Spawns[0].ChunkIndex == 1
    orc.spawn(chunks[1][0], chunks[1][1], chunks[1][2], cs, rnd, d.Spawns[0].Params)
    d0 = copy.copy(d)
    d0.SpawnCount = 0
    return cs, rnd, chunks, d0


def device_step(ctx, cs, rnd, chunks, d):
    eng = native.Engine(ctx, cs, rnd)
    sysm = native.System(eng)
    for c in range(len(chunks)):
        sysm.add_chunk()
        for plane, a in ((abi.PLANE_POSITION, chunks[c][0]), (abi.PLANE_VELOCITY, chunks[c][1]), (abi.PLANE_ATTRIBUTES, chunks[c][2])):
            sysm.upload(c, plane, a)
    sysm.step(d)
    got = [[sysm.download(c, plane) for plane in (abi.PLANE_POSITION, abi.PLANE_VELOCITY, abi.PLANE_ATTRIBUTES, abi.PLANE_RENDER_COLOR, abi.PLANE_RENDER_DATA)]
           for c in range(len(chunks))]
    counts = sysm.step_counts() if (int(d.Flags) & abi.STEP_COUNT_LIVE) else None
    sysm.close(); eng.close()
    return got, counts


def main():
    seed, slot = int(sys.argv[1]), int(sys.argv[2])
    cs, rnd, chunks, d0 = post_spawn_state(seed)
    d0.UpdateMode = abi.UPDATE_NONE
    d0.Flags = 0
    assert d0.Ops[0].Type == abi.OP_GRAVITY
    d0.OpCount = 1                                   # the Gravity op alone (a Noise op behind it is not what the variant changes)
    ctx = native.Context(0)
    got, _ = device_step(ctx, cs, rnd, chunks, d0)
    want = [[a.copy() for a in c] for c in chunks]
    orc.step(want, cs, rnd, d0)
    gv, wv = got[1][1], want[1][1]
    live = want[1][0][:, 3] > 0
    print(json.dumps({"lib": os.path.basename(native.LIB_PATH), "slot_velocity_bits_equal": bool(np.array_equal(gv[slot].view(np.uint32), wv[slot].view(np.uint32))),
                      "slot_velocity": [float(x) for x in gv[slot]], "oracle_velocity": [float(x) for x in wv[slot]],
                      "chunk_velocity_elements_differing": int((gv[live].view(np.uint32) != wv[live].view(np.uint32)).sum()),
                      "chunk_velocity_elements": int(live.sum()) * 4}))
    ctx.close()


if __name__ == "__main__":
    main()
