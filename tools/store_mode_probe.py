"""Run ON THE GPU BOX.  ILM_GATHER_STORE on ONE GPU standing in for the 8 members of a group (all members on device 0: the mirror stores
go to local memory instead of over xGMI, so this measures what the seven extra stores per texel cost the kernel, not the wire):
every cost-balanced strip of cfg5 / cfg3 rendered by its member with the store mode off and on (HIP events per member), the peer-copy
gather that the mode replaces timed alone, and the frames compared bit for bit.
    python tools/store_mode_probe.py [members] [frames]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from illuminant_amd import abi, native, scenes, sharding  # noqa: E402
from tools.strip_probe import build  # noqa: E402

members = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ambient = (0.05, 0.05, 0.05, 1.0)
print("# %d members on one device; ms per launch, HIP events, %d launches each" % (members, frames))
for name in ("cfg3", "cfg5"):
    g = native.Group([0] * members)
    built = [build(c, name) for c in g.contexts]          # replicated inputs: one field per member
    w, h, dfu, lights = built[0][:4]
    sdfs = [b[4] for b in built]
    env = scenes.environment(gbuffer_size=(w, h))
    gbs = [native.GBufferTexture(c, scenes.ground_plane_gbuffer(w, h, abi.GBUFFER_FLOAT4), abi.GBUFFER_FLOAT4) for c in g.contexts]
    glm = native.GroupLightmap(g, w, h, abi.LIGHTMAP_HALF4)
    glm.set_strips(sharding.balanced_row_strips(h, members, lights))
    result = {}
    for mode in ("copy", "store"):
        glm.store_mode(mode == "store")
        times = []
        for i, c in enumerate(g.contexts):
            b, e = glm.strips[i]
            fn = lambda: native.render_sphere_lights(c, lights, env, dfu, gbs[i], sdfs[i], ambient, glm.members[i], b, e)
            fn(); fn(); c.sync()
            c.timer_start()
            for _ in range(frames):
                fn()
            times.append(c.timer_stop() / frames)
        g.sync()
        t0 = time.perf_counter()
        for _ in range(frames):
            glm.gather(native.GATHER_STORE if mode == "store" else native.GATHER_PEER)
        g.sync()
        exch = (time.perf_counter() - t0) / frames * 1e3
        result[mode] = (times, exch, [glm.download(i) for i in (0, members - 1)])
        print("%s %-5s strips: %s   max %.4f sum %.4f   exchange alone (host clock, all members) %.4f ms" % (
            name, mode, " ".join("%.4f" % t for t in times), max(times), sum(times), exch), flush=True)
    same = all(np.array_equal(a.view(np.uint16), b.view(np.uint16)) for a, b in zip(result["copy"][2], result["store"][2]))
    print("%s frames of member 0 and %d under both modes: %s; store mode costs the strips %+.1f %% (sum)" % (
        name, members - 1, "bit-equal" if same else "DIFFER", 100.0 * (sum(result["store"][0]) / sum(result["copy"][0]) - 1.0)), flush=True)
    glm.store_mode(False)
    glm.close()
    for x in gbs + sdfs:
        x.close()
    g.close()
