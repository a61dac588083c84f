"""Experiment: can two PROCESSES form a two-rank RCCL communicator on ONE GPU (the lease has one)?  If RCCL admits it, the rank-mode
paths of csrc/group.hip -- ncclCommInitRank from a shared id, the host all-gather, the send / receive exchange of unequal strips --
run between real ranks here.  Each rank: its own process, device 0.     python tools/rccl_two_ranks_one_gpu_probe.py"""
import os, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, path):
    import numpy as np
    from illuminant_amd import abi, native
    if rank == 0:
        uid = native.Group.unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(path + ".tmp", path)
    else:
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 60:
                raise SystemExit("no id")
            time.sleep(0.01)
        uid = open(path, "rb").read()
    g = native.Group.rank(0, rank, world, uid)
    print("rank %d: communicator of %d rank(s)" % (rank, g.comm_ranks()), flush=True)
    got = g.host_all_gather(bytes([10 + rank] * 8))
    assert [b[0] for b in got] == [10 + r for r in range(world)], got
    # unequal strips of a small lightmap, exchanged range by range
    w, h = 64, 96
    glm = native.GroupLightmap(g, w, h, abi.LIGHTMAP_FLOAT4)
    strips = [(0, 64), (64, 96)] if world == 2 else None
    glm.set_strips(strips)
    b, e = glm.strips[rank]
    lm = glm.members[0]
    frame = np.zeros((lm.height, w, 4), np.float32)
    frame[b:e] = 100.0 * (rank + 1) + np.arange(b, e, dtype=np.float32)[:, None, None]
    lm.upload(frame)
    glm.gather(native.GATHER_RCCL)
    g.sync()
    out = lm.download()[:h]
    want = np.zeros((h, w, 4), np.float32)
    for r, (bb, ee) in enumerate(strips):
        want[bb:ee] = 100.0 * (r + 1) + np.arange(bb, ee, dtype=np.float32)[:, None, None]
    assert np.array_equal(out, want), "rank %d: the exchanged frame differs" % rank
    print("rank %d: host all-gather and the send / receive exchange of unequal strips are correct" % rank, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        worker(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])
        sys.exit(0)
    path = os.path.join(tempfile.gettempdir(), "ilm_probe_%d.id" % os.getpid())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), "2", path], env=env) for r in range(2)]
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=120))
        except subprocess.TimeoutExpired:
            p.kill(); rcs.append("timeout")
    if os.path.exists(path):
        os.remove(path)
    print("exit codes:", rcs)
