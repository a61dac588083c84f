"""The N > 1 path on CPU: world_size 2 over `gloo` (bench.py runs the same host logic over RCCL on the GPUs)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from illuminant_amd import scenes, sharding

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_row_strips_cover_the_frame_in_tile_bands():
    for h in (16, 17, 1080, 2160, 100):
        for world in (1, 2, 3, 4, 8):
            strips = sharding.row_strips(h, world)
            assert len(strips) == world and strips[0][0] == 0 and strips[-1][1] == h
            for r in range(world - 1):
                assert strips[r][1] == strips[r + 1][0]
                assert strips[r][1] % sharding.TILE_ROWS == 0 or strips[r][1] == h
            assert sum(e - b for b, e in strips) == h
    # 4K over 8 GPUs: 135 bands of 16 rows -> 16 or 17 bands each (8.3 MB half4 strips)
    sizes = [e - b for b, e in sharding.row_strips(2160, 8)]
    assert max(sizes) - min(sizes) <= sharding.TILE_ROWS and sum(sizes) == 2160


def test_balanced_strips_follow_the_lights():
    h, w = 1080, 1920
    # every light in the top quarter of the frame: the balanced split puts its cuts there, the equal split does not
    lights = scenes.random_lights(9, 32, w, h // 4, z=(8.0, 64.0), radius=24.0, ramp=(40.0, 80.0))
    eq = sharding.row_strips(h, 4)
    bal = sharding.balanced_row_strips(h, 4, lights)
    assert bal[0][0] == 0 and bal[-1][1] == h and all(bal[r][1] == bal[r + 1][0] for r in range(3))
    cost = sharding.light_row_cost(lights, h) + 1.0
    spread = lambda strips: np.ptp([cost[b:e].sum() for b, e in strips])
    assert spread(bal) < spread(eq)
    assert bal[0][1] < eq[0][1]


def test_strips_recut_from_measured_times():
    """rebalance_row_strips: contiguous whole-band strips covering the frame, equal cost under the piecewise-constant model it assumes
    (so a second pass with the model's own prediction changes nothing), strips that took longer get shorter."""
    h, world = 2160, 8
    strips = sharding.row_strips(h, world)
    seconds = [1.0, 1.2, 1.3, 1.5, 1.6, 1.4, 1.1, 0.9]
    cut = sharding.rebalance_row_strips(strips, seconds, h)
    assert cut[0][0] == 0 and cut[-1][1] == h and all(a[1] == b[0] for a, b in zip(cut[:-1], cut[1:]))
    assert all(b % 16 == 0 for b, _ in cut) and all(e > b for b, e in cut)
    density = np.zeros(h)
    for (b, e), t in zip(strips, seconds):
        density[b:e] = t / (e - b)
    predicted = [density[b:e].sum() for b, e in cut]
    assert max(predicted) / min(predicted) < 1.08           # equal up to the 16-row granularity
    assert (cut[4][1] - cut[4][0]) < (strips[4][1] - strips[4][0]) and (cut[7][1] - cut[7][0]) > (strips[7][1] - strips[7][0])
    again = sharding.rebalance_row_strips(cut, predicted, h)
    assert max(abs(a[0] - b[0]) for a, b in zip(again, cut)) <= 16
    # degenerate inputs: no time measured, one rank
    assert sharding.rebalance_row_strips(strips, [0.0] * world, h) == strips
    assert sharding.rebalance_row_strips([(0, h)], [3.0], h) == [(0, h)]


def test_chunk_ownership_is_a_partition():
    for n in (0, 1, 7, 16, 64):
        for world in (1, 2, 8):
            owned = [sharding.owned_chunks(n, r, world) for r in range(world)]
            assert sorted(c for o in owned for c in o) == list(range(n))
            assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1


def test_world_size_2_over_gloo():
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2", ILM_ORACLE_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=600)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out[-4000:])
        assert "rank %d ok" % rank in out


def test_bench_refuses_a_rank_count_that_does_not_match_gpus():
    """`--gpus N` must never be printed next to fewer ranks: with a launcher's WORLD_SIZE that disagrees the bench stops before
    touching a device."""
    root = os.path.dirname(HERE)
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "refusing to report a 2-GPU number from 1 rank" in (p.stderr + p.stdout)
    assert '"n_gpus"' not in p.stdout
