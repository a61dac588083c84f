"""Worker of tests/test_sharding_gloo.py: one of WORLD_SIZE processes on the `gloo` backend (CPU).

Exercises the N > 1 host logic of illuminant_amd/sharding.py exactly as bench.py uses it on RCCL, with the CPU
oracle standing in for the kernels (tests may use the oracle; the product path never does).  Any mismatch raises.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from illuminant_amd import abi, scenes, sharding
from oracle import oracle as orc


def check_particles(rank, world):
    cs, n_chunks = 16, 5
    n = cs * cs
    rnd = scenes.randomness_table(7)
    pos, vel, attr = scenes.make_particles(42, n * n_chunks, dead_fraction=0.3)

    def chunk_planes(c):
        sl = slice(c * n, (c + 1) * n)
        return [pos[sl].copy(), vel[sl].copy(), attr[sl].copy(), np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)]

    d = abi.StepDesc()
    d.FirstChunk, d.ChunkCount = 0, -1
    d.System = scenes.system_uniforms(cs, friction=0.1, max_velocity=2048.0, life_decay=20.0)   # some particles die in the step
    d.Update = abi.UpdateParams.default()
    d.OpCount = 1
    d.Ops[0].Type = abi.OP_GRAVITY
    d.Ops[0].u.Gravity = scenes.gravity_params([((128.0, 128.0, 0.0), 150.0, 60.0, 1)])
    d.UpdateMode = abi.UPDATE_POSITIONS
    d.Flags = abi.STEP_COUNT_LIVE

    # single-process answer over the whole table
    whole = [chunk_planes(c) for c in range(n_chunks)]
    want_counts = orc.step(whole, cs, rnd, d, want_counts=True)

    # this rank's shard: chunk_index mod world, stepped with no exchange
    mine = sharding.owned_chunks(n_chunks, rank, world)
    assert all(sharding.chunk_owner(c, world) == rank for c in mine)
    local = [chunk_planes(c) for c in mine]
    local_counts = orc.step(local, cs, rnd, d, want_counts=True)
    for i, c in enumerate(mine):
        for k in range(5):
            assert np.array_equal(local[i][k], whole[c][k]), ("chunk", c, "plane", k)

    counts = sharding.gather_live_counts(local_counts, n_chunks, rank, world, dist)
    assert counts.dtype == np.uint32 and np.array_equal(counts, want_counts), (counts, want_counts)
    assert 0 < int(counts.sum()) < n * n_chunks
    # every chunk has exactly one owner
    owners = sorted(c for r in range(world) for c in sharding.owned_chunks(n_chunks, r, world))
    assert owners == list(range(n_chunks))


def check_lighting(rank, world):
    w, h = 96, 80
    layout = scenes.DistanceFieldLayout(128, 128, 64.0, 6, 0.5)
    atlas = scenes.build_sdf_atlas(layout, scenes.random_obstacles(3, 6, (128, 128), 6.0, 20.0, 30.0))
    dfu = layout.uniforms(power=0.7, min_step_size=1.0, long_step_factor=0.5)
    lights = scenes.random_lights(4, 6, w, h, z=(8.0, 32.0), radius=6.0, ramp=(30.0, 60.0))
    env = scenes.environment()
    tex = orc.make_texture(atlas, abi.SDF_UNORM16)
    ambient = (0.05, 0.05, 0.05, 1.0)
    want, _ = orc.render_sphere_lights(lights, env, dfu, None, tex, ambient, w, h)

    # the layout bench.py uses: equal padded slots, one in-place all-gather
    R, strips = sharding.padded_row_strips(h, world)
    assert R % sharding.TILE_ROWS == 0 and world * R >= h and strips[-1][1] == h
    b, e = strips[rank]
    part, _ = orc.render_sphere_lights(lights, env, dfu, None, tex, ambient, w, h, row_begin=b, row_end=e)
    full = torch.zeros((world * R, w, 4), dtype=torch.float32)
    full[b:e] = torch.from_numpy(part[b:e])
    sharding.all_gather_rows(full, strips, rank, dist)
    assert np.array_equal(full[:h].numpy(), want), "padded in-place all-gather differs from the single-process frame"

    for strips in (sharding.row_strips(h, world), sharding.balanced_row_strips(h, world, lights), [(0, 48), (48, 80)][:world] if world == 2 else None):
        if strips is None:
            continue
        # strips tile the frame in whole tile bands
        assert strips[0][0] == 0 and strips[-1][1] == h
        for r in range(world - 1):
            assert strips[r][1] == strips[r + 1][0] and strips[r][1] % sharding.TILE_ROWS == 0
        b, e = strips[rank]
        part, _ = orc.render_sphere_lights(lights, env, dfu, None, tex, ambient, w, h, row_begin=b, row_end=e)
        full = torch.zeros((h, w, 4), dtype=torch.float32)
        full[b:e] = torch.from_numpy(part[b:e])
        sharding.all_gather_rows(full, strips, rank, dist)
        assert np.array_equal(full.numpy(), want), "gathered lightmap differs from the single-process frame"
        # the same strips through the point-to-point range exchange bench.py's N > 1 frames use (group.hip exchange_ranges)
        full2 = torch.zeros((h, w, 4), dtype=torch.float32)
        full2[b:e] = torch.from_numpy(part[b:e])
        sharding.exchange_row_ranges(full2, strips, rank, dist)
        assert np.array_equal(full2.numpy(), want), "range exchange differs from the single-process frame"


    # bench.py's N > 1 path re-cuts the strips from what they cost: every rank times its own strip, the times are all-gathered, every
    # rank computes the same table (sharding.rebalance_row_strips) -- here the "time" of a strip is the oracle's SDF-sample count over it
    strips = sharding.balanced_row_strips(h, world, lights)
    for _ in range(2):
        b, e = strips[rank]
        _, st = orc.render_sphere_lights(lights, env, dfu, None, tex, ambient, w, h, row_begin=b, row_end=e, want_stats=True)
        mine = torch.tensor([float(st.SdfSamples)], dtype=torch.float64)
        everyone = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(everyone, mine)
        strips = sharding.rebalance_row_strips(strips, [float(t[0]) for t in everyone], h)
        assert strips[0][0] == 0 and strips[-1][1] == h and all(strips[r][1] == strips[r + 1][0] for r in range(world - 1))
    b, e = strips[rank]
    part, _ = orc.render_sphere_lights(lights, env, dfu, None, tex, ambient, w, h, row_begin=b, row_end=e)
    full3 = torch.zeros((h, w, 4), dtype=torch.float32)
    full3[b:e] = torch.from_numpy(part[b:e])
    sharding.exchange_row_ranges(full3, strips, rank, dist)
    assert np.array_equal(full3.numpy(), want), "the frame over re-cut strips differs from the single-process frame"


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        check_particles(rank, world)
        check_lighting(rank, world)
        dist.barrier()
    finally:
        dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
