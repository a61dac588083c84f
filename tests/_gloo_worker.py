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


class StubGroup:
    """Stands in for native.Group in bench.py's N > 1 record assembly: the one thing the ranks say to each other outside the data path is
    host_all_gather (bytes in, a list of `world` bytes objects out, also a barrier) -- here over gloo instead of RCCL."""

    def __init__(self, world):
        self.world, self.calls = world, 0

    def host_all_gather(self, local_bytes):
        self.calls += 1
        mine = torch.tensor(list(local_bytes), dtype=torch.uint8)
        everyone = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(everyone, mine)
        return [bytes(t.tolist()) for t in everyone]


class StubGroupLightmap:
    """native.GroupLightmap's strip table: set_strips is a collective that fails on every rank when the ranks disagree
    (ilm_group_lightmap_set_strips, csrc/group.hip)."""

    def __init__(self, group, height):
        self.group, self.height, self.strips = group, height, None

    def set_strips(self, strips):
        import struct
        import zlib
        strips = [tuple(int(v) for v in s) for s in strips]
        assert strips[0][0] == 0 and strips[-1][1] == self.height and all(a[1] == b[0] for a, b in zip(strips[:-1], strips[1:]))
        h = zlib.crc32(repr(strips).encode())
        everyone = [struct.unpack("<I", b)[0] for b in self.group.host_all_gather(struct.pack("<I", h))]
        assert all(v == h for v in everyone), "the ranks installed different strip tables"
        self.strips = strips


class StubTimerContext:
    """ctx.TimerStart / TimerStop of the host mirror (HIP events there, the host clock here)."""

    def TimerStart(self):
        import time
        self.t0 = time.perf_counter()

    def TimerStop(self):
        import time
        return (time.perf_counter() - self.t0) * 1e3


def check_bench_record(rank, world):
    """bench.py's N > 1 branch without a GPU (VERDICT r04 #1d): the collectives helper, the timed blocks, the strip balancing loop, the
    scaling block and the final record assembly run at world 2 with the stubs above; the oracle supplies the "work".  A Python-level slip
    in any of them fails here instead of wasting the one multi-GPU hardware run."""
    import json
    import bench

    group = StubGroup(world)
    ranks = bench.Ranks(group, rank, world, sync=lambda: None)
    ranks.barrier()
    assert ranks.doubles(10.0 + rank) == [10.0 + r for r in range(world)]
    assert ranks.max(float(rank)) == world - 1 and ranks.sum(1.5) == 1.5 * world
    solo = ranks.solo()
    calls = group.calls
    solo.barrier()
    assert solo.doubles(3.0) == [3.0] and solo.world == 1 and group.calls == calls, "a solo phase must not enter a collective"

    # timed blocks -> particle rows (the oracle steps a small system; what matters is the bookkeeping)
    cs, n = 16, 256
    rnd = scenes.randomness_table(7)
    pos, vel, attr = scenes.make_particles(5 + rank, n)
    chunk = [pos, vel, attr, np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)]
    d = abi.StepDesc()
    d.FirstChunk, d.ChunkCount = 0, -1
    d.System = scenes.system_uniforms(cs, friction=0.1, max_velocity=2048.0, life_decay=0.01)
    d.Update = abi.UpdateParams.default()
    d.UpdateMode = abi.UPDATE_POSITIONS
    ctx = StubTimerContext()
    k = 3
    blocks = bench.time_blocks(ctx, ranks, lambda: orc.step([chunk], cs, rnd, d), k, 3)
    assert len(blocks) == 3 and blocks == sorted(blocks) and all(w > 0 and g > 0 for w, g in blocks)
    walls = ranks.doubles(blocks[1][0])
    assert len(set(walls)) == 1, "the wall time of a block is the max over ranks: the same on every rank"
    share = bench.particle_row(world, n, k, blocks, "oracle stand-in")
    share_solo = bench.particle_row(1, n, k, bench.time_blocks(ctx, solo, lambda: orc.step([chunk], cs, rnd, d), k, 3), "oracle stand-in")
    full = bench.particle_row(1, 8 * n, k, blocks, "oracle stand-in"); full["workload"] = "stand-in for cfg4 whole on one GPU"
    strong = bench.particle_row(1, int(ranks.sum(4 * n)), k, blocks, "oracle stand-in"); strong["workload"] = "stand-in for 64 / world chunks per rank"
    assert share["mparticle_steps_per_s"] > 0 and share["roofline"]["bytes_per_unit"] == 112 and share["timed_blocks"]["steps_per_block"] == k
    per_rank = ranks.doubles(n * k / (blocks[1][1] * 1e-3) / 1e6)

    # the strip balancing loop over a small lit frame: "ms" of a strip = the oracle's SDF sample count over it
    w, h = 96, 80
    layout = scenes.DistanceFieldLayout(128, 128, 64.0, 6, 0.5)
    atlas = scenes.build_sdf_atlas(layout, scenes.random_obstacles(3, 6, (128, 128), 6.0, 20.0, 30.0))
    dfu = layout.uniforms(power=0.7, min_step_size=1.0, long_step_factor=0.5)
    lights = scenes.random_lights(4, 6, w, h, z=(8.0, 32.0), radius=6.0, ramp=(30.0, 60.0))
    env = scenes.environment()
    tex = orc.make_texture(atlas, abi.SDF_UNORM16)

    def time_strip(b, e):
        if e <= b:
            return 0.0
        _, st = orc.render_sphere_lights(lights, env, dfu, None, tex, (0.05, 0.05, 0.05, 1.0), w, h, row_begin=b, row_end=e, want_stats=True)
        return float(st.SdfSamples) * 1e-3

    glm = StubGroupLightmap(group, h)
    history = bench.balance_strips(glm, ranks, h, list(lights), time_strip)
    assert len(history) == 3 and history[0]["how"] == "footprint model" and "ms" in history[0] and "ms" in history[1] and "ms" not in history[2]
    assert [tuple(s_) for s_ in history[-1]["strips"]] == glm.strips and len(glm.strips) == world
    strips_ms = ranks.doubles(time_strip(*glm.strips[rank]))
    frames = {"cfg3": {"strips": [list(s_) for s_ in glm.strips], "strip_ms": strips_ms, "strip_ms_max": max(strips_ms), "strip_ms_sum": sum(strips_ms),
                       "exchange_ms": 0.1, "composited_frame_ms": max(strips_ms) + 0.1, "one_gpu_frame_ms": sum(strips_ms)}}

    sd = bench.scaling_detail(world, share=share, share_solo=share_solo, full_64m_solo=full, strong_64m=strong, per_rank_rates=per_rank, frames=frames)
    assert sd["ranks"] == world and len(sd["particles"]["per_rank_mparticle_steps_per_s"]) == world
    assert abs(sd["particles"]["strong_vs_one_gpu_64m"] - strong["mparticle_steps_per_s"] / full["mparticle_steps_per_s"]) < 1e-3
    assert abs(sd["particles"]["weak_vs_share"] - share["mparticle_steps_per_s"] / (world * share_solo["mparticle_steps_per_s"])) < 1e-3
    assert sd["frames"]["cfg3"]["strip_ms_sum"] == sum(strips_ms)
    # without the 64 M rows (--no-cfg4-64m, or a rank that is not rank 0) the block still assembles
    assert "strong_vs_one_gpu_64m" not in bench.scaling_detail(world, share, share_solo, None, None, per_rank, {})["particles"]

    # the record: cfg2 headline + rows -> the N > 1 shape (cfg4's share becomes the headline, cfg2 the secondary row, the tail order)
    cfg2_roofline = {"bound": "hbm", "achieved": 1.0, "peak": bench.HBM_PEAK_GBS, "unit": "GB/s", "frac": 1.0 / bench.HBM_PEAK_GBS, "traffic": None}
    light_row = {"roofline": {"launch_ms": 1.0, "frac": 0.5, "achieved": 1.0, "peak": 2.0, "traffic": None}, "timed_frames": 4, "lit_mpixels_per_s": 1.0, "without_gbuffer_ms": None,
                 "verified_counts": True, "work_bound": {"useful_frac": 0.4, "instructions_per_sample": 46.0}, "algorithmic_rate": {"gsamples_per_s": 1.0}, "scaling": frames["cfg3"]}
    out = {"metric": "m", "value": 123.0, "unit": "Mparticle-steps/s", "n_gpus": world, "steps": k, "warmup": 0, "ms_per_step": 0.5, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": "cfg2 stand-in", "particles_per_gpu": n},
           "timed_blocks": {"blocks": 3}, "roofline": cfg2_roofline, "cfg4_share_8m_particles": share, "cfg4_full_64m_one_gpu": full,
           "lighting": {"cfg3_1080p_64_lights_unorm16": light_row, "cfg5_4k_256_lights_fp16": light_row}, "lit_mpixels_per_s": 1.0, "scaling_detail": sd}
    rec = bench.finalize_record(out, world, False, 20.0)
    line = json.dumps(rec)
    back = json.loads(line)
    assert back["value"] == share["mparticle_steps_per_s"] and back["ms_per_step"] == share["ms_per_step"] and back["scaling"] == "weak"
    assert back["cfg2_weak_row"]["mparticle_steps_per_s"] == 123.0 and back["config"]["workload"].startswith("cfg4")
    assert back["roofline"]["resident"].startswith("hbm") and back["timed_blocks"] == share["timed_blocks"]
    keys = list(rec.keys())
    assert keys[-1] == "summary" and keys[-2] == "scaling_detail" and keys.index("roofline") > keys.index("lighting")
    assert set(back["summary"]) >= {"particles_cfg2", "particles_cfg4_share", "particles_cfg4_64m_one_gpu", "lighting_cfg3", "lighting_cfg5"}
    assert back["summary"]["lighting_cfg5"]["verified_counts"] is True
    for key in ("strip_ms_max", "strip_ms_sum", "exchange_ms", "composited_frame_ms"):
        assert key in back["scaling_detail"]["frames"]["cfg3"]
    # every rank assembled the same record from the gathered numbers (rank 0 prints it)
    digest = torch.tensor([float(len(line)), back["value"], back["scaling_detail"]["particles"]["weak_vs_share"]], dtype=torch.float64)
    everyone = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(everyone, digest)
    assert all(torch.equal(e[1:2], digest[1:2]) for e in everyone)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        check_particles(rank, world)
        check_lighting(rank, world)
        check_bench_record(rank, world)
        dist.barrier()
    finally:
        dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
