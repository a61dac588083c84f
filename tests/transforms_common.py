"""The closed-form cases of tests/golden/transforms_ext.json run through a backend (the CPU oracle or the HIP path):
both test files call check_case(case, backend) so the oracle and the kernels are pinned by the same data."""
import json
import os

import numpy as np

from illuminant_amd import abi, scenes
from tests.util import assert_close

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CS = 16     # chunk size of the fixture scenes (256 slots)


def load_cases():
    with open(os.path.join(GOLDEN, "transforms_ext.json")) as f:
        return json.load(f)["cases"]


def filled(value, n=CS * CS):
    a = np.zeros((n, 4), np.float32)
    a[:] = np.asarray(value, np.float32)
    return a


class OracleBackend:
    """Runs one IlmStepDesc on host arrays with the oracle."""

    def __init__(self, oracle):
        self.orc = oracle

    def run(self, desc, rnd, chunk, spawn_positions=None, source_chunk=None, spawn_pattern=None):
        planes = [chunk[0].copy(), chunk[1].copy(), chunk[2].copy(), np.zeros_like(chunk[0]), np.zeros_like(chunk[0])]
        fb = {0: source_chunk} if source_chunk is not None else None
        sp = {0: spawn_positions} if spawn_positions is not None else None
        pt = {0: spawn_pattern} if spawn_pattern is not None else None
        self.orc.step([planes], CS, rnd, desc, spawn_positions=sp, feedback_sources=fb, spawn_patterns=pt)
        return planes


class GpuBackend:
    """Runs the same descriptor through ilm_system_step."""

    def __init__(self, ctx):
        from illuminant_amd import native
        self.native = native
        self.ctx = ctx

    def run(self, desc, rnd, chunk, spawn_positions=None, source_chunk=None, spawn_pattern=None):
        native = self.native
        eng = native.Engine(self.ctx, CS, rnd)
        sysm = native.System(eng)
        sysm.add_chunk()
        for plane, data in zip((abi.PLANE_POSITION, abi.PLANE_VELOCITY, abi.PLANE_ATTRIBUTES), chunk):
            sysm.upload(0, plane, data)
        src = None
        if source_chunk is not None:
            src = native.System(eng)
            src.add_chunk()
            for plane, data in zip((abi.PLANE_POSITION, abi.PLANE_VELOCITY, abi.PLANE_ATTRIBUTES), source_chunk):
                src.upload(0, plane, data)
            desc.Spawns[0].Feedback.SourceSystem = src.handle.value
            desc.Spawns[0].Feedback.SourceChunkIndex = 0
        if spawn_positions is not None:
            sysm.set_spawn_positions(0, spawn_positions)
        if spawn_pattern is not None:
            sysm.set_spawn_pattern(0, spawn_pattern)
        sysm.step(desc)
        out = [sysm.download(0, p) for p in (abi.PLANE_POSITION, abi.PLANE_VELOCITY, abi.PLANE_ATTRIBUTES)]
        for x in (src, sysm, eng):
            if x is not None:
                x.close()
        return out


def base_desc(dt):
    d = abi.StepDesc()
    d.FirstChunk, d.ChunkCount = 0, -1
    d.System = scenes.system_uniforms(CS, dt_seconds=dt)
    d.Update = abi.UpdateParams.default()
    d.UpdateMode = abi.UPDATE_NONE
    return d


def check_case(case, backend):
    kind = case["kind"]
    rnd = scenes.randomness_table(7)
    if kind == "matrix_multiply":
        d = base_desc(case["dt"])
        d.OpCount = 1
        d.Ops[0].Type = abi.OP_MATRIX_MULTIPLY
        m = lambda v: abi.Matrix.from_rows([v[0:4], v[4:8], v[8:12], v[12:16]])
        d.Ops[0].u.MatrixMultiply = scenes.matrix_multiply_params(scenes.area_none(case["strength"]), m(case["position_matrix"]),
                                                                  m(case["velocity_matrix"]), case["cycles_per_second"])
        out = backend.run(d, rnd, (filled(case["position"]), filled(case["velocity"]), filled([1, 1, 1, 1])))
        assert_close(out[0][37], case["expected_position"], "matrix multiply position", rtol=1e-5)
        assert_close(out[1][37], case["expected_velocity"], "matrix multiply velocity", rtol=1e-5)
    elif kind == "spatial_noise":
        d = base_desc(case["dt"])
        d.OpCount = 1
        d.Ops[0].Type = abi.OP_SPATIAL_NOISE
        noise = scenes.noise_params(scenes.area_none(1.0), (0.3 * 253, 0.6 * 127), (0.8 * 253, 0.1 * 127), 0.4, case["cycles_per_second"],
                                    case["replace_old_velocity"], position=((-0.5,) * 4, (0,) * 4, tuple(case["position_scale"])),
                                    velocity=((-0.5,) * 3, (0,) * 3, tuple(case["velocity_scale"])), speed=(-0.5, 0.0, case["speed_scale"]))
        d.Ops[0].u.SpatialNoise = scenes.spatial_noise_params(noise, case["space_scale"])
        table = np.full_like(rnd, case["table_value"])
        out = backend.run(d, table, (filled(case["position"]), filled(case["velocity"]), filled([1, 1, 1, 1])))
        assert_close(out[0][5], case["expected_position"], "spatial noise position", rtol=1e-5)
        assert_close(out[1][5], case["expected_velocity"], "spatial noise velocity", rtol=1e-5)
    elif kind == "position_buffer":
        d = base_desc(1.0 / 60.0)
        p, buf = scenes.position_buffer_spawn_params(CS, case["first"], case["last"], case["total_spawned"], (0.2 * 253, 0.7 * 127),
                                                     case["positions"], life_constant=case["life"],
                                                     position=((0, 0, 0), (0, 0, 0), (0, 0, 0), scenes.FORMULA_LINEAR),
                                                     velocity=((0, 0, 0), (0, 0, 0), (0, 0, 0), scenes.FORMULA_LINEAR), life=(case["life"], 0.0, 0.0))
        d.SpawnCount = 1
        d.Spawns[0].ChunkIndex = 0
        d.Spawns[0].Kind = abi.SPAWN_POSITION_BUFFER
        d.Spawns[0].Params = p
        old = filled([-1, -1, -1, 0])
        out = backend.run(d, rnd, (old, filled([0, 0, 0, 0]), filled([0, 0, 0, 0])), spawn_positions=buf)
        for e in case["expected"]:
            assert_close(out[0][e["slot"]], e["position"], "position buffer slot %d" % e["slot"], rtol=1e-6)
        untouched = [i for i in range(CS * CS) if not (case["first"] <= i <= case["last"])]
        assert np.array_equal(out[0][untouched], old[untouched])
    elif kind == "feedback":
        d = base_desc(1.0 / 60.0)
        p = scenes.spawn_params(CS, case["first"], case["last"], 0, (0.2 * 253, 0.7 * 127),
                                position=((0, 0, 0), (0, 0, 0), (0, 0, 0), scenes.FORMULA_LINEAR),
                                velocity=((0, 0, 0), (0, 0, 0), (0, 0, 0), scenes.FORMULA_LINEAR), life=(1.5, 0.0, 0.0))
        d.SpawnCount = 1
        d.Spawns[0].ChunkIndex = 0
        d.Spawns[0].Kind = abi.SPAWN_FEEDBACK
        d.Spawns[0].Params = p
        d.Spawns[0].Feedback = scenes.feedback_params(0, 0, case["source_index"], case["instance_multiplier"], case["source_velocity_factor"],
                                                      source_life_range=case["source_life_range"])
        n = CS * CS
        src_pos = np.zeros((n, 4), np.float32)
        src_pos[:, 0] = np.arange(n) * 3.0
        src_pos[:, 1] = 1000.0 - np.arange(n)
        src_pos[:, 2] = 7.0
        src_pos[:, 3] = 2.0
        for dead in case["dead_source_slots"]:
            src_pos[dead, 3] = 0.25          # below SourceLifeRange.x
        src_vel = np.zeros((n, 4), np.float32)
        src_vel[:, 0] = 10.0 + np.arange(n)
        src_vel[:, 1] = -4.0
        src_attr = filled([1, 1, 1, 1])
        old = filled([-1, -1, -1, 0])
        out = backend.run(d, rnd, (old, filled([0, 0, 0, 0]), filled([0, 0, 0, 0])), source_chunk=(src_pos, src_vel, src_attr))
        for slot, source in case["expected_source_of_slot"].items():
            slot = int(slot)
            if source in case["dead_source_slots"]:
                assert np.array_equal(out[0][slot], old[slot]), "slot %d must keep its contents" % slot
                continue
            assert_close(out[0][slot], [src_pos[source, 0], src_pos[source, 1], src_pos[source, 2], 1.5], "feedback position %d" % slot, rtol=1e-6)
            assert_close(out[1][slot][:3], src_vel[source, :3] * case["source_velocity_factor"], "feedback velocity %d" % slot, rtol=1e-6)
    elif kind == "pattern":
        d = base_desc(1.0 / 60.0)
        levels = [np.asarray(l, np.float32) for l in case["texture_levels"]]
        th, tw = levels[0].shape[0], levels[0].shape[1]
        p = scenes.spawn_params(CS, case["first"], case["last"], 0, (0.2 * 253, 0.7 * 127),
                                position=(tuple(case["position_constant"]), (0, 0, 0), (0, 0, 0), scenes.FORMULA_LINEAR),
                                velocity=((0, 0, 0), (0, 0, 0), (0, 0, 0), scenes.FORMULA_LINEAR), life=(case["life"], 0.0, 0.0),
                                color=(tuple(case["color_constant"]), (0, 0, 0, 0), (0, 0, 0, 0)))
        d.SpawnCount = 1
        d.Spawns[0].ChunkIndex = 0
        d.Spawns[0].Kind = abi.SPAWN_PATTERN
        d.Spawns[0].Params = p
        d.Spawns[0].Pattern = scenes.pattern_params(tw, th, case["divisor"], case["current_row"], multiply_color_constant=case["multiply"])
        old = filled([-1, -1, -1, 0])
        old_attr = filled([9, 9, 9, 9])
        out = backend.run(d, rnd, (old, filled([0, 0, 0, 0]), old_attr), spawn_pattern=levels)
        for e in case["expected"]:
            assert_close(out[0][e["slot"]], e["position"], "%s: position of slot %d" % (case["name"], e["slot"]), rtol=1e-6)
            assert_close(out[2][e["slot"]], e["attributes"], "%s: attributes of slot %d" % (case["name"], e["slot"]), rtol=1e-5)
        written = {e["slot"] for e in case["expected"]}
        untouched = [i for i in range(CS * CS) if i not in written]
        assert set(case["rejected_slots"]) <= set(untouched)
        assert np.array_equal(out[0][untouched], old[untouched]) and np.array_equal(out[2][untouched], old_attr[untouched])
    elif kind == "aligned_spawn":
        n = CS * CS
        first, last = case["first"], case["last"]
        ps, vs = case["position_scale"], case["velocity_scale"]

        def run(align):
            d = base_desc(1.0 / 60.0)
            constant = tuple(case.get("position_constant", (0.0, 0.0, 0.0)))
            p = scenes.spawn_params(CS, first, last, 0, (0.2 * 253, 0.7 * 127),
                                    position=(constant, (ps, ps, ps), (0, 0, 0), scenes.FORMULA_SPHERICAL),
                                    velocity=((0, 0, 0), (vs, vs, vs), (0, 0, 0), scenes.FORMULA_SPHERICAL), life=(1.5, 0.0, 0.0), align=align)
            d.SpawnCount = 1
            d.Spawns[0].ChunkIndex = 0
            d.Spawns[0].Params = p
            old = filled([-1, -1, -1, 0])
            if case["spawner"] == "feedback":
                d.Spawns[0].Kind = abi.SPAWN_FEEDBACK
                d.Spawns[0].Feedback = scenes.feedback_params(0, 0, 0, 1, 0.0, source_life_range=(0.5, 9999.0))
                src_pos = filled([3.0, 4.0, 5.0, 2.0])
                out = backend.run(d, rnd, (old, filled([0, 0, 0, 0]), filled([0, 0, 0, 0])),
                                  source_chunk=(src_pos, filled([0, 0, 0, 0]), filled([1, 1, 1, 1])))
                centre = np.tile(np.float32([3.0, 4.0, 5.0]), (n, 1))      # AlignPositionConstant: the constant (0) + the source's position
            else:
                tw, th = case["texture_size"]
                d.Spawns[0].Kind = abi.SPAWN_PATTERN
                d.Spawns[0].Pattern = scenes.pattern_params(tw, th, 1, 0, multiply_color_constant=True)
                levels = [np.ones((th, tw, 4), np.float32)]
                out = backend.run(d, rnd, (old, filled([0, 0, 0, 0]), filled([9, 9, 9, 9])), spawn_pattern=levels)
                k = np.arange(n) - first
                centre = np.zeros((n, 3), np.float32)
                centre[:, 0] = constant[0] + (k % tw) - tw * 0.5
                centre[:, 1] = constant[1] + (k // tw) - th * 0.5
                centre[:, 2] = constant[2]
            slots = np.arange(first, last + 1)
            offset = out[0][slots, :3].astype(np.float64) - centre[slots]
            velocity = out[1][slots, :3].astype(np.float64)
            assert (out[0][slots, 3] == 1.5).all(), "every slot of the range spawned"
            cosine = (offset * velocity).sum(axis=1) / np.maximum(np.linalg.norm(offset, axis=1) * np.linalg.norm(velocity, axis=1), 1e-30)
            return cosine
        aligned = run(True)
        assert (aligned > 1.0 - 1e-5).all(), "%s spawner: velocity must point along the position offset when aligned (min cosine %.6f)" % (
            case["spawner"], float(aligned.min()))
        loose = run(False)
        assert (loose < 0.99).mean() > 0.8, "unaligned directions are unrelated"
    else:
        raise AssertionError("unknown fixture kind %r" % kind)
