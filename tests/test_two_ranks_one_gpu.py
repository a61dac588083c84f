"""The one-process-per-GPU shape at world size 2 and 3 ON ONE GPU (r05).  RCCL refuses two ranks on one device, so until now
ilm_group_create_rank, the ncclSend / ncclRecv range exchange, the collective ilm_group_lightmap_set_strips and bench.py's whole N > 1
branch had only run at world size 1.  tests/fake_rccl.cpp is a stand-in for librccl.so.1 (the eleven entry points group.hip binds) that moves
the bytes through shared host memory between rank processes sharing GPU 0; the library loads it through ILM_RCCL_LIB.  Its semantics are
stronger than RCCL's (operations complete at ncclGroupEnd), so this proves offsets, byte counts, pairings, collective discipline and the
Python-level paths -- not RCCL's asynchrony, and not a single number."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    cmd = ["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "fake_rccl.cpp"),
           "-o", out, "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-Wl,-rpath,/opt/rocm/lib"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    return out


@pytest.mark.parametrize("world", [2, 3, 8])
def test_a_group_that_spans_processes_composites_the_frame_on_every_rank(fake_rccl, world):
    idfile = os.path.join(tempfile.mkdtemp(), "id")
    env = dict(os.environ, ILM_RCCL_LIB=fake_rccl, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_two_rank_worker.py"), str(r), str(world), idfile], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=600)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("rank %d of %d ok" % (r, world)) in out, "rank %d:\n%s" % (r, out[-4000:])


def test_bench_runs_its_whole_n_gpu_branch_at_two_ranks(fake_rccl):
    """`python bench.py --gpus 2` exactly as the driver launches it (bench.py spawns torch.distributed.run itself), both ranks on GPU 0
    (ILM_BENCH_ONE_GPU, a test hook the record names): the strips of both lit frames are cut, re-cut from measured times and exchanged,
    the instrumented counts summed over the ranks equal the oracle's totals for the whole frames, the composite call and the pipelined
    exchange reproduce the frame, and the line carries the self-describing scaling block."""
    env = dict(os.environ, ILM_RCCL_LIB=fake_rccl, ILM_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--light-frames", "2", "--light-ms", "0",
                        "--sustain-s", "0"], env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0's)"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["ranks"] == 2 and d["config"]["rccl_communicator_ranks"] == 2
    assert d["config"]["rccl_communicator_ranks_per_rank"] == [2, 2] and "one_gpu_stand_in" in d["config"]
    sd = d["scaling_detail"]
    # r06: the rows with the collectives BASELINE config 4 names run last under the watchdog; the headline is the step WITH the live-count
    # all-gather, the communication-free figure beside it; the liveness table is bit-exact on every rank (every chunk of 1024^2 live) and
    # every rank's gathered Pos+Life planes are their owners'
    wl, wg = sd["particles"]["with_live_counts"], sd["particles"]["with_position_all_gather"]
    assert d["config"]["workload"].startswith("cfg4") and d["value"] == wl["mparticle_steps_per_s"] > 0, wl
    assert d["value_without_collectives"] == d["cfg4_share_8m_particles"]["mparticle_steps_per_s"] > 0 and "ilm_group_live_counts" in d["config"]["collective_in_the_timed_steps"]
    assert wl["live_count_table"] == {"chunks": 16, "every_chunk_live": 1048576, "bit_exact_on_every_rank": True} and wl["live_count_calls_per_block"] >= 0.2, wl
    assert wg["gathered_chunks_match_their_owners"] is True and wg["exchange_ms"] > 0 and wg["bytes_received_per_rank"] >= 8 * 4 * 1024 * 1024 * 4, wg
    assert sd["ranks"] == 2 and len(sd["particles"]["per_rank_mparticle_steps_per_s"]) == 2
    assert sd["particles"]["strong_vs_one_gpu_64m"] > 0 and sd["particles"]["weak_vs_share"] > 0
    assert sd["particles"]["sharded_64m"]["workload"].startswith("64 chunks of 1024^2 over 2 rank(s): 32 chunks")
    pinned = json.load(open(os.path.join(ROOT, "tests", "golden", "full_frame_bands.json")))
    for key, pin in (("cfg3_1080p_64_lights_unorm16", "cfg3"), ("cfg5_4k_256_lights_fp16", "cfg5")):
        row = d["lighting"][key]
        assert row["verified_counts"] is True and row["sdf_samples_per_frame"] == pinned[pin]["sdf_samples"]       # summed over the two strips
        assert row["exchange"]["ranks"] == 2 and row["exchange"]["composite_call_matches"] is True
        strips = row["exchange"]["strips"]
        assert len(strips) == 2 and strips[0][0] == 0 and strips[0][1] == strips[1][0] and strips[1][1] == pinned[pin]["height"] and strips[0][1] % 16 == 0
        assert len(row["strip_balancing"]) == 3 and all(len(h["ms"]) == 2 for h in row["strip_balancing"][:2])
        fs = sd["frames"][pin]
        assert len(fs["strip_ms"]) == 2 and fs["strip_ms_max"] >= max(fs["strip_ms"]) - 1e-9 and fs["exchange_ms"] > 0 and fs["one_gpu_frame_ms"] > 0
        assert fs["pipelined_exchange"].get("both_lightmaps_hold_the_same_frame") is True, fs["pipelined_exchange"]
        assert fs["store_mode"].get("every_rank_holds_the_frame_of_the_rccl_exchange") is True and fs["store_mode"]["composited_frame_ms"] > 0, fs["store_mode"]


def test_a_hang_in_an_optional_row_costs_the_record_that_row_only(fake_rccl):
    """bench.py runs the optional frames of scaling_detail (pipelined exchange, store mode) LAST and under a watchdog thread: with the last
    rank stuck in front of them (ILM_BENCH_HANG_OPTIONAL, a test hook) the other rank blocks in the first collective, the watchdog fires,
    rank 0 prints the record as it stood -- every mandatory figure, the count checks, the serial composited frames -- with a note, and
    every rank leaves with status 0."""
    env = dict(os.environ, ILM_RCCL_LIB=fake_rccl, ILM_BENCH_ONE_GPU="1", ILM_BENCH_HANG_OPTIONAL="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--light-frames", "2", "--light-ms", "0",
                        "--sustain-s", "0", "--no-cfg4-64m", "--optional-rows-timeout", "8"], env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    sd = d["scaling_detail"]
    assert "did not finish" in sd["optional_rows"]
    assert "with_live_counts" not in sd["particles"] and d["value"] == d["value_without_collectives"] == d["cfg4_share_8m_particles"]["mparticle_steps_per_s"]
    assert d["config"]["collective_in_the_timed_steps"].startswith("none")
    for pin in ("cfg3", "cfg5"):
        assert sd["frames"][pin]["composited_frame_ms"] > 0 and "store_mode" not in sd["frames"][pin] and "pipelined_exchange" not in sd["frames"][pin]
    assert d["lighting"]["cfg5_4k_256_lights_fp16"]["verified_counts"] is True and d["value"] > 0


def test_the_collective_schedule_of_the_n_gpu_branch_is_printed_in_issue_order(fake_rccl):
    """`bench.py --gpus 2 --dry-collectives` walks the whole N > 1 branch once and prints every collective in issue order with its bytes
    (profiles/r06_collective_schedule.txt is this list at world 8): the two particle collectives, the strip tables, the lightmap exchanges,
    the store mode's arming -- and the optional phase comes last."""
    env = dict(os.environ, ILM_RCCL_LIB=fake_rccl, ILM_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-collectives", "--no-cfg4-64m"], env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    rows = [ln.split(" | ") for ln in p.stdout.splitlines() if " | " in ln and not ln.startswith("#")]
    assert len(rows) > 20 and not any(ln.startswith("{") for ln in p.stdout.splitlines())
    entry = [r[2] for r in rows]
    for name in ("ilm_group_host_all_gather", "ilm_group_live_counts", "ilm_group_gather_chunks(components 0..3)", "ilm_group_lightmap_set_strips", "ilm_group_lightmap_gather(RCCL)",
                 "ilm_group_lightmap_gather(STORE)", "ilm_group_lightmap_store_mode(1)"):
        assert name in entry, name
    assert [int(r[0]) for r in rows] == list(range(1, len(rows) + 1))
    # the rows with particle collectives and the optional exchange variants come LAST (under the watchdog): nothing mandatory after them
    last_phase = [("ilm_group_" in r[1] or "optional" in r[1]) for r in rows]
    first_optional = last_phase.index(True)
    assert all(last_phase[first_optional:]) and not any(last_phase[:first_optional]) and first_optional > 10
    chunks = [r for r in rows if r[2].startswith("ilm_group_gather_chunks")]
    assert all(int(r[4]) >= 8 * 4 * 1024 * 1024 * 4 for r in chunks)
