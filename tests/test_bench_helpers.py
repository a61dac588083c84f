"""bench.py's bookkeeping without a GPU: the committed PMC summary it quotes is parsed, the traffic figure is the profile's bytes per
slot-step scaled to the run's units, and the contract's flags exist with defaults that finish within minutes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _profile_is_stale():
    """The newest committed PMC summary carries the sha256 of the kernel sources it was taken from; between a kernel edit and the next
    profiling run (tools/profile_round.sh on the GPU box) it describes ANOTHER kernel and bench.py must not quote it."""
    rows, source = bench._newest_pmc_rows()
    assert source is not None
    if "stale" in source:
        assert rows == [] and bench.profiled_traffic("ilm::step_lean_kernel<true, false>") is None
        assert bench.profiled_per_wave("ilm::sphere_lights_kernel<0, false, false>", "SQ_INSTS_VALU") is None
        assert bench.step_traffic_fields(None, 5) == {"traffic": None}
        return True
    return False


def test_profiled_traffic_comes_from_the_newest_committed_summary():
    if _profile_is_stale():
        import pytest
        pytest.skip("profiles/*_pmc.csv is older than the kernel sources: bench.py reports null fractions until the round's profile is committed")
    rows, source = bench._newest_pmc_rows()
    assert source is not None and source.endswith("_pmc.csv") and len(rows) > 10
    t = bench.profiled_traffic("ilm::step_lean_kernel<true, false>", "ilm::step_lean_kernel<false, false>")
    assert t is not None and t["lanes"] > 0
    per = t["bytes"] / t["lanes"]
    # one lane per slot: the measured HBM bytes per slot-step are the algorithmic 112 B (padding lanes of partial units pull it down a little)
    assert 100.0 < per < 120.0
    f = bench.step_traffic_fields(t, 2_000_000)
    assert f["traffic"] == round(per * 2_000_000) and abs(f["traffic_per_unit"] - per) < 0.01
    assert f["traffic_profiled"]["slots_per_step"] == round(t["lanes"])
    assert bench.profiled_traffic("ilm::no_such_kernel") is None
    assert bench.step_traffic_fields(None, 5) == {"traffic": None}


def test_light_kernel_instruction_counts_are_in_the_summary():
    if _profile_is_stale():
        import pytest
        pytest.skip("stale profile (see above)")
    for k in ("ilm::sphere_lights_kernel<0, false, false>", "ilm::sphere_lights_kernel<1, false, false>"):
        v = bench.profiled_per_wave(k, "SQ_INSTS_VALU")
        assert v is not None and v["value"] > 1000


def test_contract_flags_and_defaults(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse_args()
    assert (a.gpus, a.steps, a.warmup) == (1, 200, 20)
    assert a.light_ms == 60.0 and a.light_frames >= 1 and a.cpu_seconds <= 30
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    a = bench.parse_args()
    assert (a.gpus, a.steps, a.warmup) == (8, 20, 5)
    assert bench.PARTICLE_BYTES_PER_SLOT == 112 and bench.HBM_PEAK_GBS == 8000.0


def test_the_mesh_gbuffer_scene_is_what_the_row_says():
    """next_rows.gbuffer_2p5d_1080p: 256 height volumes (top + front faces) and 64 billboards as float32 vertex rows, 2 507 triangles with the
    ground plane's two."""
    import numpy as np
    gd, top, front, bb = bench.gbuffer_meshes_scene()
    assert top.dtype == np.float32 and front.dtype == np.float32 and bb.dtype == np.float32
    assert top.shape[1] == 9 and front.shape[1] == 9 and bb.shape == (256, 12)
    assert len(top) % 3 == 0 and len(front) % 3 == 0
    assert 2 + len(top) // 3 + len(front) // 3 + 128 == 2507
    assert gd.TwoPointFiveD != 0 and gd.DistanceFieldExtentZ == 128.0


def test_optional_rows_merge_and_the_watchdog_prints_the_fallback():
    """bench.run_optional_rows (N > 1): the optional frames are merged into scaling_detail's frames when they finish; when they do not, a
    watchdog THREAD has rank 0 print the record as it stood and ends the process with status 0 (the main thread may be stuck in a
    collective, so nothing may depend on it returning)."""
    import subprocess
    import sys
    import types
    calls = []
    frames = {"cfg3": {"composited_frame_ms": 1.0}}
    old = bench.exchange_variant_rows
    bench.exchange_variant_rows = lambda v: {"store_mode": {"composited_frame_ms": v.x}}
    try:
        bench.run_optional_rows([("cfg3", types.SimpleNamespace(x=0.5))], frames, 30, 0, lambda: calls.append("fallback"), lambda: calls.append("barrier"))
    finally:
        bench.exchange_variant_rows = old
    assert frames == {"cfg3": {"composited_frame_ms": 1.0, "store_mode": {"composited_frame_ms": 0.5}}} and calls == ["barrier"]
    code = ("import sys, time, types; sys.path.insert(0, %r); import bench\n"
            "bench.exchange_variant_rows = lambda v: time.sleep(1e6)\n"
            "bench.run_optional_rows([('cfg3', types.SimpleNamespace())], {'cfg3': {}}, 1, int(sys.argv[1]), lambda: print('FALLBACK', flush=True), lambda: None)\n"
            "print('NOT REACHED')\n") % ROOT
    for rank, want in ((0, "FALLBACK\n"), (1, "")):
        p = subprocess.run([sys.executable, "-c", code, str(rank)], capture_output=True, text=True, timeout=120)
        assert p.returncode == 0 and p.stdout == want, (rank, p.stdout, p.stderr[-2000:])


def _row(value, ms, live=1000):
    blocks = [(ms * 1e-3 * 4, ms * 4 * 0.98)] * 3
    return bench.particle_row(1, live, 4, blocks, "stand-in kernel")


def test_the_one_gpu_headline_is_cfg4_whole_and_cfg2_stays_beside_it():
    """N = 1 (r06): `value`, `ms_per_step`, `roofline` and `config.workload` are the 64 M row (HBM-resident, the size the north star's particle
    target is quoted at); cfg2 (Infinity-Cache-resident) becomes `cfg2_cache_resident`.  Without the 64 M row (--no-cfg4-64m) cfg2 stays."""
    import copy
    c64 = _row(0, 1.25, live=67108864)
    c64["workload"] = "cfg4 whole on ONE GPU: 64 chunks of 1024^2 = 67108864 particles"
    roof = {"bound": "hbm", "achieved": 6400.0, "peak": bench.HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.8, "traffic": None}
    out = {"metric": "m", "value": 55000.0, "unit": "Mparticle-steps/s", "n_gpus": 1, "steps": 4, "warmup": 0, "ms_per_step": 0.02, "config": {"workload": "cfg2: ...", "live_particles_per_step_avg": 1.1e6},
           "timed_blocks": {"blocks": 3}, "roofline": roof, "cfg4_full_64m_one_gpu": c64}
    rec = bench.finalize_record(copy.deepcopy(out), 1, False, 20.0)
    assert rec["value"] == c64["mparticle_steps_per_s"] and rec["ms_per_step"] == c64["ms_per_step"] and rec["roofline"] == c64["roofline"]
    assert rec["config"]["workload"].startswith("cfg4-64M on one GPU: 64 chunks of 1024^2") and rec["config"]["particles_per_gpu"] == 67108864
    assert rec["cfg2_cache_resident"]["mparticle_steps_per_s"] == 55000.0 and rec["cfg2_cache_resident"]["roofline"] == roof
    assert rec["summary"]["particles_cfg2"]["mparticle_steps_per_s"] == 55000.0 and rec["summary"]["particles_cfg4_64m_one_gpu"]["ms_per_step"] == c64["ms_per_step"]
    # the fraction is stated from both clocks: HIP events (the contract's) and ms_per_step
    assert 0 < rec["roofline"]["frac_from_ms_per_step"] <= rec["roofline"]["frac"]
    del out["cfg4_full_64m_one_gpu"]
    rec = bench.finalize_record(copy.deepcopy(out), 1, False, 20.0)
    assert rec["value"] == 55000.0 and "cfg2_cache_resident" not in rec and rec["config"]["workload"] == "cfg2: ..."


def test_the_n_gpu_headline_takes_the_live_count_row_when_it_was_measured():
    """N > 1 (r06): BASELINE config 4 is "per-chunk update + RCCL all-gather".  The rows with collectives run last under the watchdog; once
    measured, `value` is the step WITH ilm_group_live_counts and the communication-free figure stays as value_without_collectives; when they
    did not finish (or failed) the headline says that it has no collective."""
    import copy
    share, wl = _row(0, 0.20, live=8388608), _row(0, 0.21, live=8388608)
    wl["live_count_calls_per_block"] = 1.0
    roof = {"bound": "hbm", "achieved": 1.0, "peak": bench.HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.1, "traffic": None}
    out = {"metric": "m", "value": 1.0, "unit": "Mparticle-steps/s", "n_gpus": 2, "steps": 4, "warmup": 0, "ms_per_step": 0.02, "config": {"workload": "cfg2: ..."},
           "timed_blocks": {"blocks": 3}, "roofline": roof, "cfg4_share_8m_particles": share}
    rec = bench.finalize_record(copy.deepcopy(out), 2, False, 20.0, {"with_live_counts": wl, "with_position_all_gather": {"error": "x"}})
    assert rec["value"] == wl["mparticle_steps_per_s"] < rec["value_without_collectives"] == share["mparticle_steps_per_s"]
    assert rec["ms_per_step"] == wl["ms_per_step"] and "ilm_group_live_counts" in rec["config"]["collective_in_the_timed_steps"]
    for rows in (None, {}, {"with_live_counts": {"error": "RuntimeError: x"}}):
        rec = bench.finalize_record(copy.deepcopy(out), 2, False, 20.0, rows)
        assert rec["value"] == share["mparticle_steps_per_s"] == rec["value_without_collectives"] and rec["config"]["collective_in_the_timed_steps"].startswith("none")
    sd = bench.scaling_detail(2, share, share, None, None, [1.0, 1.0], {}, collective_rows={"with_live_counts": wl, "with_position_all_gather": {"error": "x"}})
    assert sd["particles"]["with_live_counts"]["vs_without_collectives"] == round(wl["mparticle_steps_per_s"] / share["mparticle_steps_per_s"], 4)
    assert sd["particles"]["with_position_all_gather"] == {"error": "x"}


def test_the_collective_log_folds_repeats_and_keeps_the_order():
    log = bench.CollectiveLog()
    log.phase("a")
    for _ in range(3):
        log.add("ilm_group_host_all_gather", "ncclAllGather", 8)
    log.add("ilm_group_live_counts", "ncclAllGather", 32)
    log.phase("b")
    log.add("ilm_group_host_all_gather", "ncclAllGather", 8)
    t = log.table()
    assert len(t) == 4 and t[1].endswith("| x3") and "| a |" in t[1] and "ilm_group_live_counts" in t[2] and "| b |" in t[3] and t[3].endswith("| x1")


def test_the_particle_rows_with_collectives_run_first_under_the_watchdog():
    import types
    order = []
    old_p, old_x = bench.particle_collective_rows, bench.exchange_variant_rows
    bench.particle_collective_rows = lambda v: (order.append("particles"), {"with_live_counts": {"mparticle_steps_per_s": 1.0}})[1]
    bench.exchange_variant_rows = lambda v: (order.append("frames"), {"store_mode": {}})[1]
    rows, frames = {}, {"cfg3": {}}
    try:
        bench.run_optional_rows([("cfg3", types.SimpleNamespace())], frames, 30, 0, lambda: order.append("fallback"), lambda: order.append("barrier"),
                                particle_rows=(types.SimpleNamespace(), rows))
    finally:
        bench.particle_collective_rows, bench.exchange_variant_rows = old_p, old_x
    assert order == ["particles", "frames", "barrier"] and rows == {"with_live_counts": {"mparticle_steps_per_s": 1.0}} and "store_mode" in frames["cfg3"]
