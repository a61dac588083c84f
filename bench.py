#!/usr/bin/env python3
"""bench.py -- throughput of the two hot paths on MI355X (contract: see the task's bench section).

    python bench.py --gpus N --steps K --warmup W

A "step" is one ParticleSystem.Update over the whole synthetic particle system (BASELINE.json configs[1]:
1 M particles in 16 chunks of 256^2, Spawner + Gravity(4 attractors) + Noise + UpdatePositions, fp32).
`value` = whole-job Mparticle-steps/s with all inputs resident in HBM.  With N > 1 every rank owns the same
number of chunks (chunks never interact, ParticleSystem.cs:743-745): no data-path collective, weak scaling.
The second hot path (sphere-light SDF cone trace) is measured after the timed particle region and reported
in the same JSON line under "lighting" (1080p/64 lights and 4K/256 lights fp16; screen strips + RCCL
all-gather of the lightmap when N > 1).

The host side is the C++ mirror of the reference's ParticleSystem / LightingRenderer (illuminant_amd/host),
calling the kernels only through the C ABI of include/illuminant_hip.h.  The CPU oracle is used solely for
the "cpu_baseline" leg (rank 0, N == 1).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
PARTICLE_BYTES_PER_SLOT = 112   # 48 B read (pos+life, vel+cat, attributes) + 64 B written (pos, vel, render colour, render data)
# fp32 VALU issue: 256 CUs x 4 SIMD-32, one wave64 instruction per 2 cycles (MI355X_MICROARCH.md "Wave scheduling") at 2.4 GHz;
# tools/ubench/valu (independent v_fma_f32 chains, 8 waves per SIMD) sustains 858 G/s -- the clock settles near 1.9 GHz under that load
# (profiles/r01/r01_valu_issue_calibration.md)
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2.0 / 1e9      # G wave-instructions / s
VALU_ISSUE_CALIBRATED = 858.0
SDF_SAMPLE_BYTES = 32           # one sampleDistanceFieldEx = 4 bilinear taps x 8 B RGBA16
# Work-based bound of the cone trace: VALU instructions one coneTraceStep + sampleDistanceFieldEx needs per SAMPLE.  The yardstick is
# r02's loop (60; VERDICT r02 fixed it so that the fraction is comparable across rounds); the loop the shipped kernel runs is listed
# beside it (lighting.hip cone_trace_loop<FAST>: 56 for fp16 fields, 52 for unorm16 ones since the cell array of r03; 12 fewer in the
# iterations whose visibility division is skipped).
TRACE_INSTRUCTIONS_PER_SAMPLE = 60          # r02's yardstick, kept so that the fraction stays comparable across rounds
TRACE_LOOP_INSTRUCTIONS = {"fp16": 56, "unorm16": 52}      # with the visibility division taken; 44 / 40 when the wave skips it (docs/experiments.md 3.2)
# The loop the shipped kernel runs, weighted by how often a wave's iteration skips the division (82 % of them on cfg5, measured when the
# skip went in, docs/experiments.md 3.2): 0.82 x 44 + 0.18 x 56 and 0.82 x 40 + 0.18 x 52.  This is the PRIMARY work-based figure since r04.
TRACE_LOOP_WEIGHTED = {"fp16": 46.0, "unorm16": 42.0}
INFINITY_CACHE_MB = 256


def kernel_sources_sha256():
    """sha256 over the sources a PMC profile is a profile OF, in name order: every csrc/*.hip that contains device code (`__global__`:
    the kernels, and api.hip, which also plans their launches), every csrc/*.hpp and the Makefile.  group.hip is host code over the C ABI
    (no kernel, no launch of its own): an edit there does not change what a counter counted (r05)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "illuminant_amd", "csrc")
    hips = [f for f in glob.glob(os.path.join(d, "*.hip")) if b"__global__" in open(f, "rb").read()]
    for f in sorted(hips + glob.glob(os.path.join(d, "*.hpp")) + [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()


def _newest_pmc_rows():
    """Rows of the NEWEST committed rocprofv3 PMC summary (profiles/*_pmc.csv, written by tools/profile_bench.sh on the SAME bench
    command).  Only the newest file counts: a kernel that is missing from it (renamed, rewritten since) has no profile, and a figure
    from an older round's kernel would be a wrong figure.  Since r04 the summary carries the sha256 of the kernel sources it was taken
    from (<tag>_pmc.meta.json): when the tree's sources differ -- a kernel was edited after the profile -- the profile is not used at all
    (the per-wave instruction counts and traffic figures derived from it would describe another kernel) and the fractions read null."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.csv")))
    if not files:
        return [], None
    meta = files[-1][:-4] + ".meta.json"
    if os.path.exists(meta):
        try:
            if json.load(open(meta)).get("kernel_sources_sha256") != kernel_sources_sha256():
                return [], os.path.basename(files[-1]) + " (stale: the kernel sources changed since)"
        except (OSError, ValueError):
            return [], None
    return list(csv.DictReader(open(files[-1]))), os.path.basename(files[-1])


def profiled_traffic(*kernel_prefixes):
    """HBM bytes per launch of a kernel -- or, with several prefixes, of the launches that make up one step (their sum) -- from the
    committed PMC summary: FETCH_SIZE x 2 + WRITE_SIZE, both in KB.  The x2 is the gfx950 FETCH_SIZE correction of
    MI355X_MICROARCH.md ("reports exactly half of the bytes of a wide coalesced streaming read"), confirmed here on the step kernel's
    known 48 B/slot read.  Counters cannot be collected from inside the timed run (rocprofv3 wraps the process), so the figure is
    carried over from the profile; None when a kernel is absent from it."""
    rows, source = _newest_pmc_rows()
    total, lanes = 0.0, 0.0
    for prefix in kernel_prefixes:
        hit = [r for r in rows if r["kernel"].startswith(prefix) and r.get("FETCH_SIZE") and r.get("WRITE_SIZE")]
        if not hit:
            return None
        total += (float(hit[0]["FETCH_SIZE"]) * 2.0 + float(hit[0]["WRITE_SIZE"])) * 1024.0
        lanes += float(hit[0].get("SQ_WAVES") or 0.0) * 64.0
    return {"bytes": total, "lanes": lanes, "source": source}


def step_traffic_fields(t, units):
    """The step kernels run one lane per slot, so the profile's bytes / (SQ_WAVES x 64) is the measured HBM traffic per slot-step of
    the profiled launches; `traffic` = that figure x the units of THIS run's launch (the profiled run has fewer live slots than the
    timed blocks end with), the profile's own numbers beside it."""
    if not t or not t["lanes"]:
        return {"traffic": None}
    per = t["bytes"] / t["lanes"]
    return {"traffic": round(per * units), "traffic_per_unit": round(per, 2),
            "traffic_profiled": {"bytes_per_step": round(t["bytes"]), "slots_per_step": round(t["lanes"])},
            "traffic_source": "profiles/%s: (FETCH_SIZE x 2 + WRITE_SIZE) KB / (SQ_WAVES x 64 slots) of the step's launches, x units_per_launch" % t["source"]}


def profiled_per_wave(kernel_prefix, column):
    """Average of a per-dispatch SQ counter divided by SQ_WAVES from the committed PMC summary (same source and caveat as
    profiled_traffic); None when absent."""
    rows, source = _newest_pmc_rows()
    for row in rows:
        if row["kernel"].startswith(kernel_prefix) and row.get(column) and row.get("SQ_WAVES") and float(row["SQ_WAVES"]) > 0:
            return {"value": float(row[column]) / float(row["SQ_WAVES"]), "source": source}
    return None


def calibrate_hbm_copy(native, device, ctx, gib=1, reps=6):
    """What THIS box's HBM gives a plain copy: `gib` GiB device-to-device on the context's stream (hipMemcpyAsync: the runtime's copy kernel), bytes
    read + bytes written over the time between two HIP events around `reps` copies.  No kernel of this library is involved; buffers are far
    larger than the 256 MiB Infinity Cache."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    n = int(gib) << 30
    a, b = C.c_void_p(), C.c_void_p()
    ctx.Sync()
    cal = native.Context(device)          # a context of its own: handing out a context's stream changes how that context steps (one stream)
    stream = cal.stream()
    if hip.hipMalloc(C.byref(a), C.c_size_t(n)) != 0 or hip.hipMalloc(C.byref(b), C.c_size_t(n)) != 0:
        cal.close()
        raise RuntimeError("hipMalloc of the calibration buffers failed")
    try:
        hip.hipMemsetAsync(a, 1, C.c_size_t(n), C.c_void_p(stream)); hip.hipMemsetAsync(b, 2, C.c_size_t(n), C.c_void_p(stream))
        for _ in range(2):
            hip.hipMemcpyAsync(b, a, C.c_size_t(n), 3, C.c_void_p(stream))          # 3 = hipMemcpyDeviceToDevice
        cal.sync()
        cal.timer_start()
        for _ in range(reps):
            hip.hipMemcpyAsync(b, a, C.c_size_t(n), 3, C.c_void_p(stream))
        ms = cal.timer_stop() / reps
    finally:
        cal.sync()
        hip.hipFree(a); hip.hipFree(b)
        cal.close()
    runtime_rate = 2.0 * n / (ms * 1e-3) / 1e9
    # a float4 copy KERNEL (plain / non-temporal stores / non-temporal loads and stores), from a library of its own that is not the product
    # (csrc/calib.hip -> lib/libilluminant_calib.so); the runtime's copy is slower than a kernel's and only the fallback
    rates, lib_path = None, os.path.join(os.path.dirname(os.path.abspath(__file__)), "illuminant_amd", "lib", "libilluminant_calib.so")
    if os.path.exists(lib_path):
        cl = C.CDLL(lib_path)
        cl.ilm_calib_copy_rates.argtypes = [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
        out3 = (C.c_double * 3)()
        if cl.ilm_calib_copy_rates(int(device), C.c_size_t(n), int(reps), out3) == 0:
            rates = [float(out3[i]) for i in range(3)]
    best = max(rates) if rates else runtime_rate
    return {"copy_gb_per_s": round(best, 1), "frac_of_spec": round(best / HBM_PEAK_GBS, 4), "gib_per_copy": gib, "copies": reps,
            "copy_kernel_gb_per_s": ({"plain": round(rates[0], 1), "nt_stores": round(rates[1], 1), "nt_loads_and_stores": round(rates[2], 1)} if rates else None),
            "runtime_d2d_copy_gb_per_s": round(runtime_rate, 1),
            "is": "bytes read + bytes written of a %d GiB device-to-device copy over HIP events around %d copies, this box, this run: the best of a float4 copy kernel's "
                  "three forms (csrc/calib.hip, not the product library), hipMemcpyAsync beside it; MI355X_MICROARCH.md quotes 6.29 TB/s for such a kernel "
                  "against the 8 TB/s spec" % (gib, reps)}


def light_launch_waves(native_ctx):
    """Waves of the context's last light-pass launch, from the library itself (ilm_debug_last_light_launch: workgroups of four waves, exit-only
    workgroups of partial tile groups and the members of split tiles included -- what SQ_WAVES counts).  r04 re-derived the grid here and
    was wrong for strips (tile groups of 4 x 4, tapered split: ADVICE r04)."""
    return native_ctx.last_light_launch()[0] * 4


class CollectiveLog:
    """The ordered list of collectives a run issues: every call of a group entry point that ends in RCCL traffic is
    noted here BEFORE it is issued -- phase, entry point, what it puts on the wire, bytes per rank -- with consecutive identical calls
    folded into one line with a count.  With N > 1 every NEW line also goes to stderr at once (flushed), so that a hang on hardware is
    located by the last line of the log; `bench.py --dry-collectives` walks the whole N > 1 branch once at minimal repetition and prints
    the list (profiles/r06_collective_schedule.txt).  Host-side bookkeeping only: nothing here touches the data path."""

    def __init__(self):
        self.lines, self.echo, self.rank, self._phase = [], False, 0, "start-up"

    def phase(self, name):
        self._phase = name

    def add(self, entry_point, wire, bytes_per_rank):
        key = (self._phase, entry_point, wire, int(bytes_per_rank))
        if self.lines and self.lines[-1][0] == key:
            self.lines[-1][1] += 1
            return
        self.lines.append([key, 1])
        if self.echo:
            print("[collective %4d r%d] %s | %s | %s | %d B/rank" % (len(self.lines), self.rank, key[0], key[1], key[2], key[3]), file=sys.stderr, flush=True)

    def table(self):
        out = ["# seq | phase | entry point | on the wire | bytes per rank | consecutive calls"]
        for i, (key, n) in enumerate(self.lines):
            out.append("%4d | %s | %s | %s | %d | x%d" % (i + 1, key[0], key[1], key[2], key[3], n))
        return out


COLLECTIVES = CollectiveLog()


def log_group_collectives(native, world):
    """Wrap the bindings' methods that issue collectives so that every call is noted in COLLECTIVES first (see CollectiveLog)."""
    def wrap(cls, name, describe):
        orig = getattr(cls, name)

        def logged(self, *a, **k):
            d = describe(self, *a, **k)
            if d is not None:
                COLLECTIVES.add(*d)
            return orig(self, *a, **k)
        setattr(cls, name, logged)
    G, L = native.Group, native.GroupLightmap

    def lightmap_exchange(glm, mode):
        base = mode & ~native.GATHER_ASYNC
        row = glm.width * {0: 16, 1: 8, 2: 4}[glm.format]
        asy = " on the exchange stream" if (mode & native.GATHER_ASYNC) else ""
        if base == native.GATHER_STORE:
            return ("ilm_group_lightmap_gather(STORE)", "fence: ncclAllGather", 8)
        if base == native.GATHER_RCCL:
            equal = all(glm.strips[r] == (min(r * glm.slot_rows, glm.height), min((r + 1) * glm.slot_rows, glm.height)) for r in range(world))
            if equal:
                return ("ilm_group_lightmap_gather(RCCL)", "ncclAllGather in place" + asy, glm.slot_rows * row)
            b, e = glm.strips[glm.group.first_rank]
            return ("ilm_group_lightmap_gather(RCCL)", "ncclGroup of %d ncclSend + %d ncclRecv (strips)%s" % (world - 1, world - 1, asy), (e - b) * row)
        return None
    wrap(G, "host_all_gather", lambda self, b: ("ilm_group_host_all_gather", "hipMemcpy H2D + ncclAllGather + hipMemcpy D2H, stream sync", len(b) // self.n_local))
    wrap(G, "live_counts", lambda self, systems, total, saturate16=False: ("ilm_group_live_counts", "stream sync + ncclAllGather of the per-chunk counts", 4 * ((total + world - 1) // world)))
    wrap(G, "gather_chunks", lambda self, src, dst, total, first=0, count=4, gather=2:
         ("ilm_group_gather_chunks(components %d..%d)" % (first, first + count - 1), "ncclGroup of %d x (ncclSend + ncclRecv), one per owned chunk and peer" % (((total + world - 1) // world) * (world - 1)),
          ((total + world - 1) // world) * count * int(dst[0].device_ptr(0, 0)[1]) * 4))
    wrap(G, "render_sphere_lights", lambda self, lights, env, df, gb, sdfs, amb, glm, gather=1, want_stats=False: lightmap_exchange(glm, gather) and
         ("ilm_group_render_sphere_lights -> " + lightmap_exchange(glm, gather)[0], lightmap_exchange(glm, gather)[1], lightmap_exchange(glm, gather)[2]))
    wrap(L, "gather", lambda self, mode: lightmap_exchange(self, mode))
    wrap(L, "set_strips", lambda self, strips=None: ("ilm_group_lightmap_set_strips", "ncclAllGather of the table's hash (host payload)", 8))
    wrap(L, "store_mode", lambda self, enable=True: ("ilm_group_lightmap_store_mode(%d)" % (1 if enable else 0),
                                                      "arming: ncclAllGather of 64 B IPC handles, of 8 B verdicts, 3 x 8 B of the stamp proof (first arming of a lightmap); disarming: one 8 B ncclAllGather", 64))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--chunks", type=int, default=16, help="uploaded 256^2 chunks per GPU (16 = 1 M particles)")
    ap.add_argument("--chunk-size", type=int, default=256)
    ap.add_argument("--light-frames", type=int, default=5)
    ap.add_argument("--light-ms", type=float, default=60.0, help="GPU time a timed block of lit frames fills at least (0: exactly --light-frames)")
    ap.add_argument("--sustain-s", type=float, default=5.5, help="GPU time the timed block of cfg5 frames fills at least (the longest GPU phase of the run: "
                                                                 "a 5 s device-utilisation sampler sees it); 0: as --light-ms")
    ap.add_argument("--no-lighting", action="store_true")
    ap.add_argument("--no-cfg4", action="store_true", help="skip the 8 M-particle (cfg4 per-GPU share) measurement")
    ap.add_argument("--optional-rows-timeout", type=int, default=300, help="N > 1: seconds the optional frames of scaling_detail (pipelined exchange, store mode) may take "
                    "before a watchdog prints the record without them and ends the job with status 0")
    ap.add_argument("--no-store-mode-row", action="store_true", help="N > 1: skip the store-mode frames (IPC-mapped buffers of the other ranks) of scaling_detail")
    ap.add_argument("--no-cfg4-64m", action="store_true", help="skip cfg4 whole on one GPU (64 chunks of 1024^2 = 67 M particles, 5.4 GB; N > 1: measured by rank 0 alone)")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the SURVEY 8f measurements (read-back, particle lights, resolve)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--dry-collectives", action="store_true", help="N > 1 (or ILM_BENCH_FORCE_DIST): walk the whole branch once at minimal repetition and print the ordered list of "
                    "collectives with their byte counts instead of the record (profiles/r06_collective_schedule.txt)")
    ap.add_argument("--no-particle-collective-rows", action="store_true", help="N > 1: skip the particle rows with collectives (live counts, Pos+Life all-gather)")
    ap.add_argument("--blocks", type=int, default=0, help="timed K-step blocks (0: as many as fill ~50 ms of GPU time, 3..15); "
                                                          "the headline is the median block, min / max are reported beside it")
    args = ap.parse_args()
    if args.dry_collectives:
        args.steps, args.warmup, args.blocks, args.light_frames, args.light_ms, args.sustain_s = 5, 0, 1, 1, 0.0, 0.0
        args.no_cpu_baseline = args.no_next_rows = True
    return args


def exchange_unique_id(native, rank, world):
    """Rank 0 creates the RCCL id (ilm_group_unique_id) and leaves it in a file only this launch can name: the ranks of one node share
    their launcher (torchrun agent) as parent, so <parent pid, MASTER_PORT, world> identifies the job.  One node only, as the contract."""
    import atexit
    import tempfile
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    path = os.path.join(base, "ilm_bench_%d_%s_%d.id" % (os.getppid(), os.environ.get("MASTER_PORT", "0"), world))
    if rank == 0:
        uid = native.Group.unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(path + ".tmp", path)
        atexit.register(lambda: os.path.exists(path) and os.remove(path))
        return uid
    deadline = time.time() + 180.0
    while time.time() < deadline:
        try:
            with open(path, "rb") as f:
                data = f.read()
            if len(data) == 128:
                return data
        except FileNotFoundError:
            pass
        time.sleep(0.01)
    raise SystemExit("bench.py: rank %d waited 180 s for rank 0's RCCL id (%s)" % (rank, path))


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks ourselves, one process per GPU (LOCAL_RANK = i, rendezvous on
    127.0.0.1), exactly the command the driver uses.  The ranks fail loudly when the box has fewer than N GPUs."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


# ---- what the ranks of one job say to each other outside the data path -----------------------------------------------

class Ranks:
    """Barrier, max / sum / list over the ranks of the job: small host payloads through `group.host_all_gather` (ilm_group_host_all_gather,
    one RCCL all-gather on the job's only communicator; bytes in, a list of `world` bytes objects out).  `group` None = one rank.  `sync`
    drains this rank's device (hipStreamSynchronize of the context stream).  Everything the N > 1 record needs from the other ranks goes
    through this class, so tests/_gloo_worker.py can run the record assembly at world 2 on CPU with a stand-in group."""

    def __init__(self, group, rank, world, sync):
        self.group, self.rank, self.world, self.sync = group, rank, world, sync

    def barrier(self):
        """Device idle on every rank, then (N > 1) a collective that no rank leaves before every rank has entered it."""
        self.sync()
        if self.group is not None:
            self.group.host_all_gather(b"\0" * 8)

    def doubles(self, x):
        """x of every rank, in rank order (identical list on every rank)."""
        import struct
        if self.group is None:
            return [float(x)]
        return [struct.unpack("<d", b[:8])[0] for b in self.group.host_all_gather(struct.pack("<d", float(x)))]

    def max(self, x):
        return max(self.doubles(x))

    def sum(self, x):
        return sum(self.doubles(x))

    def solo(self):
        """The same interface for a phase ONE rank runs alone while the others wait at the next barrier."""
        return Ranks(None, self.rank, 1, self.sync)


def time_blocks(ctx, ranks, one_step, k, n_blocks):
    """n_blocks timed blocks of EXACTLY k calls of one_step() each, every block between two barriers: [(wall seconds, max over ranks;
    this rank's device milliseconds, HIP events on the context stream)], sorted by wall time."""
    out = []
    for _ in range(n_blocks):
        ranks.barrier()
        ctx.TimerStart()
        t0 = time.perf_counter()
        for _ in range(k):
            one_step()
        g = ctx.TimerStop()
        ranks.barrier()
        out.append((ranks.max(time.perf_counter() - t0), g))
    out.sort()
    return out


def particle_row(world_units, live_per_rank, k, blocks, kernel, traffic=None):
    """A particle-step row from time_blocks(): median block; `world_units` = how many ranks' units the wall time covers."""
    w, g = blocks[len(blocks) // 2]
    gbs = live_per_rank * PARTICLE_BYTES_PER_SLOT / (g / k * 1e-3) / 1e9
    gbs_wall = live_per_rank * PARTICLE_BYTES_PER_SLOT / (w / k) / 1e9      # the same bytes over ms_per_step (wall clock, max over ranks, barriers included)
    return {"mparticle_steps_per_s": round(world_units * live_per_rank * k / w / 1e6, 1), "ms_per_step": round(w / k * 1e3, 5), "steps": k,
            "particles_per_gpu": live_per_rank,
            "timed_blocks": {"blocks": len(blocks), "steps_per_block": k, "headline": "median block", "ms_per_step_min": round(blocks[0][0] / k * 1e3, 5),
                             "ms_per_step_max": round(blocks[-1][0] / k * 1e3, 5)},
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                         "frac_from_ms_per_step": round(gbs_wall / HBM_PEAK_GBS, 4),
                         **step_traffic_fields(traffic, live_per_rank), "kernel": kernel, "bytes_per_unit": PARTICLE_BYTES_PER_SLOT,
                         "units_per_launch": live_per_rank, "launch_ms": round(g / k, 5)}}


def balance_strips(glm, ranks, height, packed_lights, time_strip, rounds=2):
    """The strips of an N-rank lit frame: whole 16-row bands cut where the lights' raster footprints say the work is equal
    (sharding.balanced_row_strips; every rank computes the same table from the same packed lights), then re-cut `rounds` times from what
    the strips COST: every rank times its own strip (time_strip(begin, end) -> ms), the times are all-gathered, every rank computes the
    same new table (sharding.rebalance_row_strips) -- the footprint model cannot see the obstacle field.  glm.set_strips is a collective
    that checks that the ranks agree (ilm_group_lightmap_set_strips).  Returns the history [{strips, how, ms}]."""
    from illuminant_amd import sharding
    glm.set_strips(sharding.balanced_row_strips(height, ranks.world, packed_lights))
    history = [{"strips": [list(s_) for s_ in glm.strips], "how": "footprint model"}]
    for round_ in range(rounds):
        b, e = glm.strips[ranks.rank]
        times = ranks.doubles(time_strip(b, e))
        history[-1]["ms"] = [round(t, 4) for t in times]
        glm.set_strips(sharding.rebalance_row_strips(glm.strips, times, height))
        history.append({"strips": [list(s_) for s_ in glm.strips], "how": "measured, round %d" % (round_ + 1)})
    return history


def scaling_detail(world, share, share_solo, full_64m_solo, strong_64m, per_rank_rates, frames, collective_rows=None):
    """The N > 1 line describes itself (VERDICT r04 #1c): every ratio names its denominator, and the denominators are measured by rank 0
    ALONE in the same job (the other ranks wait at a barrier), so that no ratio divides across workloads or across boxes.
      share          the headline row: 8 chunks of 1024^2 per rank, all ranks stepping together (aggregate over the job)
      share_solo     the same 8 chunks stepped by rank 0 alone                               -> weak_vs_share
      full_64m_solo  64 chunks of 1024^2 = cfg4 whole on rank 0's GPU alone                   -> strong_vs_one_gpu_64m
      strong_64m     cfg4 whole sharded over the job: 64 / world chunks per rank (chunk c on rank c mod world), all ranks together
      frames         per lit frame: this job's strips (per-rank ms, max, sum), the exchange alone, the composited frame, and the whole
                     frame on rank 0's GPU alone."""
    d = {"ranks": world,
         "particles": {
             "aggregate_mparticle_steps_per_s": share["mparticle_steps_per_s"],
             "per_rank_mparticle_steps_per_s": [round(r, 1) for r in per_rank_rates],
             "share_one_gpu_alone": {"workload": "8 chunks of 1024^2 on rank 0's GPU, the other ranks idle", "mparticle_steps_per_s": share_solo["mparticle_steps_per_s"],
                                     "ms_per_step": share_solo["ms_per_step"]},
             "weak_vs_share": round(share["mparticle_steps_per_s"] / (world * share_solo["mparticle_steps_per_s"]), 4),
             "weak_vs_share_is": "aggregate of the job / (ranks x share_one_gpu_alone): 1.0 = no rank slows another"}}
    if full_64m_solo and strong_64m:
        d["particles"].update({
            "one_gpu_64m": {"workload": full_64m_solo["workload"], "mparticle_steps_per_s": full_64m_solo["mparticle_steps_per_s"],
                            "ms_per_step": full_64m_solo["ms_per_step"], "frac_of_hbm_peak": full_64m_solo["roofline"]["frac"]},
            "sharded_64m": {"workload": strong_64m["workload"], "mparticle_steps_per_s": strong_64m["mparticle_steps_per_s"], "ms_per_step": strong_64m["ms_per_step"]},
            "strong_vs_one_gpu_64m": round(strong_64m["mparticle_steps_per_s"] / full_64m_solo["mparticle_steps_per_s"], 4),
            "strong_vs_one_gpu_64m_is": "64 M particles stepped by the job (64 / ranks chunks per rank) / the same 64 M on rank 0's GPU alone; the north star asks >= 6.4 at 8 ranks"})
    if collective_rows:
        # BASELINE config 4 / SURVEY 8e row P: the step WITH the collectives the path has, beside the communication-free figure above
        for key in ("with_live_counts", "with_position_all_gather"):
            if key in collective_rows:
                row = collective_rows[key]
                if "mparticle_steps_per_s" in row:
                    row = dict(row, vs_without_collectives=round(row["mparticle_steps_per_s"] / share["mparticle_steps_per_s"], 4))
                d["particles"][key] = row
    d["frames"] = frames
    return d


def finalize_record(out, world, forced_dist, step_us_cfg2, collective_rows=None):
    """The last stage of the record: with N > 1 the whole-job number is taken where the north star puts the scaling target -- 64 M particles
    on 8 GPUs = cfg4's per-GPU share (8 chunks of 1024^2 per rank, HBM-resident) -- and N x cfg2 (cache-resident on every GPU) becomes the
    secondary row; then the summary and the key order (the driver keeps the parsed keys and the LAST ~2000 characters of the line: bulky
    rows first; the rooflines, the CPU baseline, the scaling block and a compact summary of both hot paths last).  Pure: dict in, dict out."""
    c4h = out.get("cfg4_share_8m_particles")
    multi = world > 1 or forced_dist
    if multi and c4h:
        out["cfg2_weak_row"] = {"mparticle_steps_per_s": out["value"], "ms_per_step": out["ms_per_step"], "roofline": out["roofline"],
                                "workload": out["config"]["workload"], "timed_blocks": out.pop("timed_blocks")}
        out["value"] = c4h["mparticle_steps_per_s"]
        out["ms_per_step"] = c4h["ms_per_step"]
        out["roofline"] = dict(c4h["roofline"], resident="hbm (0.67 GB of particle state per GPU > %d MiB Infinity Cache)" % INFINITY_CACHE_MB)
        out["timed_blocks"] = c4h["timed_blocks"]
        out["config"]["workload"] = "cfg4: %d particles per GPU in 8 chunks of 1024^2 (64 M on 8 GPUs), Gravity(4 attractors)+Noise+UpdatePositions" % c4h["particles_per_gpu"]
        out["config"]["particles_per_gpu"] = c4h["particles_per_gpu"]
        out["config"]["parallelism"] = ("particles: chunk c on rank c mod %d; lit frame: cost-balanced row strips of whole 16-row bands "
                                        "(balanced_row_strips, then re-cut twice from the ranks' measured strip times: rebalance_row_strips), range exchange over RCCL send/recv" % world)
        # BASELINE config 4 is "per-chunk update + RCCL all-gather": when the rows with the collectives were measured (they run last, under
        # the watchdog), the headline is the step WITH the live-count all-gather of every liveness interval (ParticleLiveness.cs:14,
        # ParticleEngine.cs:282-386); the communication-free figure stays beside it.  Otherwise the headline says that it has no collective.
        wl = (collective_rows or {}).get("with_live_counts") or {}
        out["value_without_collectives"] = c4h["mparticle_steps_per_s"]
        if "mparticle_steps_per_s" in wl:
            out["value"] = wl["mparticle_steps_per_s"]
            out["ms_per_step"] = wl["ms_per_step"]
            out["timed_blocks"] = wl["timed_blocks"]
            out["roofline"] = dict(wl["roofline"], resident=out["roofline"]["resident"])
            out["config"]["collective_in_the_timed_steps"] = ("ilm_group_live_counts (one small RCCL all-gather of the per-chunk counts) after every counting step: "
                                                              "%s call(s) in the %d timed steps of a block" % (wl.get("live_count_calls_per_block"), wl["steps"]))
        else:
            out["config"]["collective_in_the_timed_steps"] = "none: the rows with collectives (scaling_detail.particles) did not finish; `value` is the communication-free step"
    c64h = out.get("cfg4_full_64m_one_gpu")
    if not multi and c64h:
        # N = 1 (VERDICT r05 #1c): the north star's particle target is defined at 64 M particles (ParticleSystem.cs:49: 64 chunks), the
        # size at which the state (5.4 GB) is HBM-resident -- that row is `value`; cfg2 (84 MB, Infinity-Cache-resident) stays beside it.
        out["cfg2_cache_resident"] = {"mparticle_steps_per_s": out["value"], "ms_per_step": out["ms_per_step"], "roofline": out["roofline"],
                                      "workload": out["config"]["workload"], "timed_blocks": out.pop("timed_blocks"),
                                      "live_particles_per_step_avg": out["config"].pop("live_particles_per_step_avg", None),
                                      "spawned_in_run": out["config"].pop("spawned_per_gpu_in_run", None), "chunks_at_end": out["config"].pop("chunks_at_end", None)}
        out["value"] = c64h["mparticle_steps_per_s"]
        out["ms_per_step"] = c64h["ms_per_step"]
        out["roofline"] = c64h["roofline"]
        out["timed_blocks"] = c64h["timed_blocks"]
        out["config"]["workload"] = "cfg4-64M on one GPU: " + c64h["workload"].split(": ", 1)[1]
        out["config"]["particles_per_gpu"] = c64h["particles_per_gpu"]
        out["config"]["parallelism"] = "one GPU: 64 chunks in one launch pair per step (the chunk range halved over the context's two streams)"
    tail_keys = ["cpu_baseline", "roofline", "roofline_hbm_resident", "cfg4_full_64m_one_gpu", "roofline_lighting_cfg3", "roofline_lighting", "lit_mpixels_per_s", "scaling_detail", "summary"]
    c4 = out.get("cfg4_share_8m_particles")
    if c4:
        out["roofline_hbm_resident"] = dict(c4["roofline"], workload="cfg4 per-GPU share: 8 chunks of 1024^2 = 8.4 M particles, 0.67 GB of state (> Infinity Cache)",
                                            ms_per_step=c4["ms_per_step"], mparticle_steps_per_s=c4["mparticle_steps_per_s"])
    cfg2_row = out.get("cfg2_weak_row") or out.get("cfg2_cache_resident") or {"mparticle_steps_per_s": out["value"], "roofline": out["roofline"]}
    summary = {"particles_cfg2": {"mparticle_steps_per_s": cfg2_row["mparticle_steps_per_s"], "us_per_step": round(step_us_cfg2, 2),
                                  "frac_of_hbm_peak": cfg2_row["roofline"]["frac"], "resident": "infinity-cache"}}
    if c4:
        summary["particles_cfg4_share"] = {"mparticle_steps_per_s": c4["mparticle_steps_per_s"], "us_per_step": round(c4["roofline"]["launch_ms"] * 1e3, 1),
                                           "frac_of_hbm_peak": c4["roofline"]["frac"], "resident": "hbm"}
    c64 = out.get("cfg4_full_64m_one_gpu")
    if c64:
        summary["particles_cfg4_64m_one_gpu"] = {"mparticle_steps_per_s": c64["mparticle_steps_per_s"], "ms_per_step": c64["ms_per_step"],
                                                 "frac_of_hbm_peak": c64["roofline"]["frac"], "resident": "hbm"}
    for key, short in (("cfg3_1080p_64_lights_unorm16", "lighting_cfg3"), ("cfg5_4k_256_lights_fp16", "lighting_cfg5")):
        row = out.get("lighting", {}).get(key)
        if row:
            summary[short] = {"ms_per_frame": row["roofline"]["launch_ms"], "timed_frames": row["timed_frames"], "lit_mpixels_per_s": row["lit_mpixels_per_s"],
                              "gbuffer": "bound", "without_gbuffer_ms": row["without_gbuffer_ms"], "verified_counts": row.get("verified_counts"),
                              "two_frames_in_flight_ms": (row.get("two_frames_in_flight") or {}).get("ms_per_frame"),
                              "valu_issue_frac": row["roofline"]["frac"], "useful_frac": row["work_bound"]["useful_frac"],
                              "gsamples_per_s": row["algorithmic_rate"]["gsamples_per_s"]}
    out["summary"] = summary
    for k in tail_keys:
        if k in out:
            out[k] = out.pop(k)
    return out


# ---- scenes ------------------------------------------------------------------------------------

CHUNK_1024_IMAGES = {}     # rank -> the eight 1024^2 chunk images of cfg4 (generated once per process: ~1 s of host time per image)


def exchange_variant_rows(v):
    """The OPTIONAL frames of scaling_detail (N > 1): the composited frame with the exchange pipelined (ILM_GATHER_ASYNC) and with no copy
    phase at all (ILM_GATHER_STORE across processes), beside the serial strip + RCCL exchange that the record's figures are.  `v`: what
    the lit frame's block of main() had in hand (a namespace).  They run LAST, under a watchdog (run_optional_rows): nothing measured
    before them depends on them, and a hang in them costs the record these rows only.  Returns {"pipelined_exchange": .., "store_mode": ..}."""
    H, native, abi, args = v.H, v.native, v.abi, v.args
    ctx, group, glm, r, L, ranks = v.ctx, v.group, v.glm, v.r, v.L, v.ranks
    w, h, row_begin, row_end, light_frames, n_s, frame_ms, one_gpu_ms = v.w, v.h, v.row_begin, v.row_end, v.light_frames, v.n_s, v.frame_ms, v.one_gpu_ms
    barrier, max_over_ranks = ranks.barrier, ranks.max
    frame_scaling = {}

    def time_strip(b_, e_, n_=4, sync=barrier):
        for _ in range(2):
            r.RenderLighting(1.0, b_, e_, False)
        sync()
        ctx.TimerStart()
        for _ in range(n_):
            r.RenderLighting(1.0, b_, e_, False)
        return ctx.TimerStop() / n_
    # The same frames with the exchange PIPELINED (r05, ILM_GATHER_ASYNC): a ring of two group lightmaps (the reference's
    # BufferRing); the exchange of frame N runs on the member's second stream while its context stream renders the strip of
    # frame N + 1 into the other lightmap.  Reported beside the serial composited frame; a failure here is recorded, not fatal.
    try:
        glm_b = native.GroupLightmap(group, w, h, abi.LIGHTMAP_HALF4)
        glm_b.set_strips(glm.strips)
        rc_b = H.RendererConfiguration(w, h)
        rc_b.DefaultQuality = r.Configuration.DefaultQuality
        rc_b.MaximumFieldUpdatesPerFrame = 9999
        rc_b.EnableGBuffer = True
        r_b = H.LightingRenderer(ctx, rc_b, L["env"], glm_b.members[0].device_ptr())
        r_b.DistanceField = L["field"]
        r_b.UpdateFields()
        ring_ = ((r, glm), (r_b, glm_b))
        mode_ = native.GATHER_RCCL | native.GATHER_ASYNC
        for i_ in range(4):
            rr_, gg_ = ring_[i_ & 1]
            gg_.wait(); rr_.RenderLighting(1.0, row_begin, row_end, False); gg_.gather(mode_)
        glm.wait(); glm_b.wait(); barrier()
        t0 = time.perf_counter()
        for i_ in range(light_frames):
            rr_, gg_ = ring_[i_ & 1]
            gg_.wait()                                  # the exchange queued on this lightmap two frames ago
            rr_.RenderLighting(1.0, row_begin, row_end, False)
            gg_.gather(mode_)
        glm.wait(); glm_b.wait(); barrier()
        pipe_ms = max_over_ranks(time.perf_counter() - t0) / light_frames * 1e3
        same_ = bool(np.array_equal(glm.download(0).view(np.uint16), glm_b.download(0).view(np.uint16)))
        frame_scaling["pipelined_exchange"] = {
            "composited_frame_ms": round(pipe_ms, 4), "vs_serial": round(pipe_ms / frame_ms, 4), "speedup_vs_one_gpu_frame": round(one_gpu_ms / pipe_ms, 3),
            "both_lightmaps_hold_the_same_frame": same_,
            "how": "ring of two group lightmaps; ilm_group_lightmap_gather(RCCL | ILM_GATHER_ASYNC) on the member's second stream, ilm_group_lightmap_wait in front of a lightmap's reuse"}
        del ring_, rr_, gg_, r_b
        glm_b.close()
    except Exception as e_:      # noqa: BLE001 -- the serial figures above stand
        frame_scaling["pipelined_exchange"] = {"error": "%s: %s" % (type(e_).__name__, e_)}
    # The same frames with NO copy phase (r05, ILM_GATHER_STORE across processes): every rank maps the other ranks' buffers
    # through IPC handles (a collective: every rank arms or none does) and the light kernel's final store writes its strip
    # into every rank's copy over xGMI while the strip runs; what is left of the exchange is the fence (an 8-byte collective)
    # in front of the strips (readers of the old frame) and behind them.  Recorded beside the serial figures; not fatal.
    if not args.no_store_mode_row:
        try:
            for m_ in glm.members:
                m_.clear()
            glm.store_mode(True)
            try:
                for _ in range(3):
                    glm.gather(native.GATHER_STORE); r.RenderLighting(1.0, row_begin, row_end, False); glm.gather(native.GATHER_STORE)
                ctx.Sync(); barrier()
                t0 = time.perf_counter()
                for _ in range(light_frames):
                    glm.gather(native.GATHER_STORE)
                    r.RenderLighting(1.0, row_begin, row_end, False)
                    glm.gather(native.GATHER_STORE)
                ctx.Sync(); barrier()
                store_ms = max_over_ranks(time.perf_counter() - t0) / light_frames * 1e3
                store_strip_ms = ranks.doubles(time_strip(row_begin, row_end, n_s))
                glm.gather(native.GATHER_STORE); ctx.Sync(); barrier()
                stored_ = glm.download(0).view(np.uint16).copy()
            finally:
                glm.store_mode(False)
            r.RenderLighting(1.0, row_begin, row_end, False)
            glm.gather(native.GATHER_RCCL); ctx.Sync(); barrier()
            now_ = glm.download(0).view(np.uint16)
            bad_rows_ = np.nonzero((stored_.reshape(h, -1) != now_.reshape(h, -1)).any(axis=1))[0]
            rows_differing_ = [[int(v) for v in t] for t in zip(ranks.doubles(float(len(bad_rows_))), ranks.doubles(float(bad_rows_[0]) if len(bad_rows_) else -1.0),
                                                              ranks.doubles(float(bad_rows_[-1]) if len(bad_rows_) else -1.0),
                                                              ranks.doubles(float((stored_ != now_).sum())))]
            same_ = bool(ranks.sum(0.0 if np.array_equal(stored_, now_) else 1.0) == 0.0)
            frame_scaling["store_mode"] = {
                "composited_frame_ms": round(store_ms, 4), "vs_serial": round(store_ms / frame_ms, 4), "speedup_vs_one_gpu_frame": round(one_gpu_ms / store_ms, 3),
                "strip_ms_max": round(max(store_strip_ms), 4), "strip_ms": [round(t, 4) for t in store_strip_ms],
                "every_rank_holds_the_frame_of_the_rccl_exchange": same_, "rows_differing_per_rank_count_first_last_elements": rows_differing_,
                "how": "ilm_group_lightmap_store_mode (IPC-mapped buffers of the other ranks); per frame: fence, strip with mirror stores, fence (ilm_group_lightmap_gather(ILM_GATHER_STORE))"}
            del stored_
        except Exception as e_:      # noqa: BLE001 -- the serial figures above stand
            frame_scaling["store_mode"] = {"error": "%s: %s" % (type(e_).__name__, e_)}
    return frame_scaling


class SystemHandle:
    """A mirror ParticleSystem (illuminant_amd/host) as the ctypes bindings of the group entry points want it: something with .handle."""

    def __init__(self, abi, handle):
        self.handle = abi.Handle(int(handle))


def step_counted_live(ps):
    """Did the ParticleSystem.Update just issued carry ILM_STEP_COUNT_LIVE?  (IlmStepDesc.Flags at byte 20 of the descriptor the mirror
    launched last: the mirror counts every LivenessCheckInterval-th frame, ParticleLiveness.cs:14,80-105, fused into the update launch.)"""
    import struct
    return bool(struct.unpack_from("<I", ps.LastStepBytes(), 20)[0] & 1)


def particle_collective_rows(v):
    """The N > 1 particle rows WITH the collectives of the path (BASELINE config 4: "per-chunk update + RCCL all-gather"):
      with_live_counts          cfg4's per-GPU share stepped as in the headline block, and after every step that counted live particles
                                (the reference's liveness interval) ilm_group_live_counts: the whole table, identical on every rank,
                                through one small RCCL all-gather -- inside the timed steps.  The table is checked (every chunk 1024^2 live).
      with_position_all_gather  the same steps, each followed by ilm_group_gather_chunks of Pos+Life (components 0..3: 16 B per slot, 134 MB
                                per rank at 8 chunks of 1024^2) into every rank's gathered system -- what a global consumer (particle lights,
                                host readback) needs; the exchange alone by HIP events beside it; a checksum of every rank's first chunk
                                is compared with the owner's.
    They run LAST under the watchdog (run_optional_rows).  Returns {"with_live_counts": .., "with_position_all_gather": ..}."""
    import struct
    import zlib
    H, native, abi, scenes, args = v.H, v.native, v.abi, v.scenes, v.args
    ctx, group, ranks, rank, world = v.ctx, v.group, v.ranks, v.rank, v.world
    dt, k = 1.0 / 60.0, v.k
    rows = {}
    Q = build_particle_system(H, ctx, scenes, abi, 1024, 8, rank, with_spawner=False)
    ps, tp, frame = Q["ps"], Q["tp"], [0]
    sysh = SystemHandle(abi, ps.Handle)
    total = 8 * world
    state = {"calls": 0, "table": None}

    def plain():
        tp.Advance(dt); ps.Update(frame[0]); frame[0] += 1

    def with_counts():
        plain()
        if step_counted_live(ps):
            state["table"] = group.live_counts([sysh], total)
            state["calls"] += 1
    for _ in range(20):
        plain()
    try:
        COLLECTIVES.phase("particles: cfg4 share + ilm_group_live_counts at every counting step")
        n_blocks = 5
        blocks = time_blocks(ctx, ranks, with_counts, k, n_blocks)
        row = particle_row(world, Q["live"], k, blocks, v.kernel, v.traffic)
        table = state["table"]
        ok = table is not None and len(table) == total and bool((table == 1024 * 1024).all())
        row.update({"live_count_calls_per_block": round(state["calls"] / float(n_blocks), 2),
                    "live_count_table": {"chunks": total, "every_chunk_live": 1024 * 1024, "bit_exact_on_every_rank": bool(ranks.sum(0.0 if ok else 1.0) == 0.0)} if table is not None else None,
                    "collective": "ilm_group_live_counts: ilm_system_step_counts of the rank's chunks (synchronises the stream), one RCCL all-gather of %d B per rank, the table in chunk order" % (4 * 8),
                    "is": "the headline block's steps with the liveness table gathered after every counting step (every 5th Update of the mirror, as the reference's FramesUntilNextLivenessCheck)"})
        rows["with_live_counts"] = row
    except Exception as e_:      # noqa: BLE001 -- the communication-free figure stands
        rows["with_live_counts"] = {"error": "%s: %s" % (type(e_).__name__, e_)}
    try:
        COLLECTIVES.phase("particles: cfg4 share + ilm_group_gather_chunks(Pos+Life) after every step")
        nctx = native.Context(v.local_rank, borrowed_handle=ctx.Handle)
        geng = native.Engine(nctx, 1024, Q["rnd"])
        gsys = native.System(geng)
        for _ in range(total):
            gsys.add_chunk()

        def with_gather():
            plain()
            group.gather_chunks([sysh], [gsys], total, 0, 4, native.GATHER_RCCL)
        for _ in range(2):
            with_gather()
        blocks = time_blocks(ctx, ranks, with_gather, k, 3)
        row = particle_row(world, Q["live"], k, blocks, v.kernel, None)
        ranks.barrier()
        n_x = max(3, min(k, 10))
        ctx.TimerStart()
        for _ in range(n_x):
            group.gather_chunks([sysh], [gsys], total, 0, 4, native.GATHER_RCCL)
        x_ms = ranks.doubles(ctx.TimerStop() / n_x)
        ranks.barrier()
        # every rank's copy of chunk (r, 0) = table chunk r against its owner's planes: CRC32 of the first 65 536 Pos+Life slots
        own = np.zeros((65536, 4), np.float32)
        import ctypes as C_
        native.check(native.lib().ilm_chunk_download(sysh.handle, 0, abi.PLANE_POSITION, own.ctypes.data_as(C_.c_void_p), 0, 65536))
        crcs = [struct.unpack("<Q", b)[0] for b in group.host_all_gather(struct.pack("<Q", zlib.crc32(own.tobytes())))]
        same = all(zlib.crc32(np.ascontiguousarray(gsys.download(r, abi.PLANE_POSITION, 0, 65536)).tobytes()) == crcs[r] for r in range(world))
        per_rank_bytes = 8 * 4 * int(gsys.device_ptr(0, 0)[1]) * 4          # chunks x components x stride (floats) x 4 B
        row.update({"exchange_ms": round(max(x_ms), 4), "exchange_ms_per_rank": [round(t, 4) for t in x_ms],
                    "exchange_is": "HIP events around back-to-back ilm_group_gather_chunks calls alone, max over ranks",
                    "bytes_sent_per_rank_and_peer": per_rank_bytes, "bytes_received_per_rank": per_rank_bytes * (world - 1),
                    "egress_gb_per_s_per_rank": round(per_rank_bytes * (world - 1) / (max(x_ms) * 1e-3) / 1e9, 1) if world > 1 else None,
                    "gathered_chunks_match_their_owners": bool(ranks.sum(0.0 if same else 1.0) == 0.0),
                    "collective": "ilm_group_gather_chunks(components 0..3 = Pos+Life, ILM_GATHER_RCCL): one group of %d ncclSend + %d ncclRecv of %d B per rank and call, chunk planes to chunk planes (no packing)"
                                  % (8 * (world - 1), 8 * (world - 1), per_rank_bytes // 8),
                    "is": "the same steps, each followed by the Pos+Life all-gather a global consumer needs (SURVEY 8e: optional; reported with and without)"})
        row["roofline"] = None       # (the row's time is the exchange's, not the step kernel's)
        rows["with_position_all_gather"] = row
        gsys.close(); geng.close()
    except Exception as e_:      # noqa: BLE001
        rows["with_position_all_gather"] = {"error": "%s: %s" % (type(e_).__name__, e_)}
    del Q, ps
    return rows


def run_optional_rows(deferred, frames_scaling, timeout_s, rank, emit_fallback, barrier, particle_rows=None, hang_rank=False):
    """particle_rows: (namespace, dict) -- particle_collective_rows(namespace) merged into dict, first (they decide the headline).
    deferred: [(pin, namespace)] -- exchange_variant_rows for each, merged into frames_scaling[pin].  A watchdog THREAD (the main thread
    may sit inside a collective) ends the job cleanly when they have not finished in timeout_s: rank 0 prints the record assembled WITHOUT
    them (emit_fallback), every rank leaves with status 0 -- the hardware run's figures survive a hang in an optional row."""
    import threading
    done = threading.Event()

    def watchdog():
        if done.wait(timeout_s):
            return
        try:
            if rank == 0:
                emit_fallback()
        finally:
            os._exit(0)
    t = threading.Thread(target=watchdog, daemon=True)
    t.start()
    if os.environ.get("ILM_BENCH_HANG_OPTIONAL") and hang_rank:      # TEST HOOK (tests/test_two_ranks_one_gpu.py): the last rank never arrives
        time.sleep(1e6)
    if particle_rows is not None:
        particle_rows[1].update(particle_collective_rows(particle_rows[0]))
    for pin, v in deferred:
        COLLECTIVES.phase("lit frame %s: optional exchange variants (pipelined, store mode)" % pin)
        frames_scaling[pin].update(exchange_variant_rows(v))
    v = None
    barrier()               # EVERY rank is through: a rank that finished while another hangs waits here, under its own watchdog
    done.set()


def cfg4_images(scenes, rank):
    if rank not in CHUNK_1024_IMAGES:
        CHUNK_1024_IMAGES[rank] = scenes.make_particles(1000 + rank, 8 * 1024 * 1024, pos_lo=(0, 0, 0), pos_hi=(1920, 1080, 32), life=(50.0, 90.0))
    return CHUNK_1024_IMAGES[rank]


def build_particle_system(H, ctx, scenes, abi, chunk_size, n_chunks, rank, with_spawner=True, replicate_cfg4_images=False):
    """cfg2: n_chunks full chunks uploaded through Spawn(initializers) + a Spawner-fed chunk, Gravity x4 + Noise.
    replicate_cfg4_images (chunk_size 1024): chunk c holds image c mod 8 of the rank's eight cfg4 images, its particles moved by
    (3, 2) x (c div 8) pixels -- 64 chunks without 64 s of host-side generation, no two chunks alike; one Spawn call per chunk."""
    rnd = scenes.randomness_table(7)
    tp = H.ManualTimeProvider()
    ecfg = H.ParticleEngineConfiguration(chunk_size)
    ecfg.TimeProvider = tp
    engine = H.ParticleEngine(ctx, ecfg, rnd)
    cfg = H.ParticleSystemConfiguration()
    cfg.Size = [4.0, 4.0]                    # sprite half size in pixels (read by the rasteriser only)
    cfg.Friction = 0.02
    cfg.MaximumVelocity = 2048.0
    cfg.LifeDecayPerSecond = 0.01           # nobody dies during the timed steps
    col = H.ParticleColor()
    col.OpacityFromLife = 2.5
    cfg.Color = col
    ps = H.ParticleSystem(engine, cfg)
    n = chunk_size * chunk_size * n_chunks
    if replicate_cfg4_images or (chunk_size == 1024 and n_chunks == 8):
        pos, vel, attr = cfg4_images(scenes, rank)
    else:
        pos, vel, attr = scenes.make_particles(1000 + rank, n, pos_lo=(0, 0, 0), pos_hi=(1920, 1080, 32), life=(50.0, 90.0))
    if replicate_cfg4_images:
        assert chunk_size == 1024
        mc = chunk_size * chunk_size
        for c in range(n_chunks):
            sl = slice((c % 8) * mc, (c % 8 + 1) * mc)
            p = pos[sl].copy()
            p[:, 0] += np.float32(3.0 * (c // 8)); p[:, 1] += np.float32(2.0 * (c // 8))
            ps.Spawn(mc, p, np.ascontiguousarray(vel[sl]), np.ascontiguousarray(attr[sl]))
        pos = vel = attr = None           # (the CPU baseline replays cfg2, never this system)
    else:
        ps.Spawn(n, pos, vel, attr)
    assert len(ps.Chunks) == n_chunks, "ParticleSystem.MaxChunkCount is 64 (ParticleSystem.cs:49): use a larger --chunk-size"

    sp = H.Spawner(11 + rank)
    sp.MinRate = sp.MaxRate = 65536.0       # 1092 / 1093 slots per 1/60 s step through the RateError carry
    f = H.Formula3(); f.Constant = [960, 540, 0]; f.RandomScale = [900, 450, 0]; f.Type = H.FormulaType.Spherical
    sp.Position = f
    g = H.Formula3(); g.RandomScale = [60, 60, 60]; g.Type = H.FormulaType.Spherical
    sp.Velocity = g
    life = H.Formula1(); life.Constant = 50.0; life.RandomScale = 2.7
    sp.Life = life
    gr = H.Gravity()
    gr.MaximumAcceleration = 1024.0
    atts = []
    # radii / strengths in the pattern of TestGame Scenes/SimpleParticles.cs:164-180,372-398 at time 0
    for (p, r, s) in (((400., 300., 0.), 70., 600.), ((1500., 300., 0.), 150., 900.), ((400., 800., 0.), 200., 1200.), ((1500., 800., 0.), 100., 1500.)):
        a = H.Attractor(); a.Position = list(p); a.Radius = r; a.Strength = s; a.Type = H.AttractorType.Linear
        atts.append(a)
    gr.Attractors = atts
    nz = H.Noise(3 + rank)                   # defaults of Transforms.cs:192-204, Interval 1000 ms
    for t in ((sp, gr, nz) if with_spawner else (gr, nz)):
        ps.AddTransform(t)
    return dict(engine=engine, ps=ps, tp=tp, live=n, transforms=(sp, gr, nz), rnd=rnd, init=(pos, vel, attr))


def gbuffer_meshes_scene():
    """SURVEY 8f-1's bench frame: 1080p, 256 height volumes (top + front faces) and 64 billboards, as the vertex arrays a host hands over
    (float32 rows of HeightVolumeVertex / BillboardVertex).  Returns (desc, top, front, billboards)."""
    from illuminant_amd import scenes
    r16 = scenes.uniform(77, (256, 8))
    tops, fronts = [], []
    vols = []
    for v in range(256):
        cx, cy, rad = 40 + r16[v, 0] * 1840, 80 + r16[v, 1] * 960, 12 + r16[v, 2] * 50
        nv = 4 + int(r16[v, 3] * 4)
        ang = np.sort(scenes.uniform(770 + v, (nv,)) * 2 * np.pi)
        vols.append(([(cx + rad * np.cos(a), cy + rad * np.sin(a)) for a in ang], r16[v, 4] * 8, 6 + r16[v, 5] * 70))
    for poly, zb, hh in sorted(vols, key=lambda t: -(t[1] + t[2])):
        tops.append(scenes.top_face_mesh(poly, zb, hh))
        fronts.append(scenes.front_face_mesh(poly, zb, hh))
    top, front = np.concatenate(tops), np.concatenate(fronts)
    rb = scenes.uniform(78, (64, 4))
    bbv = scenes.billboard_vertices([dict(screen_bounds=((rb[k, 0] * 1800, rb[k, 1] * 960), (rb[k, 0] * 1800 + 24 + rb[k, 2] * 60, rb[k, 1] * 960 + 40 + rb[k, 3] * 80)))
                                     for k in range(64)], 0.0, 0.6)
    so, zso = scenes.self_occlusion_hacks(0.25, 128.0, 33)
    gd = scenes.gbuffer_mesh_desc(z_to_y=0.6, extent_z=128.0, self_occlusion_hack=so, z_self_occlusion_hack=zso)
    return gd, np.ascontiguousarray(top, np.float32), np.ascontiguousarray(front, np.float32), np.ascontiguousarray(bbv, np.float32)


def build_lighting(H, ctx, scenes, abi, width, height, n_lights, resolution, world, sdf_fmt, external_ptr=0, plain_twin=False, light_seed=12):
    """The configured frame: EnableGBuffer is the reference's default (LightingRenderer.Configuration.cs:106) and SURVEY 8d defines cfg3 /
    cfg5 with the ground-plane G-buffer (texel (0.5, 1, 0, 1), Vector4 = 16 B per pixel: highQualityGBuffer defaults to true,
    Configuration.cs:178-188) -- UpdateFields renders it, every pixel goes through sampleGBuffer's texture branch (LightCommon.fxh:69-144).
    plain_twin: a second renderer over the same environment and field WITHOUT a G-buffer (what r01-r03 timed), for the row beside it.
    light_seed: the lights are scenes.random_lights(light_seed, n_lights, width, height, ramp = (200, 550) x width / 1920) set on the mirror's
    SphereLightSource objects -- with seed 12 (cfg3) / 13 (cfg5) the frames tests/golden/full_frame_bands.json pins band by band."""
    env = H.LightingEnvironment()
    env.Ambient = [0.05, 0.05, 0.05, 1.0]
    sc = width / 1920.0
    xs = scenes.uniform(light_seed + 1, (n_lights,), 0, width); ys = scenes.uniform(light_seed + 2, (n_lights,), 0, height)
    zs = scenes.uniform(light_seed + 3, (n_lights,), 8.0, 64.0); rs = scenes.uniform(light_seed + 4, (n_lights,), 200.0 * sc, 550.0 * sc)
    cols = scenes.uniform(light_seed + 5, (n_lights, 3), 0.2, 1.0)
    lights = []
    for i in range(n_lights):
        l = H.SphereLightSource()
        l.Position = [float(xs[i]), float(ys[i]), float(zs[i])]
        l.Radius = 24.0; l.RampLength = float(rs[i]); l.Color = [float(cols[i, 0]), float(cols[i, 1]), float(cols[i, 2]), 1.0]
        lights.append(l)
    env.Lights = lights
    rc = H.RendererConfiguration(width, height)
    q = H.RendererQualitySettings()           # the values every demo uses (Scenes/LightProbes.cs:113-118)
    q.MinStepSize = 1.0; q.LongStepFactor = 0.5; q.MaxStepCount = 64; q.MaxConeRadius = 24.0; q.OcclusionToOpacityPower = 0.7
    rc.DefaultQuality = q
    rc.MaximumFieldUpdatesPerFrame = 9999      # the whole field in one UpdateFields (the reference default of 1 slice / frame is an amortisation knob)
    rc.EnableGBuffer = True
    renderer = H.LightingRenderer(ctx, rc, env, external_ptr)
    field = H.DistanceField(ctx, world, world, 128.0, 32, resolution, 128, sdf_fmt)
    renderer.DistanceField = field
    # the field is generated on the GPU from LightObstructions: 256 random ellipsoids / boxes, seed 11
    for (typ, center, size) in scenes.random_obstacles(11, 256, (world, world)):
        env.Obstructions.Add(H.LightObstruction(typ - 1, list(center), list(size), 0.0))
    renderer.UpdateFields()                    # the G-buffer (ground plane) and the whole field
    ctx.Sync()
    renderer.Configuration.EnableGBuffer = False      # (the field's timing below is the field's alone; the G-buffer stays bound)
    gen_iters = 5
    ctx.TimerStart()
    for _ in range(gen_iters):
        renderer.InvalidateFields()
        renderer.UpdateFields()
    gen_ms = ctx.TimerStop() / gen_iters
    renderer.Configuration.EnableGBuffer = True
    plain = None
    if plain_twin:
        rc2 = H.RendererConfiguration(width, height)
        rc2.DefaultQuality = q
        rc2.MaximumFieldUpdatesPerFrame = 9999
        rc2.EnableGBuffer = False
        plain = H.LightingRenderer(ctx, rc2, env, 0)
        plain.DistanceField = field
    texels = field.PhysicalSliceCount * field.SliceWidth * field.SliceHeight     # texels one full generation writes (8 B each)
    # The field pass is arithmetic (one analytic distance function per covering obstruction, slice and texel; 8 B written per texel):
    # its bound is vector-instruction issue, 256 CUs x 4 SIMDs x one wave64 instruction per 4 cycles at 2.4 GHz
    # (MI355X_MICROARCH.md).  Instructions per wave come from the committed PMC profile of this same scene.
    tiles = ((field.SliceWidth + 31) // 32) * ((field.SliceHeight + 7) // 8)     # one workgroup (4 waves) per 32 x 8 texels
    waves = field.PhysicalSliceCount * tiles * 4
    valu = profiled_per_wave("ilm::render_slices_kernel<%d>" % (1 if sdf_fmt == abi.SDF_FP16 else 0), "SQ_INSTS_VALU")
    issue_peak = VALU_ISSUE_PEAK
    issue = (waves * valu["value"] / (gen_ms * 1e-3) / 1e9) if valu else None
    gen = {"ms_per_field": round(gen_ms, 4), "atlas": "%dx%d RGBA16 (%d slices of %dx%d)" % (field.TextureWidth, field.TextureHeight, field.SliceCount, field.SliceWidth, field.SliceHeight),
           "obstructions": 256, "mtexels_per_s": round(texels / (gen_ms * 1e-3) / 1e6, 1),
           "hbm_write_gb_per_s": round(texels * 8 / (gen_ms * 1e-3) / 1e9, 1),
           "roofline": {"bound": "valu", "achieved": round(issue, 1) if issue else None, "peak": round(issue_peak, 1), "unit": "G wave-instr/s",
                        "frac": round(issue / issue_peak, 4) if issue else None,
                        "calibrated_peak": VALU_ISSUE_CALIBRATED, "calibrated_frac": round(issue / VALU_ISSUE_CALIBRATED, 4) if issue else None,
                        "kernel": "ilm::render_slices_kernel", "waves_per_launch": waves,
                        "valu_instructions_per_wave": round(valu["value"], 1) if valu else None,
                        "valu_source": ("profiles/%s: SQ_INSTS_VALU / SQ_WAVES" % valu["source"]) if valu else None,
                        "launch_ms": round(gen_ms, 4)}}
    return dict(renderer=renderer, renderer_plain=plain, env=env, field=field, width=width, height=height, n_lights=n_lights, field_generation=gen)


def build_collision_scene(H, ctx, scenes, abi):
    """The field the demo the configs are modelled on collides its particles with (TestGame Scenes/SimpleParticles.cs:210-284): 1920 x 1080
    x 64, 9 slices at resolution 1/4, maximumEncodedDistance 320; four cylinders of size 100-200 x 30 on a 32-pixel grid and four boxes
    around the screen's edges, generated on the device through UpdateFields."""
    env = H.LightingEnvironment()
    rc = H.RendererConfiguration(64, 64)
    rc.MaximumFieldUpdatesPerFrame = 9999
    renderer = H.LightingRenderer(ctx, rc, env, 0)
    field = H.DistanceField(ctx, 1920, 1080, 64.0, 9, 0.25, 320, abi.SDF_UNORM16)
    renderer.DistanceField = field
    r = scenes.uniform(23, (4, 3))
    for i in range(4):
        sz = 100.0 + 100.0 * float(r[i, 2])
        env.Obstructions.Add(H.LightObstruction(2, [float(int(r[i, 0] * 61) * 32), float(int(r[i, 1] * 34) * 32), 0.0], [sz, sz, 30.0], 0.0))      # Cylinder
    for center, size in (((0, -45, 0), (1920, 50, 60)), ((0, 1080 + 45, 0), (1920, 50, 60)), ((-45, 0, 0), (50, 1080, 60)), ((1920 + 45, 0, 0), (50, 1080, 60))):
        env.Obstructions.Add(H.LightObstruction(1, [float(c) for c in center], [float(c) for c in size], 0.0))                                    # Box
    renderer.UpdateFields()
    ctx.Sync()
    return dict(renderer=renderer, env=env, field=field)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    # stdout carries exactly ONE line, the JSON record: whatever native libraries print on the way (RCCL's version banner goes to
    # stdout) is sent to stderr by pointing fd 1 there until the record is written
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    if world != n_gpus:
        raise SystemExit("bench.py: launched with WORLD_SIZE=%d but --gpus %d: refusing to report a %d-GPU number from %d rank(s)"
                         % (world, n_gpus, n_gpus, world))
    from illuminant_amd import abi, native, scenes
    from illuminant_amd import _host as H

    if native.device_count() <= 0:
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    one_gpu_stand_in = bool(os.environ.get("ILM_BENCH_ONE_GPU"))
    if one_gpu_stand_in:
        # TEST HOOK (tests/test_two_ranks_one_gpu.py): every rank on device 0, the exchange through tests/fake_rccl.cpp (ILM_RCCL_LIB) -- the
        # whole N > 1 branch executed at world size > 1 on the build box's one GPU.  The record says so; its numbers mean nothing.
        local_rank = 0
    if local_rank >= native.device_count() and native.device_count() == 1 and (os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")):
        # a launcher that hands every rank ONE visible GPU of its own: that GPU is device 0 here (two ranks that were handed the SAME
        # physical GPU are refused below by RCCL itself: it will not build a communicator with a device twice)
        local_rank = 0
    if local_rank >= native.device_count():
        raise SystemExit("bench.py: rank %d wants GPU %d but this box has %d GPU(s)" % (rank, local_rank, native.device_count()))

    group = None
    comm_ranks = 0
    if world > 1 or os.environ.get("ILM_BENCH_FORCE_DIST"):
        # One process per GPU.  Everything between the ranks goes through the C ABI (ilm_group_*, csrc/group.hip) on ONE RCCL
        # communicator: the lightmap all-gather of the data path, and the barrier / max-over-ranks of the timing (small host payloads,
        # ilm_group_host_all_gather).  PyTorch is deliberately NOT imported in the ranks: its wheel carries its own HIP 7.0 + HSA
        # runtime, and a process that has initialised /opt/rocm's runtime (this library) cannot initialise a second one
        # (tools/hip_runtime_order_probe.py: "No HIP GPUs are available").  torchrun is only the launcher; rank 0 hands the 128-byte
        # RCCL id to the other ranks of the node through a file keyed by the launcher's pid.
        log_group_collectives(native, world)
        COLLECTIVES.rank, COLLECTIVES.echo = rank, (world > 1 and rank in (0, world - 1)) or bool(os.environ.get("ILM_BENCH_LOG_COLLECTIVES"))
        group = native.Group.rank(local_rank, rank, world, exchange_unique_id(native, rank, world))
        comm_ranks = group.comm_ranks()
        if comm_ranks != n_gpus:
            raise SystemExit("bench.py: the RCCL communicator reports %d rank(s) but --gpus %d" % (comm_ranks, n_gpus))
        ctx = H.DeviceContext.FromHandle(group.contexts[0].handle.value)      # the group member's context: one stream for render + gather
    else:
        ctx = H.DeviceContext(local_rank)

    import struct
    import ctypes as C_

    ranks = Ranks(group, rank, world, ctx.Sync)
    barrier, max_over_ranks, sum_over_ranks = ranks.barrier, ranks.max, ranks.sum
    forced_dist = bool(os.environ.get("ILM_BENCH_FORCE_DIST"))
    multi = world > 1 or forced_dist        # the N > 1 shape of the record (ILM_BENCH_FORCE_DIST: the same code at world 1, for one-GPU boxes)

    # ---- particles (the timed region of the contract) ------------------------------------------------------------
    COLLECTIVES.phase("particles: cfg2, all ranks (barriers and max-over-ranks of the timed blocks)")
    P = build_particle_system(H, ctx, scenes, abi, args.chunk_size, args.chunks, rank)
    ps, tp = P["ps"], P["tp"]
    dt = 1.0 / 60.0
    frame = 0
    for _ in range(args.warmup):
        tp.Advance(dt); ps.Update(frame); frame += 1
    barrier()
    spawner = P["transforms"][0]
    live_slots = P["live"]
    # One timed block = EXACTLY K steps between two barriers (the contract).  K = 20 steps are 0.6 ms of GPU time, so the block is
    # repeated: the headline is the MEDIAN block, min and max are reported beside it.  The spawner keeps adding 1092 / 1093
    # particles per step, so every block counts its own units.
    n_blocks = args.blocks if args.blocks > 0 else int(min(15, max(3, round(0.050 / max(args.steps * 30e-6, 1e-9)))))
    blocks = []
    for _ in range(n_blocks):
        spawned_before = int(spawner.TotalSpawned)      # the Spawner's own count (ps.TotalSpawnCount also counts the upload)
        barrier()
        ctx.TimerStart()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tp.Advance(dt); ps.Update(frame); frame += 1
        gpu_ms = ctx.TimerStop()          # HIP events on the context stream around the K steps (also synchronises)
        barrier()
        wall = max_over_ranks(time.perf_counter() - t0)
        spawned_after = int(spawner.TotalSpawned)
        # Units = live particles taken through the whole pass list.  The uploaded particles live through every step (LifeDecay is
        # tiny); the spawner adds 1092 / 1093 more per step (life 50 s), each updated from the step that spawns it on, so step k of
        # the block carries uploaded + spawned_before + (k + 1) * rate live particles.  Dead slots of the spawn-target chunks are
        # streamed too but are NOT counted.
        live_avg = live_slots + spawned_before + (spawned_after - spawned_before) * (args.steps + 1) / (2.0 * args.steps)
        blocks.append(dict(wall=wall, gpu_ms=gpu_ms, live_avg=live_avg, value=world * live_avg * args.steps / wall / 1e6,
                           gbs=live_avg * PARTICLE_BYTES_PER_SLOT / (gpu_ms / args.steps * 1e-3) / 1e9))
    spawned = int(spawner.TotalSpawned)
    by_value = sorted(blocks, key=lambda b: b["value"])
    med = by_value[len(by_value) // 2]
    wall, live_avg, value = med["wall"], med["live_avg"], med["value"]
    step_ms_gpu = med["gpu_ms"] / args.steps
    achieved_gbs = med["gbs"]
    step_traffic = profiled_traffic("ilm::step_lean_kernel<true, false>", "ilm::step_lean_kernel<false, false>")

    out = {
        "metric": "Mparticle-steps/sec + lit Mpixels/sec (4K, 256 point lights) at 1/2/4/8 GPUs",
        "value": round(value, 2),
        "unit": "Mparticle-steps/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "cfg2: %d particles/GPU in %d chunks of %d^2, Spawner(65536/s)+Gravity(4 attractors)+Noise+UpdatePositions"
                               % (live_slots, args.chunks, args.chunk_size),
                   "particles_per_gpu": live_slots, "spawned_per_gpu_in_run": int(spawned), "live_particles_per_step_avg": round(live_avg, 1),
                   "chunks_at_end": len(ps.Chunks), "parallelism": "chunks sharded, %d rank(s)" % world,
                   "ranks": world, "rccl_communicator_ranks": comm_ranks,
                   **({"one_gpu_stand_in": "ILM_BENCH_ONE_GPU: every rank on device 0 behind a stand-in for RCCL -- a functional run of the N > 1 branch, NOT a measurement"}
                      if one_gpu_stand_in else {}),
                   "lightmap_exchange": ("ilm_group_lightmap_gather (RCCL all-gather in place, csrc/group.hip)" if group is not None else "none (one GPU)")},
        "timed_blocks": {"blocks": n_blocks, "steps_per_block": args.steps, "headline": "median block",
                         "value_min": round(by_value[0]["value"], 2), "value_median": round(med["value"], 2), "value_max": round(by_value[-1]["value"], 2),
                         "ms_per_step_min": round(min(b["wall"] for b in blocks) / args.steps * 1e3, 5),
                         "ms_per_step_max": round(max(b["wall"] for b in blocks) / args.steps * 1e3, 5),
                         "gpu_ms_total": round(sum(b["gpu_ms"] for b in blocks), 3),
                         "roofline_frac_min": round(min(b["gbs"] for b in blocks) / HBM_PEAK_GBS, 4),
                         "roofline_frac_max": round(max(b["gbs"] for b in blocks) / HBM_PEAK_GBS, 4)},
        "roofline": {"bound": "hbm", "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
                     # cfg2's state (80 B x slots of every chunk) is smaller than the 256 MiB Infinity Cache: this rate is L3 service, priced
                     # against the HBM peak because that is the contract's roofline; the HBM-resident figure is `roofline_hbm_resident`
                     # (cfg4's per-GPU share) and the knee is on file in profiles/r03_step_working_set_sweep.txt
                     "resident": "infinity-cache (%.0f MB of particle state < %d MiB)" % (len(ps.Chunks) * args.chunk_size * args.chunk_size * 80 / 1e6, INFINITY_CACHE_MB),
                     **step_traffic_fields(step_traffic, live_avg),
                     "kernel": "ilm::step_lean_kernel<spawning> + <no spawn> (one ParticleSystem.Update = two launches: the chunk range halved over the context's two streams)",
                     "bytes_per_unit": PARTICLE_BYTES_PER_SLOT, "units_per_launch": round(live_avg, 1),
                     "launch_ms": round(step_ms_gpu, 5)},
    }

    # ---- cfg4's per-GPU share: 8 chunks of 1024^2 = 8 M particles, Gravity + Noise + UpdatePositions (no Spawner) ---------------
    # Not `value` (the contract's N = 1 workload is cfg2); reported because at this size the working set (0.9 GB) no longer fits
    # the Infinity Cache and the same kernel meets HBM.
    next_rows = {}
    if not args.no_next_rows:
        # read-back: FillReadbackResult as an ordered device-side compaction; every live particle becomes a 48-byte
        # draw-call record that lands in the context's page-locked buffer (ilm_system_readback_view): the wall time below is the
        # compaction kernels + the PCIe copy of the records
        ps.PerformReadbackView()
        barrier()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            rb = ps.PerformReadbackView()
        rb_ms = (time.perf_counter() - t0) / reps * 1e3
        n_rec = int(rb.shape[0])
        next_rows["readback_cfg2"] = {"records": n_rec, "ms_per_readback_incl_pcie": round(rb_ms, 3),
                                      "mrecords_per_s": round(n_rec / (rb_ms * 1e-3) / 1e6, 1), "host_gb_per_s": round(n_rec * 48 / (rb_ms * 1e-3) / 1e9, 2),
                                      "note": "records arrive in pinned host memory; the reference copies 3 float4 planes per chunk (48 B per SLOT) and filters on the CPU"}
        del rb
        # rasterisation: ParticleSystem.Render, technique RasterizeParticlesNoTexture -- every live particle of the cfg2
        # system as a rotated 8 x 8-pixel quad (Size (4, 4)), alpha-blended in slot order onto a 1920 x 1080 RGBA8 target
        target = H.RenderTarget(ctx, 1920, 1080, abi.LIGHTMAP_RGBA8)
        target.Clear([0.0, 0.0, 0.0, 1.0])
        rstats = ps.Render(target, abi.BLEND_ALPHA, [0.0, 0.0], [1.0, 1.0], [1.0, 1.0], [0.0, 0.0], True)
        barrier()
        reps = 5
        ctx.TimerStart()
        for _ in range(reps):
            ps.Render(target, abi.BLEND_ALPHA, [0.0, 0.0], [1.0, 1.0], [1.0, 1.0], [0.0, 0.0], False)
        rast_ms = ctx.TimerStop() / reps
        next_rows["rasterize_cfg2_1080p"] = {"ms_per_frame": round(rast_ms, 4), "live_quads": int(rstats[0]), "quad_tile_pairs": int(rstats[1]),
                                             "shaded_pixels": int(rstats[2]), "msprites_per_s": round(rstats[0] / (rast_ms * 1e-3) / 1e6, 1),
                                             "mfragments_per_s": round(rstats[2] / (rast_ms * 1e-3) / 1e6, 1),
                                             "note": "setup + scan + key emit + stable radix sort on the tile bits + one workgroup per 16 x 16 tile "
                                                     "(crowded tiles in 2048-sprite segments, combined in order); ordered blending"}
        del target
    if not args.no_next_rows and world == 1:
        # collision update (UpdateParticleSystemWithDistanceField.fx:29-147): Gravity x 4 + Noise (no spawner) stepped with
        # UpdateWithDistanceField through the demo's own field, next to the same system stepped with UpdatePositions -- cfg2's particles
        # (1 M in 16 chunks of 256^2, cache-resident) and cfg4's per-GPU share (8.4 M in 8 chunks of 1024^2, 0.67 GB: HBM-resident).  Unit of
        # the roofline: 112 B per live slot-step + 32 B per sampleDistanceFieldEx call, the call count taken on the device
        # (ilm_debug_step_sdf_samples).  Since r06 the update runs in a kernel of its own (step_lean_df_kernel, particles.hip); the
        # interpreting kernel's time (ILM_DF_LEAN=0, a launch decision read per step) is taken beside it in the same run.
        scene = build_collision_scene(H, ctx, scenes, abi)

        def collision_system(cs_c, n_c, collide):
            C = build_particle_system(H, ctx, scenes, abi, cs_c, n_c, rank, with_spawner=False)
            if collide:
                col = H.ParticleCollision()
                col.DistanceField = scene["field"]
                col.DistanceFieldMaximumZ = 256.0                         # SimpleParticles.cs:293
                col.LifePenalty = 1.0                                     # :132-134
                cfgc = C["ps"].Configuration
                cfgc.Collision = col
                C["ps"].Configuration = cfgc
            return C

        def time_collision_steps(C, fc, kc=20, n=7):
            cs_, ctp, blocks_c = C["ps"], C["tp"], []
            for _ in range(n):
                barrier()
                ctx.TimerStart()
                for _ in range(kc):
                    ctp.Advance(dt); cs_.Update(fc[0]); fc[0] += 1
                blocks_c.append(ctx.TimerStop() / kc)
            blocks_c.sort()
            return blocks_c
        for key, cs_c, n_c, resident in (("collision_step_1m", args.chunk_size, args.chunks, "infinity-cache (0.8 MB field + 84 MB of particle state < %d MiB)" % INFINITY_CACHE_MB),
                                         ("collision_step_8m", 1024, 8, "hbm (0.67 GB of particle state per step > %d MiB Infinity Cache; the 4 MB field and its cells stay in the L2 / Infinity Cache)" % INFINITY_CACHE_MB)):
            if key == "collision_step_8m" and args.no_cfg4:
                continue
            rows = {}
            for label, collide, env_ in (("plain", False, None), ("collision", True, None), ("interpreter", True, {"ILM_DF_LEAN": "0"})):
                for k_, v_ in (env_ or {}).items():
                    os.environ[k_] = v_
                try:
                    C = collision_system(cs_c, n_c, collide)
                    fc = [0]
                    for _ in range(10):
                        C["tp"].Advance(dt); C["ps"].Update(fc[0]); fc[0] += 1
                    samples_per_step = 0
                    if collide:
                        out_n = C_.c_uint64(0)
                        native.check(native.lib().ilm_debug_step_sdf_samples(ctx.Handle, 1, None))
                        C["tp"].Advance(dt); C["ps"].Update(fc[0]); fc[0] += 1
                        native.check(native.lib().ilm_debug_step_sdf_samples(ctx.Handle, 0, C_.byref(out_n)))
                        samples_per_step = int(out_n.value)
                    blocks_c = time_collision_steps(C, fc, n=(7 if label != "interpreter" else 3))
                    rows[label] = dict(ms=blocks_c[len(blocks_c) // 2], ms_min=blocks_c[0], samples=samples_per_step, live=C["live"])
                    del C
                finally:
                    for k_ in (env_ or {}):
                        os.environ.pop(k_, None)
            pl, co, it = rows["plain"], rows["collision"], rows["interpreter"]
            alg = co["live"] * PARTICLE_BYTES_PER_SLOT + co["samples"] * SDF_SAMPLE_BYTES
            units_k = 4 if key == "collision_step_8m" else 2
            kname = "ilm::step_lean_df_kernel<6, false, %s, %d>" % ("true" if key == "collision_step_8m" else "false", units_k)
            ct = profiled_traffic(kname)
            lanes_c = None
            for row_ in _newest_pmc_rows()[0]:
                if row_["kernel"].startswith(kname) and row_.get("SQ_THREAD_CYCLES_VALU") and row_.get("SQ_ACTIVE_INST_VALU") and float(row_["SQ_ACTIVE_INST_VALU"]) > 0:
                    lanes_c = round(float(row_["SQ_THREAD_CYCLES_VALU"]) / float(row_["SQ_ACTIVE_INST_VALU"]), 1)
            next_rows[key] = {
                "us_per_step": round(co["ms"] * 1e3, 2), "us_per_step_min": round(co["ms_min"] * 1e3, 2), "us_per_step_update_positions": round(pl["ms"] * 1e3, 2),
                "us_per_step_interpreting_kernel": round(it["ms"] * 1e3, 2), "vs_interpreting_kernel": round(co["ms"] / it["ms"], 3),
                "ratio_to_plain_step": round(co["ms"] / pl["ms"], 3), "particles": co["live"], "sdf_samples_per_step": co["samples"],
                "sdf_samples_per_particle": round(co["samples"] / max(co["live"], 1), 3), "sample_counts_equal_the_interpreting_kernels": bool(co["samples"] == it["samples"]),
                "field": "1920x1080x64, 9 slices at 1/4 resolution, max encoded distance 320, 4 cylinders + 4 edge boxes (SimpleParticles.cs:210-284)",
                "lanes_active_per_valu_instruction": lanes_c,
                "roofline": {"bound": "hbm", "achieved": round(alg / (co["ms"] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(alg / (co["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": round(ct["bytes"]) if ct else None,
                             "traffic_source": ("profiles/%s: FETCH_SIZE x 2 + WRITE_SIZE of the step's launches (two per step)" % ct["source"]) if ct else None,
                             "kernel": "ilm::step_lean_df_kernel<unorm16 | slice 0 | cells, %d units per wave%s> (two launches per step: the chunk range halved over the context's two streams)"
                                       % (units_k, ", streaming" if key == "collision_step_8m" else ""),
                             "bytes_per_unit": "112 B per live slot-step + 32 B per SDF sample (SURVEY 8d: 4 taps x 8 B; the kernel fetches ONE 8-byte cell per sample)",
                             "units_per_launch": {"slots": co["live"], "sdf_samples": co["samples"]},
                             "launch_ms": round(co["ms"], 5), "resident": resident,
                             "note": ("cache-served bytes priced against the HBM peak because that is the contract's roofline: read it like cfg2's fraction" if key == "collision_step_1m"
                                      else "the particle planes stream from HBM (112 B per slot); the samples' 32 B each are L2 / Infinity-Cache service priced against the HBM peak")}}
        del scene
    cpu_init, cpu_rnd, cpu_desc_bytes = P["init"], P["rnd"], ps.LastStepBytes()
    cfg4_desc_bytes = None
    scaling_particles = None
    if not args.no_cfg4:
        del P, ps, spawner          # free cfg2's chunks before the 0.9 GB system is built
        stream_kernel = "ilm::step_lean_kernel<no spawn, streaming> (two launches per step: the chunk range halved over the context's two streams)"
        stream_traffic = profiled_traffic("ilm::step_lean_kernel<false, true>")
        k4 = args.steps if multi else 30      # N > 1: this row is the headline -> EXACTLY K steps per block

        def stepper(S, warm):
            """one_step() of system S after `warm` untimed steps: the first ~40 steps after a large upload run ~10 % slower (clocks settle),
            and the rows report the steady state -- median block, the spread beside it"""
            s_, tp_, f_ = S["ps"], S["tp"], [0]

            def one():
                tp_.Advance(dt); s_.Update(f_[0]); f_[0] += 1
            for _ in range(warm):
                one()
            return one

        COLLECTIVES.phase("particles: cfg4 share (8 chunks of 1024^2 per rank), communication-free steps")
        Q = build_particle_system(H, ctx, scenes, abi, 1024, 8, rank, with_spawner=False)
        q_step = stepper(Q, 60)
        share_solo = None
        if multi:
            # rank 0 alone first (the other ranks idle at the barrier): the denominator of scaling_detail.weak_vs_share
            barrier()
            if rank == 0:
                share_solo = particle_row(1, Q["live"], k4, time_blocks(ctx, ranks.solo(), q_step, k4, 5), stream_kernel, stream_traffic)
            barrier()
        b4 = time_blocks(ctx, ranks, q_step, k4, 5)
        out["cfg4_share_8m_particles"] = particle_row(world, Q["live"], k4, b4, stream_kernel, stream_traffic)
        per_rank_rates = ranks.doubles(Q["live"] * k4 / (b4[len(b4) // 2][1] * 1e-3) / 1e6)      # each rank's own device clock
        live_share = Q["live"]
        del Q, q_step

        # cfg4 WHOLE on one device: 64 chunks of 1024^2 = 67 M particles, the reference's own ceiling (MaxChunkCount = 64,
        # ParticleSystem.cs:49), 5.4 GB of state -- the denominator of the north star's "x 6.4 from 1 to 8 GPUs at 64 M particles".
        # N = 1: a row of the line.  N > 1: rank 0 measures it ALONE, in this job, before the ranks step their shards together.
        full_64m = strong_64m = None
        COLLECTIVES.phase("particles: cfg4 whole (64 M) on rank 0 alone, then sharded over the job")
        if not args.no_cfg4_64m:
            k64 = args.steps
            if rank == 0:
                F = build_particle_system(H, ctx, scenes, abi, 1024, 64, rank, with_spawner=False, replicate_cfg4_images=True)
                full_64m = particle_row(1, F["live"], k64, time_blocks(ctx, ranks.solo(), stepper(F, 20), k64, 5), stream_kernel, stream_traffic)
                cfg4_desc_bytes = F["ps"].LastStepBytes()
                full_64m["workload"] = ("cfg4 whole on ONE GPU: 64 chunks of 1024^2 = %d particles (5.4 GB of state), Gravity(4 attractors)+Noise+UpdatePositions; "
                                        "chunk c = image c mod 8 of the share's eight, moved by (3, 2) x (c div 8) px" % F["live"])
                full_64m["roofline"]["resident"] = "hbm (5.4 GB of particle state)"
                del F
                out["cfg4_full_64m_one_gpu"] = full_64m
            if multi:
                barrier()
                # ... then the same 64 M sharded over the job: chunk c on rank c mod world = 64 / world chunks per rank, no collective
                n_mine = len(range(rank, 64, world))
                if n_mine == 8:
                    strong_64m = dict(out["cfg4_share_8m_particles"], workload="64 chunks of 1024^2 over %d ranks = the share row (8 chunks per rank)" % world)
                else:
                    S = build_particle_system(H, ctx, scenes, abi, 1024, n_mine, rank, with_spawner=False, replicate_cfg4_images=True)
                    bs = time_blocks(ctx, ranks, stepper(S, 20), k64, 5)
                    total_live = int(ranks.sum(S["live"]))
                    strong_64m = particle_row(1, total_live, k64, bs, stream_kernel, None)
                    strong_64m["particles_per_gpu"] = S["live"]
                    strong_64m["roofline"] = None        # (the row's rate is the job's; per-GPU rooflines are the share row's and one_gpu_64m's)
                    strong_64m["workload"] = "64 chunks of 1024^2 over %d rank(s): %d chunks on this rank, %d particles in the job" % (world, n_mine, total_live)
                    del S
        if multi:
            share_solo = share_solo or out["cfg4_share_8m_particles"]
            scaling_particles = dict(share=out["cfg4_share_8m_particles"], share_solo=share_solo, full_64m_solo=full_64m, strong_64m=strong_64m,
                                     per_rank_rates=per_rank_rates)

    # ---- lighting (second hot path; not part of the timed `value`) -------------------------------------------------
    if not args.no_lighting:
        lighting = {}
        frames_scaling = {}
        deferred_rows, kept_alive = [], []
        pinned = json.load(open(os.path.join(ROOT, "tests", "golden", "full_frame_bands.json")))      # the oracle's counts over the same two frames
        for name, (w, h, nl, res, wsize, fmt, lseed, pin) in (("cfg3_1080p_64_lights_unorm16", (1920, 1080, 64, 0.25, 2048, abi.SDF_UNORM16, 12, "cfg3")),
                                                               ("cfg5_4k_256_lights_fp16", (3840, 2160, 256, 0.125, 4096, abi.SDF_FP16, 13, "cfg5"))):
            # screen split into `world` equal strips of whole 16-row tile bands; with N > 1 the frame lives in the group's
            # lightmap (every rank holds world * R rows, the frame is the first h) and the strips are all-gathered in place over xGMI
            # by ilm_group_lightmap_gather on the render stream itself
            glm = None
            ext = 0
            row_begin, row_end = 0, h
            COLLECTIVES.phase("lit frame %s: strips, exchange, composited frame" % pin)
            if group is not None:
                glm = native.GroupLightmap(group, w, h, abi.LIGHTMAP_HALF4)
                ext = glm.members[0].device_ptr()
            L = build_lighting(H, ctx, scenes, abi, w, h, nl, res, wsize, fmt, ext, plain_twin=(group is None), light_seed=lseed)
            r = L["renderer"]
            if group is not None:
                # cost-balanced strips, re-cut twice from what they cost (balance_strips above); exchanged range by range at their true
                # rows (ilm_group_lightmap_set_strips + ncclSend / ncclRecv, group.hip exchange_ranges)
                packed = [abi.LightVertex.from_buffer_copy(H.LightingRenderer.PackSphereLightBytes(lsrc, 1.0, True)) for lsrc in L["env"].Lights]

                def time_strip(b_, e_, n_=4, sync=barrier):
                    for _ in range(2):
                        r.RenderLighting(1.0, b_, e_, False)
                    sync()
                    ctx.TimerStart()
                    for _ in range(n_):
                        r.RenderLighting(1.0, b_, e_, False)
                    return ctx.TimerStop() / n_
                strip_history = balance_strips(glm, ranks, h, packed, time_strip)
                row_begin, row_end = glm.strips[rank]
            if group is None and os.environ.get("ILM_BENCH_STRIP"):
                # EXPERIMENT (tools/ab_tilemap.sh): one GPU renders strip k of n equal bands only -- what a rank of an n-GPU frame launches
                spec = os.environ["ILM_BENCH_STRIP"]
                k_, n_ = (int(v) for v in spec.split(":")[-1].split("/"))
                if spec.startswith("balanced:"):        # the cost-balanced strips an n-rank run cuts (same call as below)
                    from illuminant_amd import sharding
                    packed = [abi.LightVertex.from_buffer_copy(H.LightingRenderer.PackSphereLightBytes(lsrc, 1.0, True)) for lsrc in L["env"].Lights]
                    row_begin, row_end = sharding.balanced_row_strips(h, n_, packed)[k_]
                else:
                    band = (((h + 15) // 16 + n_ - 1) // n_) * 16
                    row_begin, row_end = min(h, k_ * band), min(h, (k_ + 1) * band)
            stats = r.RenderLighting(1.0, row_begin, row_end, True)     # instrumented frame: exact SDF sample count
            # Frames per timed block: at least --light-frames, and enough to fill ~60 ms of GPU time -- a 1 ms frame timed over five
            # launches sits on the clock ramp (cfg3: 1.00 ms per frame over 4 frames, 0.92 over 40, 0.89 over 400 on the same box)
            barrier()
            ctx.TimerStart()
            for _ in range(2):
                r.RenderLighting(1.0, row_begin, row_end, False)
            est_ms = max(ctx.TimerStop() / 2.0, 1e-3)
            fill_ms = max(args.light_ms, args.sustain_s * 1e3) if (name.startswith("cfg5") and args.light_ms > 0) else args.light_ms
            light_frames = int(max_over_ranks(float(min(max(args.light_frames, int(np.ceil(fill_ms / est_ms))), 1200))))
            barrier()
            ctx.TimerStart()
            t0 = time.perf_counter()
            for _ in range(light_frames):
                r.RenderLighting(1.0, row_begin, row_end, False)
                if glm is not None:
                    glm.gather(native.GATHER_RCCL)               # queued behind the strip on the same stream: no host hand-off
            gms = ctx.TimerStop()
            barrier()
            lwall = max_over_ranks(time.perf_counter() - t0)
            samples = int(stats[0])
            pairs_local, traced_local = int(stats[1]), int(stats[2])
            samples_total = int(sum_over_ranks(samples))
            frame_ms = lwall / light_frames * 1e3
            # The timed frame IS the pinned frame: the instrumented launch's SDF-sample, pixel.light-pair and traced-pair
            # totals over the job's strips must equal the CPU oracle's over the same frame (tests/golden/full_frame_bands.json, every
            # 16-row band of it: tests/test_full_frame_bands_gpu.py) -- integer equality inside this run, or the run stops.
            verified_counts = None
            if not os.environ.get("ILM_BENCH_STRIP"):
                got_counts = (samples_total, int(sum_over_ranks(pairs_local)), int(sum_over_ranks(traced_local)))
                want_counts = (pinned[pin]["sdf_samples"], pinned[pin]["pairs"], pinned[pin]["traced"])
                if got_counts != want_counts:
                    raise SystemExit("bench.py: %s: the timed frame did (samples, pairs, traced) = %s but the oracle's pinned frame has %s" % (name, got_counts, want_counts))
                verified_counts = True
            frame_scaling = None
            if glm is not None:
                # the pieces of the composited frame, each alone: this job's strips (every rank its own, all at once), the exchange
                # (events around the gathers only) and the WHOLE frame on rank 0's GPU alone (the other ranks wait) = the named
                # denominator of speedup_vs_one_gpu_frame
                n_s = int(max(8, min(light_frames, 200)))
                strips_ms = ranks.doubles(time_strip(row_begin, row_end, n_s))
                barrier()
                ctx.TimerStart()
                for _ in range(n_s):
                    glm.gather(native.GATHER_RCCL)
                exchange_ms = ranks.doubles(ctx.TimerStop() / n_s)
                barrier()
                one_gpu_ms = 0.0
                if rank == 0:
                    one_gpu_ms = time_strip(0, h, int(max(4, min(light_frames // 4, 40))), sync=ctx.Sync)
                    r.RenderLighting(1.0, row_begin, row_end, False)
                barrier()
                glm.gather(native.GATHER_RCCL)        # (rank 0's copy of the other strips comes from their owners again)
                one_gpu_ms = ranks.max(one_gpu_ms)
                frame_scaling = {
                    "strips": [list(s_) for s_ in glm.strips], "strip_ms": [round(t, 4) for t in strips_ms], "strip_ms_max": round(max(strips_ms), 4),
                    "strip_ms_sum": round(sum(strips_ms), 4), "exchange_ms": round(max(exchange_ms), 4), "exchange_ms_per_rank": [round(t, 4) for t in exchange_ms],
                    "exchange_is": "HIP events around back-to-back ilm_group_lightmap_gather calls alone (ncclSend / ncclRecv of the strips), max over ranks",
                    "composited_frame_ms": round(frame_ms, 4), "composited_frame_is": "strip + gather per frame on one stream, wall clock, max over ranks",
                    "one_gpu_frame_ms": round(one_gpu_ms, 4), "one_gpu_frame_is": "the whole frame rendered by rank 0's GPU alone in this job (the other ranks idle)",
                    "share_ms": round(one_gpu_ms / world, 4), "speedup_vs_one_gpu_frame": round(one_gpu_ms / frame_ms, 3)}
                # The optional frames (pipelined exchange, store mode between processes) run LAST, under a watchdog: exchange_variant_rows
                import types as types_
                deferred_rows.append((pin, types_.SimpleNamespace(H=H, native=native, abi=abi, args=args, ctx=ctx, group=group, glm=glm, r=r, L=L, ranks=ranks, w=w, h=h,
                                                                  row_begin=row_begin, row_end=row_end, light_frames=light_frames, n_s=n_s, frame_ms=frame_ms, one_gpu_ms=one_gpu_ms)))
                frames_scaling[pin] = frame_scaling
            my_px = (row_end - row_begin) * w
            # this rank's launch: SDF samples + the G-buffer texel of every pixel (Vector4) + lightmap write (half4) + light records
            alg_bytes = samples * SDF_SAMPLE_BYTES + my_px * (16 + 8) + nl * 128
            kern_ms = gms / light_frames
            without_gbuffer_ms = None
            if L.get("renderer_plain") is not None:
                rp = L["renderer_plain"]
                n_plain = max(8, light_frames // 8)
                for _ in range(2):
                    rp.RenderLighting(1.0, row_begin, row_end, False)
                ctx.TimerStart()
                for _ in range(n_plain):
                    rp.RenderLighting(1.0, row_begin, row_end, False)
                without_gbuffer_ms = ctx.TimerStop() / n_plain
            # Two frames in flight (r05).  The reference keeps a ring of lightmaps because frame N + 1 is built while frame N renders
            # (BufferRing, LightingRenderer.cs:472-485).  A launch on one stream starts when the previous one has drained; rendering alternate
            # frames on a SIBLING context (ilm_ctx_create_sibling: own stream and lightmap, the SAME field, read through the library's
            # cross-context ordering) lets the next frame's first waves fill the slots the previous frame's drain leaves empty.  Same frames,
            # same bits (tests/test_frames_in_flight_gpu.py); N frames between two barriers, wall clock.  Reported BESIDE ms_per_frame, which
            # stays the one-stream figure.
            two_in_flight = None
            if group is None:
                sib = abi.Handle(0)
                native.check(native.lib().ilm_ctx_create_sibling(abi.Handle(int(ctx.Handle)), C_.byref(sib)))
                ctx2 = H.DeviceContext.FromHandle(sib.value)
                rc2 = H.RendererConfiguration(w, h)
                rc2.DefaultQuality = r.Configuration.DefaultQuality
                rc2.MaximumFieldUpdatesPerFrame = 9999
                rc2.EnableGBuffer = True
                r2 = H.LightingRenderer(ctx2, rc2, L["env"], 0)
                r2.DistanceField = L["field"]                       # owned by the first context, read by both
                r2.UpdateFields()                                   # (its own ground-plane G-buffer; the field is valid already)
                for _ in range(2):
                    r2.RenderLighting(1.0, row_begin, row_end, False)
                ctx2.Sync(); barrier()
                t0 = time.perf_counter()
                for i_ in range(light_frames):
                    (r2 if (i_ & 1) else r).RenderLighting(1.0, row_begin, row_end, False)
                ctx.Sync(); ctx2.Sync()
                two_ms = (time.perf_counter() - t0) / light_frames * 1e3
                same_bits = bool(np.array_equal(np.asarray(r.ReadLightmap(row_begin, row_end - row_begin)), np.asarray(r2.ReadLightmap(row_begin, row_end - row_begin))))
                two_in_flight = {"ms_per_frame": round(two_ms, 4), "lit_mpixels_per_s": round(w * h / (two_ms * 1e-3) / 1e6, 2), "timed_frames": light_frames,
                                 "vs_one_stream": round(two_ms / frame_ms, 4), "frames_bit_equal": same_bits,
                                 "how": "alternate frames on two sibling contexts (ilm_ctx_create_sibling): own streams and lightmaps, one field; wall clock over the block"}
                del r2
                import gc as gc_
                gc_.collect()
                if native.lib().ilm_ctx_destroy(sib) != 0:
                    print("bench.py: the sibling context still has live objects: %s" % native.lib().ilm_last_error().decode(), file=sys.stderr)
            kname = "ilm::sphere_lights_kernel<%d, false, false>" % (1 if fmt == abi.SDF_FP16 else 0)
            lt = profiled_traffic(kname) if world == 1 else None
            # the kernel's binding resource is VALU issue, not HBM (the atlas is cache-resident): wave-instructions of the committed PMC
            # profile of this same frame / this run's launch time
            lv = profiled_per_wave(kname, "SQ_INSTS_VALU") if world == 1 else None
            # the LAUNCHED grid (the profile's per-wave average is over SQ_WAVES, exit-only workgroups of partial tile groups included):
            # groups of 6 x 6 tiles dealt to 8 XCDs, four waves per tile -- a whole frame is one workgroup per tile (lighting.hip)
            waves_l = light_launch_waves(native.Context(local_rank, borrowed_handle=ctx.Handle))
            issue = (waves_l * lv["value"] / (kern_ms * 1e-3) / 1e9) if lv else None
            loop_w = TRACE_LOOP_WEIGHTED["fp16" if fmt == abi.SDF_FP16 else "unorm16"]
            lighting[name] = {
                "lit_mpixels_per_s": round(w * h / (frame_ms * 1e-3) / 1e6, 2),
                "ms_per_frame": round(frame_ms, 4), "timed_frames": light_frames, "rows": [int(row_begin), int(row_end)],
                "strip_balancing": strip_history if group is not None else None,
                "sdf_samples_per_frame": samples_total, "verified_counts": verified_counts,
                "verified_counts_is": ("the instrumented launch of THIS run: (SDF samples, pixel.light pairs, traced pairs) = the CPU oracle's totals over the same frame, "
                                       "tests/golden/full_frame_bands.json[%s] = (%d, %d, %d)" % (pin, pinned[pin]["sdf_samples"], pinned[pin]["pairs"], pinned[pin]["traced"])) if verified_counts else None,
                "scaling": frame_scaling, "two_frames_in_flight": two_in_flight,
                "pixel_light_pairs_this_rank": pairs_local, "traced_pairs_this_rank": traced_local,
                "field_generation": L["field_generation"],
                "gbuffer": "ground plane rendered by UpdateFields, Vector4 (16 B per pixel), bound: every pixel decodes its texel (LightCommon.fxh:69-144)",
                "without_gbuffer_ms": round(without_gbuffer_ms, 4) if without_gbuffer_ms else None,
                # What binds the kernel, named by the counters (profiles/r03_summary.md): vector-instruction issue.  The trace reads the
                # field's cell array (138 MB on cfg5) through L1 / L2 / Infinity Cache: L2 hit rate 98.6 %, ~0.2 GB per frame from beyond.
                "roofline": {"bound": "valu", "achieved": round(issue, 1) if issue else None, "peak": round(VALU_ISSUE_PEAK, 1), "unit": "G wave-instr/s",
                             "frac": round(issue / VALU_ISSUE_PEAK, 4) if issue else None,
                             "calibrated_peak": VALU_ISSUE_CALIBRATED, "calibrated_frac": round(issue / VALU_ISSUE_CALIBRATED, 4) if issue else None,
                             "kernel": "ilm::sphere_lights_kernel", "waves_per_launch": waves_l,
                             "valu_instructions_per_wave": round(lv["value"], 1) if lv else None,
                             "valu_instructions_per_sdf_sample": round(waves_l * lv["value"] * 64 / max(samples, 1), 1) if lv else None,
                             "counter": ("profiles/%s: SQ_INSTS_VALU / SQ_WAVES" % lv["source"]) if lv else None,
                             "traffic": round(lt["bytes"]) if lt else None, "launch_ms": round(kern_ms, 4)},
                # Issue-slot occupancy above counts whatever the kernel executes.  The WORK-based figure: S samples x the instructions a
                # sample needs / 64 lanes, over the launch time, against the same nominal issue rate -- what fraction of the chip's
                # vector issue went into necessary trace work (idle lanes, per-pair code and the launch tail all lower it).
                "work_bound": {"bound": "valu", "unit": "G wave-instr/s", "peak": round(VALU_ISSUE_PEAK, 1),
                               "instructions_per_sample": loop_w,
                               "useful_frac": round(samples * loop_w / 64.0 / (kern_ms * 1e-3) / 1e9 / VALU_ISSUE_PEAK, 4),
                               "useful_frac_r02_yardstick": round(samples * TRACE_INSTRUCTIONS_PER_SAMPLE / 64.0 / (kern_ms * 1e-3) / 1e9 / VALU_ISSUE_PEAK, 4),
                               "loop_instructions_per_sample_shipped": TRACE_LOOP_INSTRUCTIONS["fp16" if fmt == abi.SDF_FP16 else "unorm16"],
                               "useful_frac_shipped_loop": round(samples * TRACE_LOOP_INSTRUCTIONS["fp16" if fmt == abi.SDF_FP16 else "unorm16"] / 64.0 / (kern_ms * 1e-3) / 1e9 / VALU_ISSUE_PEAK, 4)},
                # SURVEY 8d's figure for this path -- S samples x 32 B + pixels x 8 B + lights x 128 B over the launch time.  It prices
                # cache-served tap bytes, so it is a sample rate: reported, without a fraction of the HBM peak (it exceeds it on cfg5).
                "algorithmic_rate": {"value": round(alg_bytes / (kern_ms * 1e-3) / 1e9, 1), "unit": "GB/s", "bytes_per_unit": SDF_SAMPLE_BYTES,
                                     "units_per_launch": samples, "gsamples_per_s": round(samples / (kern_ms * 1e-3) / 1e9, 2)},
            }
            if world == 1:
                # How far is the parity frame (fp32 accumulation over the lights) from what the reference's HalfVector4 render target would
                # hold (rounded by the ROP after every light, ilm_ctx_set_lightmap_blend)?  Same kernel, same lights, both read back.
                def read_frame():
                    raw = np.asarray(r.ReadLightmap(0, h))
                    return (raw.view(np.float16) if raw.dtype == np.uint16 else raw).astype(np.float32)
                plain = read_frame()
                nctx_b = native.Context(local_rank, borrowed_handle=ctx.Handle)
                nctx_b.set_lightmap_blend(True)
                r.RenderLighting(1.0, row_begin, row_end, False)
                ctx.Sync()
                nctx_b.set_lightmap_blend(False)
                rop = read_frame()
                r.RenderLighting(1.0, row_begin, row_end, False)
                rel = np.abs(rop[..., :3] - plain[..., :3]) / np.maximum(np.abs(plain[..., :3]), 1e-3)
                lighting[name]["fp16_per_light_blend_vs_fp32_accumulate"] = {
                    "max_relative_rgb": round(float(rel.max()), 6), "mean_relative_rgb": round(float(rel.mean()), 7),
                    "texels_that_differ": round(float((rop[..., :3] != plain[..., :3]).any(axis=-1).mean()), 4),
                    "lights_per_pixel_max": int(plain[..., 3].max() - 1.0),
                    "note": "the reference's lightmap is HalfVector4 and its ROP rounds after every light (LightingRenderer.cs:476-479); the 1e-4 parity "
                            "bar is defined against fp32 accumulation (both read back from the fp16 lightmap here)"}
            if glm is not None:
                # the composite entry point a C# host calls (ilm_group_render_sphere_lights: strip + gather in one call) must give
                # the frame the mirror-rendered strip + ilm_group_lightmap_gather gave: checksum of both on rank 0's copy
                ref_frame = glm.download(0)
                glm2 = native.GroupLightmap(group, w, h, abi.LIGHTMAP_HALF4)
                glm2.set_strips(glm.strips)
                verts = (abi.LightVertex * nl)()
                for i, lsrc in enumerate(L["env"].Lights):
                    verts[i] = abi.LightVertex.from_buffer_copy(H.LightingRenderer.PackSphereLightBytes(lsrc, 1.0, True))
                envu = abi.Environment.from_buffer_copy(r.GetEnvironmentUniformsBytes())
                dfuu = abi.DistanceFieldUniforms.from_buffer_copy(r.GetDistanceFieldUniformsBytes())

                class _Sdf:      # the mirror's field, by handle
                    handle = abi.Handle(int(L["field"].TextureHandle))
                group.render_sphere_lights(verts, envu, dfuu, None, [_Sdf], (0.05, 0.05, 0.05, 1.0), glm2, native.GATHER_RCCL)
                group.sync()
                same = bool(np.array_equal(glm2.download(0).view(np.uint16), ref_frame.view(np.uint16)))
                lighting[name]["exchange"] = {"mode": "ilm_group_lightmap_gather: cost-balanced strips (balanced_row_strips), ncclSend / ncclRecv range exchange in place"
                                                      if world > 1 else "ilm_group_lightmap_gather (one rank: nothing to exchange)", "ranks": world,
                                              "bytes_this_rank": (row_end - row_begin) * w * 8, "strips": glm.strips,
                                              "composite_call_matches": same}
                glm2.close()
                del ref_frame
            if name.startswith("cfg5"):
                # cfg5: + 16 M particles = 16 chunks of 1024^2 stepped in the same frame (cfg2's transform list without the
                # spawner): ALL of them at N = 1 (VERDICT r05 #1d), 16 / N per rank otherwise; particle step and lit frame as two phases
                # and as a whole, one stream
                n5 = len(range(rank, 16, world))              # 16 chunks of 1024^2 = the config's 16 M particles: all of them at N = 1, chunk c on rank c mod N otherwise
                Q5 = build_particle_system(H, ctx, scenes, abi, 1024, n5, rank, with_spawner=False, replicate_cfg4_images=True)
                q5, q5tp = Q5["ps"], Q5["tp"]
                for f5 in range(3):
                    q5tp.Advance(1.0 / 60.0); q5.Update(f5)
                barrier()
                frames5 = max(args.light_frames, 12)
                ctx.TimerStart()
                for f5 in range(frames5):
                    q5tp.Advance(1.0 / 60.0); q5.Update(3 + f5)
                step5_ms = ctx.TimerStop() / frames5
                barrier()
                t5 = time.perf_counter()
                ctx.TimerStart()
                for f5 in range(frames5):
                    q5tp.Advance(1.0 / 60.0); q5.Update(3 + frames5 + f5)
                    r.RenderLighting(1.0, row_begin, row_end, False)
                    if glm is not None:
                        glm.gather(native.GATHER_RCCL)
                whole5_dev = ctx.TimerStop() / frames5        # this rank's stream, HIP events (r04: the host-clock figure over three frames carried 0.3-0.5 ms of bracket)
                barrier()
                whole5 = max_over_ranks(time.perf_counter() - t5) / frames5 * 1e3
                # the lit frame alone through the same loop (same frame count, same clock): what the step adds is the difference
                barrier()
                ctx.TimerStart()
                for f5 in range(frames5):
                    r.RenderLighting(1.0, row_begin, row_end, False)
                    if glm is not None:
                        glm.gather(native.GATHER_RCCL)
                lit5_dev = ctx.TimerStop() / frames5
                lighting[name]["with_particles"] = {
                    "particles_per_gpu": Q5["live"], "chunks_per_gpu": n5, "particles_in_the_job": int(sum_over_ranks(Q5["live"])),
                    "is": "cfg5's 16 M particles (16 chunks of 1024^2, chunk c on rank c mod %d) stepped in the same frame as the lit frame: two phases and the whole, one stream" % world,
                    "particle_step_ms": round(step5_ms, 4),
                    "particle_step_gb_per_s": round(Q5["live"] * PARTICLE_BYTES_PER_SLOT / (step5_ms * 1e-3) / 1e9, 1),
                    "frame_ms_step_plus_lighting": round(whole5, 4), "frame_ms_step_plus_lighting_device_clock": round(whole5_dev, 4),
                    "lit_frame_alone_same_loop_device_clock": round(lit5_dev, 4), "timed_frames": frames5,
                    "lit_mpixels_per_s_with_particles": round(w * h / (whole5 * 1e-3) / 1e6, 2)}
                del Q5, q5
                if group is None and world == 1:
                    # (r06) The same frame with the two phases SIDE BY SIDE: ParticleSystem.Update and RenderLighting of a frame share nothing
                    # (the particles light nothing here), the step is HBM-bound and the cone trace issue-bound, so a host that keeps the
                    # particle system on a SIBLING context (ilm_ctx_create_sibling: streams of its own) lets the step run under the lit
                    # frame.  Same steps, same frames, same bits (each context's work is ordered as before); wall clock over the block,
                    # both contexts drained.  Reported BESIDE the one-stream figure.
                    import gc as gc5_
                    sib5 = abi.Handle(0)
                    native.check(native.lib().ilm_ctx_create_sibling(abi.Handle(int(ctx.Handle)), C_.byref(sib5)))
                    ctx5 = H.DeviceContext.FromHandle(sib5.value)
                    Q5b = build_particle_system(H, ctx5, scenes, abi, 1024, n5, rank, with_spawner=False, replicate_cfg4_images=True)
                    for f5 in range(3):
                        Q5b["tp"].Advance(1.0 / 60.0); Q5b["ps"].Update(f5)
                        r.RenderLighting(1.0, row_begin, row_end, False)
                    ctx5.Sync(); ctx.Sync()
                    # (the light pass fills every SIMD's registers -- eight waves of 64 -- so a step queued BEFORE it simply runs first; queued
                    # AFTER it, the step's waves take the slots the frame's drain leaves empty: both orders are timed)
                    side5, f5n = {}, 3
                    for order in ("step_first", "frame_first"):
                        t5 = time.perf_counter()
                        for f5 in range(frames5):
                            if order == "frame_first":
                                r.RenderLighting(1.0, row_begin, row_end, False)
                            Q5b["tp"].Advance(1.0 / 60.0); Q5b["ps"].Update(f5n); f5n += 1
                            if order == "step_first":
                                r.RenderLighting(1.0, row_begin, row_end, False)
                        ctx5.Sync(); ctx.Sync()
                        side5[order] = (time.perf_counter() - t5) / frames5 * 1e3
                    best5 = min(side5.values())
                    lighting[name]["with_particles"]["step_beside_the_lit_frame"] = {
                        "frame_ms": round(best5, 4), "frame_ms_step_queued_first": round(side5["step_first"], 4), "frame_ms_lit_frame_queued_first": round(side5["frame_first"], 4),
                        "vs_one_stream": round(best5 / whole5, 4), "lit_mpixels_per_s": round(w * h / (best5 * 1e-3) / 1e6, 2), "timed_frames": frames5,
                        "how": "the particle system lives on a sibling context (ilm_ctx_create_sibling: its own streams): the 16 M-particle step runs beside the lit frame; wall clock over the block, both contexts drained"}
                    del Q5b
                    gc5_.collect()
                    if native.lib().ilm_ctx_destroy(sib5) != 0:
                        print("bench.py: the particle sibling context still has live objects: %s" % native.lib().ilm_last_error().decode(), file=sys.stderr)
            if rank == 0 and world == 1 and not args.no_cpu_baseline:
                # the oracle on a bounded band of rows of the same frame (same generated field, same packed lights), on the host cores
                from oracle import oracle as orc
                atlas_host = L["field"].Save()
                verts = (abi.LightVertex * nl)()
                for i, lsrc in enumerate(L["env"].Lights):
                    verts[i] = abi.LightVertex.from_buffer_copy(H.LightingRenderer.PackSphereLightBytes(lsrc, 1.0, True))
                envu = abi.Environment.from_buffer_copy(r.GetEnvironmentUniformsBytes())
                dfuu = abi.DistanceFieldUniforms.from_buffer_copy(r.GetDistanceFieldUniformsBytes())
                tex = orc.make_texture(atlas_host, fmt)
                # a band of rows around the frame's middle, grown until the oracle works on it for at least ~2.5 s (the 0.2 s / 1.1 s
                # samples of r05 were order-of-magnitude only); the band's SDF samples are counted by the oracle itself, so the CPU's rate is
                # stated per sample too -- the frames differ 5 x in samples per pixel, the CPU's sample rate should not
                rows, mid, want_s = 4, h // 2, max(2.5, min(args.cpu_seconds / 3.0, 6.0))
                while True:
                    rows = int(min(rows, h))                # (cfg3's whole frame is ~1.5 s of 16 cores: then the sample is the frame)
                    t0 = time.perf_counter()
                    _, ost = orc.render_sphere_lights(verts, envu, dfuu, None, tex, (0.05, 0.05, 0.05, 1.0), w, h, row_begin=mid - rows // 2, row_end=mid - rows // 2 + rows, want_stats=True)
                    el = time.perf_counter() - t0
                    if el >= want_s or rows >= h:
                        break
                    rows = int(np.ceil(rows * min(max(1.25 * want_s / max(el, 1e-3), 1.5), 64.0)))
                lighting[name]["cpu_baseline"] = {"value": round(rows * w / el / 1e6, 4), "unit": "lit Mpixels/s", "cores": orc.num_threads(), "kind": "port",
                                                  "msamples_per_s": round(int(ost.SdfSamples) / el / 1e6, 1), "sdf_samples_in_sample": int(ost.SdfSamples),
                                                  "sample": "%d rows x %d px around the frame's middle (oracle/ilm_oracle.c, OpenMP, %.1f s)" % (rows, w, el)}
            if glm is not None and deferred_rows and deferred_rows[-1][1].glm is glm:
                kept_alive.append((L, r, glm))          # (the optional rows at the end render with them)
            elif glm is not None:
                glm.close()
            del L, r
        out["lighting"] = lighting
        out["lit_mpixels_per_s"] = lighting["cfg5_4k_256_lights_fp16"]["lit_mpixels_per_s"]
        # the second hot path's roofline next to the first one's, where the driver's `parsed` sees it
        l5 = lighting["cfg5_4k_256_lights_fp16"]
        out["roofline_lighting"] = {"workload": "cfg5: 4K, 256 lights, fp16 samples", "bound": "valu", "achieved": l5["roofline"]["achieved"], "peak": l5["roofline"]["peak"],
                                    "unit": "G wave-instr/s", "frac": l5["roofline"]["frac"], "useful_frac": l5["work_bound"]["useful_frac"],
                                    "instructions_per_sample": l5["work_bound"]["instructions_per_sample"], "launch_ms": l5["roofline"]["launch_ms"], "verified_counts": l5["verified_counts"],
                                    "gbuffer": "bound (ground plane, Vector4)", "without_gbuffer_ms": l5["without_gbuffer_ms"],
                                    "timed_frames": l5["timed_frames"], "sdf_samples_per_frame": l5["sdf_samples_per_frame"], "traffic": l5["roofline"]["traffic"]}

        l3 = lighting["cfg3_1080p_64_lights_unorm16"]
        # cfg3 beside it, with what ELSE limits it said in the record: its issue fraction is not slack -- the unorm16
        # field's samples are two 8-byte typed loads per lane whose data path (TD) is 82 % busy at this sample rate
        # (profiles/r03_typed_unorm16_loads.txt, docs/experiments.md 3.2 "cfg3, accounted": 87 % of its vector instructions are trace)
        out["roofline_lighting_cfg3"] = {"workload": "cfg3: 1080p, 64 lights, unorm16 samples", "bound": "valu", "achieved": l3["roofline"]["achieved"], "peak": l3["roofline"]["peak"],
                                         "unit": "G wave-instr/s", "frac": l3["roofline"]["frac"], "useful_frac": l3["work_bound"]["useful_frac"],
                                         "co_limit": "TD 82 % busy: the texture-data path returns the unorm16 taps as 2 x 8 B typed loads per lane-sample (TD_TD_BUSY, profiles/r03_typed_unorm16_loads.txt); "
                                                     "the issue fraction is not slack", "launch_ms": l3["roofline"]["launch_ms"], "verified_counts": l3["verified_counts"],
                                         "sdf_samples_per_frame": l3["sdf_samples_per_frame"], "traffic": l3["roofline"]["traffic"]}
        if not args.no_next_rows and world == 1:
            # particle lights: 4 096 live particles of a 64^2 chunk lighting a 1080p frame through cfg3's field
            L = build_lighting(H, ctx, scenes, abi, 1920, 1080, 0, 0.25, 2048, abi.SDF_UNORM16)
            rnd = scenes.randomness_table(7)
            ecfg = H.ParticleEngineConfiguration(64)
            eng = H.ParticleEngine(ctx, ecfg, rnd)
            pcfg = H.ParticleSystemConfiguration()
            pcfg.LifeDecayPerSecond = 0.01
            lsys = H.ParticleSystem(eng, pcfg)
            pos, vel, attr = scenes.make_particles(91, 4096, pos_lo=(0, 0, 4), pos_hi=(1920, 1080, 48), life=(50.0, 90.0))
            lsys.Spawn(4096, pos, vel, attr)
            lsys.Update(0)                                   # one step so that RenderColor is populated
            pls = H.ParticleLightSource()
            tmpl = H.SphereLightSource()
            tmpl.Radius = 4.0; tmpl.RampLength = 60.0; tmpl.Color = [1.0, 0.9, 0.8, 1.0]
            pls.Template = tmpl
            pls.System = lsys
            L["env"].ParticleLights = [pls]
            r = L["renderer"]
            stats = r.RenderLighting(1.0, 0, -1, True)
            ctx.Sync()
            # (r06: the row follows seconds of CPU-only work -- the lighting cpu_baseline legs -- and three frames after one warm-up frame were
            # timed on a device still at its idle clocks: 1.30-1.34 ms where the same library gives 1.04.  Warm for >= 40 frames, then the
            # median of seven blocks of ten.)
            # (under the profiling flags -- --light-ms 0 -- two warm-up frames and one block of three: every frame here is also an ambient-only
            # launch of the SPHERE-light kernel, and 110 of those made it the median dispatch of that kernel in the PMC summary)
            for _ in range(40 if args.light_ms > 0 else 2):
                r.RenderLighting(1.0, 0, -1, False)
            ctx.Sync()
            frames, pl_blocks = (10 if args.light_ms > 0 else 3), []
            for _ in range(7 if args.light_ms > 0 else 1):
                ctx.TimerStart()
                for _ in range(frames):
                    r.RenderLighting(1.0, 0, -1, False)
                pl_blocks.append(ctx.TimerStop() / frames)
            pl_blocks.sort()
            pl_ms = pl_blocks[len(pl_blocks) // 2]
            alg = int(stats[0]) * SDF_SAMPLE_BYTES + 1920 * 1080 * 16 + 4096 * (32 + 128)
            # vector-instruction issue of the wide-binning instantiation of the light kernel in the committed PMC profile of this bench
            plv = profiled_per_wave("ilm::sphere_lights_kernel<0, false, true>", "SQ_INSTS_VALU")
            # waves of the launch: whole groups of 6 x 6 tiles (lighting.hip tile_map 4), four waves per tile -- what SQ_WAVES counted
            pl_waves = light_launch_waves(native.Context(local_rank, borrowed_handle=ctx.Handle))
            pl_issue = (pl_waves * plv["value"] / (pl_ms * 1e-3) / 1e9) if plv else None
            next_rows["particle_lights_1080p_4096"] = {
                "ms_per_frame": round(pl_ms, 4), "ms_per_frame_min": round(pl_blocks[0], 4), "ms_per_frame_max": round(pl_blocks[-1], 4),
                "timed_blocks": {"blocks": len(pl_blocks), "frames_per_block": frames, "headline": "median block"},
                "lit_mpixels_per_s": round(1920 * 1080 / (pl_ms * 1e-3) / 1e6, 1), "lights": 4096,
                "sdf_samples_per_frame": int(stats[0]), "pixel_light_pairs": int(stats[1]),
                "roofline": {"bound": "valu", "achieved": round(pl_issue, 1) if pl_issue else None, "peak": round(VALU_ISSUE_PEAK, 1), "unit": "G wave-instr/s",
                             "frac": round(pl_issue / VALU_ISSUE_PEAK, 4) if pl_issue else None, "traffic": None,
                             "calibrated_peak": VALU_ISSUE_CALIBRATED, "calibrated_frac": round(pl_issue / VALU_ISSUE_CALIBRATED, 4) if pl_issue else None,
                             "kernel": "ilm::sphere_lights_kernel<unorm16, wide tile lists> (accumulate, device-side light count) + particle_light_count/emit",
                             "valu_instructions_per_wave": round(plv["value"], 1) if plv else None,
                             "counter": ("profiles/%s: SQ_INSTS_VALU / SQ_WAVES (the launch also contains the exit-only workgroups of partial tile groups)" % plv["source"]) if plv else None,
                             "launch_ms": round(pl_ms, 4)},
                "algorithmic_rate": {"value": round(alg / (pl_ms * 1e-3) / 1e9, 1), "unit": "GB/s", "bytes_per_unit": SDF_SAMPLE_BYTES,
                                     "units_per_launch": int(stats[0])}}
            del L, r, lsys, eng, pls
            # light probes: 256 probes under cfg3's 64 lights and field, one synchronous ilm_render_light_probes call
            L = build_lighting(H, ctx, scenes, abi, 1920, 1080, 64, 0.25, 2048, abi.SDF_UNORM16)
            r = L["renderer"]
            pverts = (abi.LightVertex * 64)()
            for i, lsrc in enumerate(L["env"].Lights):
                pverts[i] = abi.LightVertex.from_buffer_copy(H.LightingRenderer.PackSphereLightBytes(lsrc, 1.0, True))
            penv = abi.Environment.from_buffer_copy(r.GetEnvironmentUniformsBytes())
            pdfu = abi.DistanceFieldUniforms.from_buffer_copy(r.GetDistanceFieldUniformsBytes())

            class _ProbeSdf:      # the mirror's field, by handle
                handle = abi.Handle(int(L["field"].TextureHandle))
            pctx = native.Context(local_rank, borrowed_handle=ctx.Handle)
            npr = 256
            ppos = np.ones((npr, 4), np.float32)
            ppos[:, 0] = scenes.uniform(5, (npr,), 0.0, 1920.0); ppos[:, 1] = scenes.uniform(6, (npr,), 0.0, 1080.0); ppos[:, 2] = scenes.uniform(7, (npr,), 0.0, 32.0)
            pnrm = np.zeros((npr, 4), np.float32); pnrm[:, 2] = 1.0; pnrm[:, 3] = 1.0
            for _ in range(5):
                pv = native.render_light_probes(pctx, pverts, ppos, pnrm, penv, pdfu, _ProbeSdf)
            t0 = time.perf_counter()
            for _ in range(100):
                pv = native.render_light_probes(pctx, pverts, ppos, pnrm, penv, pdfu, _ProbeSdf)
            probe_us = (time.perf_counter() - t0) / 100 * 1e6
            next_rows["light_probes_256_x_64_lights"] = {
                "us_per_call": round(probe_us, 1), "probes": npr, "lights": 64, "lit_probe_light_pairs": int(pv[:, 3].sum()),
                "note": "synchronous call on the host's clock: one pinned block in, prepare kernel, one wave per (64 probes, light), a sum in light order, values read from the same block (r03's one-lane-per-probe kernel with three uploads and a read-back: 1 207 us)"}
            # the reference's default cadence for a dynamic field: MaximumFieldUpdatesPerFrame = 1 slice triplet re-rendered per
            # frame in front of the lit frame (LightingRenderer.Configuration.cs:91) -- cfg3's frame with and without it
            r.Configuration.MaximumFieldUpdatesPerFrame = 1
            dyn = {}
            for tag in ("static", "one_triplet_per_frame"):
                r.InvalidateFields(); r.Configuration.MaximumFieldUpdatesPerFrame = 9999; r.UpdateFields()
                r.Configuration.MaximumFieldUpdatesPerFrame = 1
                r.RenderLighting(1.0, 0, -1, False)
                ctx.Sync()
                ctx.TimerStart()
                for k in range(44):                           # four passes over the field's eleven triplets
                    if tag != "static":
                        if k % 11 == 0:
                            r.InvalidateFields()
                        r.UpdateFields()
                    r.RenderLighting(1.0, 0, -1, False)
                dyn[tag] = ctx.TimerStop() / 44
            next_rows["cfg3_frame_with_a_dynamic_field"] = {
                "ms_per_frame": round(dyn["one_triplet_per_frame"], 4), "static_field_ms_per_frame": round(dyn["static"], 4),
                "note": "one slice triplet (of eleven) re-rendered per frame + the cells of the four slices around it, then the lit frame; docs/experiments.md 3.4"}
            r.Configuration.MaximumFieldUpdatesPerFrame = 9999
            del L, r
            # lightmap resolve: 4K HalfVector4 lightmap -> RGBA8, ToneMap; 8 B read + 4 B written per pixel
            L = build_lighting(H, ctx, scenes, abi, 3840, 2160, 8, 0.125, 4096, abi.SDF_FP16)
            r = L["renderer"]
            r.RenderLighting(1.0, 0, -1, False)
            hc = abi.HDRConfiguration()
            hc.Mode, hc.InverseScaleFactor, hc.Exposure, hc.Gamma, hc.WhitePoint = abi.HDR_TONE_MAP, 1.0, 1.2, 1.0 / 2.2, 3.0
            hdr = bytes(hc)
            rs_ms = r.BenchResolve(hdr, abi.LIGHTMAP_RGBA8, 20)
            px = 3840 * 2160
            next_rows["resolve_4k_half4_to_rgba8"] = {
                "ms_per_frame": round(rs_ms, 4), "mpixels_per_s": round(px / (rs_ms * 1e-3) / 1e6, 1),
                "roofline": {"bound": "hbm", "achieved": round(px * 12 / (rs_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(px * 12 / (rs_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "ilm::resolve_kernel",
                             "bytes_per_unit": 12, "units_per_launch": px, "launch_ms": round(rs_ms, 4)}}
            del L, r

        if not args.no_next_rows and world == 1:
            # 2.5D G-buffer from the host's meshes: 1080p, 256 height volumes (top + front faces) and 64 billboards
            nctx = native.Context(local_rank, borrowed_handle=ctx.Handle)
            gd, top, front, bbv = gbuffer_meshes_scene()
            gbt = native.GBufferTexture(nctx, None, abi.GBUFFER_FLOAT4, size=(1920, 1080))
            gruns = [(None, 0, 64, abi.BILLBOARD_MASK)]
            for _ in range(5):
                gbt.render_meshes(gd, top, front, bbv, gruns)
            nctx.sync()
            nctx.timer_start()
            for _ in range(200):
                gbt.render_meshes(gd, top, front, bbv, gruns)
            gb_ms = nctx.timer_stop() / 200
            tris = 2 + len(top) // 3 + len(front) // 3 + 128
            px = 1920 * 1080
            next_rows["gbuffer_2p5d_1080p"] = {
                "ms_per_frame": round(gb_ms, 4), "triangles": int(tris), "mpixels_per_s": round(px / (gb_ms * 1e-3) / 1e6, 1),
                "roofline": {"bound": "hbm", "achieved": round(px * 16 / (gb_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(px * 16 / (gb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                             "kernel": "ilm::gbuffer_setup_kernel (reads the vertex arrays in the pinned ring) + ilm::gbuffer_bin_kernel + ilm::gbuffer_meshes_kernel", "bytes_per_unit": 16, "units_per_launch": px,
                             "launch_ms": round(gb_ms, 4),
                             "note": "one 16 B store per texel is the algorithmic traffic; the frame is the setup kernel reading the vertex arrays where the host left them (11 us), the block-binning kernel (6 us) and the raster kernel (40 us), which is bound by instruction issue (per-candidate scalar code and per-pixel shader arithmetic), not by the store: docs/experiments.md 3.4"}}
            gbt.close()

    if next_rows:
        out["next_rows"] = next_rows

    # ---- CPU baseline: the oracle restatement on the host cores (rank 0, N == 1 only) -------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        import ctypes as C
        cs, nch = args.chunk_size, args.chunks
        n = cs * cs
        pos, vel, attr = cpu_init
        chunks = []
        for c in range(nch):
            sl = slice(c * n, (c + 1) * n)
            chunks.append([pos[sl].copy(), vel[sl].copy(), attr[sl].copy(), np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)])
        chunks.append([np.zeros((n, 4), np.float32) for _ in range(5)])
        d = abi.StepDesc.from_buffer_copy(cpu_desc_bytes)   # the very descriptor the GPU ran last
        steps_done = 0
        t0 = time.perf_counter()
        while True:
            orc.step(chunks, cs, cpu_rnd, d)
            steps_done += 1
            el = time.perf_counter() - t0
            if el > (args.cpu_seconds / 3.0 if cfg4_desc_bytes is not None else args.cpu_seconds):
                break
        cfg2_cpu = {"value": round(live_slots * steps_done / el / 1e6, 2), "unit": "Mparticle-steps/s",   # spawned particles not counted here (< 1 % over the sample)
                    "cores": orc.num_threads(), "kind": "port",
                    "sample": "%d steps of the same cfg2 system (oracle/ilm_oracle.c, OpenMP, %.1f s)" % (steps_done, el)}
        out["cpu_baseline"] = cfg2_cpu
        if cfg4_desc_bytes is not None:
            # the headline workload is cfg4 (64 chunks of 1024^2): its bounded sample = 2 of the 64 chunks (images 0 and 1 of the share's
            # eight, 2.1 M particles) through the very descriptor the GPU ran last (Gravity x 4 + Noise + UpdatePositions)
            p4, v4, a4 = cfg4_images(scenes, rank)
            m4 = 1024 * 1024
            chunks4 = [[p4[c * m4:(c + 1) * m4].copy(), v4[c * m4:(c + 1) * m4].copy(), a4[c * m4:(c + 1) * m4].copy(), np.zeros((m4, 4), np.float32), np.zeros((m4, 4), np.float32)]
                       for c in range(2)]
            d4 = abi.StepDesc.from_buffer_copy(cfg4_desc_bytes)
            d4.FirstChunk, d4.ChunkCount = 0, -1
            steps4 = 0
            t0 = time.perf_counter()
            while True:
                orc.step(chunks4, 1024, cpu_rnd, d4)
                steps4 += 1
                el4 = time.perf_counter() - t0
                if el4 > args.cpu_seconds:
                    break
            out["cpu_baseline"] = {"value": round(2 * m4 * steps4 / el4 / 1e6, 2), "unit": "Mparticle-steps/s", "cores": orc.num_threads(), "kind": "port",
                                   "sample": "%d steps of 2 of cfg4's 64 chunks of 1024^2 (2.1 M particles, Gravity x 4 + Noise + UpdatePositions; oracle/ilm_oracle.c, OpenMP, %.1f s)" % (steps4, el4)}
            out["cpu_baseline_cfg2"] = cfg2_cpu

    # Why the CPU baseline is the oracle ("port") and not the reference on D3D WARP: probed, not assumed.
    import platform
    import shutil
    probe = {"os": platform.system(), "tools_on_path": {t: bool(shutil.which(t)) for t in ("dotnet", "mono", "msbuild", "csc", "fxc", "dxc", "wine")}}
    probe["warp_available"] = bool(probe["os"] == "Windows" and probe["tools_on_path"]["dotnet"])
    probe["note"] = ("D3D WARP is a Windows component and the reference needs .NET + fxc + Fracture + FNA; none of it is on this box, "
                     "so cpu_baseline.kind is \"port\" (oracle/, OpenMP)") if not probe["warp_available"] else "WARP could run here; the reference itself is still not shippable to the box"
    out["warp_probe"] = probe

    if group is not None:
        out["config"]["rccl_communicator_ranks_per_rank"] = [struct.unpack("<i", b[:4])[0] for b in group.host_all_gather(struct.pack("<ii", comm_ranks, 0))]
    import copy

    collective_rows = {}

    # ---- the box's own HBM rate beside the spec figure ("a calibration run ... the calibrated number next to the spec number") ----
    if rank == 0 and not args.dry_collectives:
        try:
            cal = calibrate_hbm_copy(native, local_rank, ctx)
            out["hbm_calibration"] = cal
            for key in ("roofline", "roofline_hbm_resident"):
                rf = out.get(key)
                if isinstance(rf, dict) and rf.get("bound") == "hbm" and rf.get("achieved"):
                    rf["calibrated_peak"] = cal["copy_gb_per_s"]
                    rf["calibrated_frac"] = round(rf["achieved"] / cal["copy_gb_per_s"], 4)
            for key in ("cfg4_full_64m_one_gpu", "cfg4_share_8m_particles"):
                rf = (out.get(key) or {}).get("roofline")
                if isinstance(rf, dict) and rf.get("achieved"):
                    rf["calibrated_peak"] = cal["copy_gb_per_s"]
                    rf["calibrated_frac"] = round(rf["achieved"] / cal["copy_gb_per_s"], 4)
        except Exception as e_:      # noqa: BLE001 -- a calibration that cannot run leaves the spec-based fractions as they are
            out["hbm_calibration"] = {"error": "%s: %s" % (type(e_).__name__, e_)}

    def assemble(optional_rows_note=None):
        o = copy.deepcopy(out)
        if multi and scaling_particles:
            o["scaling_detail"] = scaling_detail(world, frames=(copy.deepcopy(frames_scaling) if not args.no_lighting else {}),
                                                 collective_rows=copy.deepcopy(collective_rows), **scaling_particles)
            o["scaling_detail"]["note"] = ("the contract's top-level \"scaling\" stays the string \"weak\" (fixed work per GPU in the headline row); "
                                           "this block carries the named ratios")
            if optional_rows_note:
                o["scaling_detail"]["optional_rows"] = optional_rows_note
        return finalize_record(o, world, forced_dist, step_ms_gpu * 1e3, copy.deepcopy(collective_rows))

    def emit(o):
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        if rank == 0:
            print(json.dumps(o), flush=True)
        os.dup2(2, 1)

    particle_rows = None
    if multi and scaling_particles and not args.no_particle_collective_rows:
        import types as types_
        particle_rows = (types_.SimpleNamespace(H=H, native=native, abi=abi, scenes=scenes, args=args, ctx=ctx, group=group, ranks=ranks, rank=rank, world=world,
                                                local_rank=local_rank, k=args.steps, kernel=stream_kernel, traffic=stream_traffic), collective_rows)
    if args.no_lighting:
        deferred_rows, frames_scaling, kept_alive = [], {}, []
    if deferred_rows or particle_rows:
        # every figure above is in hand: the record as it stands is the fallback a watchdog prints if an optional row hangs
        fallback = assemble("the rows with particle collectives / pipelined_exchange / store_mode did not finish within %d s: the record was printed without "
                            "(some of) them" % args.optional_rows_timeout)
        run_optional_rows(deferred_rows, frames_scaling, args.optional_rows_timeout, rank, lambda: emit(fallback), ranks.barrier,
                          particle_rows=particle_rows, hang_rank=(rank == world - 1))
        deferred_rows.clear()
        while kept_alive:
            L_, r_, g_ = kept_alive.pop()
            del L_, r_
            g_.close()
    if args.dry_collectives:
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        if rank == 0:
            print("# bench.py --gpus %d --dry-collectives: every collective of the N > 1 branch in issue order (rank 0's view; %d rank(s)%s)"
                  % (n_gpus, world, ", ILM_BENCH_ONE_GPU stand-in" if one_gpu_stand_in else ""))
            print("\n".join(COLLECTIVES.table()), flush=True)
        os.dup2(2, 1)
    else:
        emit(assemble())
    if group is not None:
        import gc
        del ctx
        gc.collect()
        try:
            group.close()
        except native.IlluminantError as e:      # something of the member context is still referenced: report, the numbers stand
            print("bench.py: %s" % e, file=sys.stderr)


if __name__ == "__main__":
    main()
