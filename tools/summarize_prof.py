#!/usr/bin/env python3
"""Condense rocprofv3 output (tools/profile_bench.sh) into the small files committed under profiles/.

    summarize_prof.py <raw dir> <summary dir> <tag>

  <tag>_kernel_stats.csv   the rocprofv3 --stats per-kernel table (calls, total / average / min / max ns)
  <tag>_pmc.csv            per kernel: every collected counter of the kernel's MEDIAN dispatch
  <tag>_summary.md         the few numbers bench.py's roofline object is checked against
"""
import collections
import csv
import glob
import json
import os
import sys


def find(root, suffix):
    hits = sorted(glob.glob(os.path.join(root, "**", "*" + suffix), recursive=True))
    return hits[0] if hits else None


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def main():
    raw, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs(out, exist_ok=True)
    lines = ["# rocprofv3 summary `%s`" % tag, ""]

    stats = find(os.path.join(raw, "stats"), "kernel_stats.csv")
    kernel_avg = {}
    if stats:
        rows = list(csv.DictReader(open(stats)))
        with open(os.path.join(out, "%s_kernel_stats.csv" % tag), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
                kernel_avg[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]))
        lines += ["## kernel trace (`rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5`, the driver's exact command; PMC passes: `--steps 40 --warmup 5 --light-frames 1 --light-ms 0 --no-cpu-baseline`)", "",
                  "| kernel | calls | average us | % of GPU time |", "|---|---|---|---|"]
        for r in rows[:8]:
            lines.append("| `%s` | %s | %.2f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
        lines.append("")
        # The light pass is also launched two at a time (bench.py's two_frames_in_flight row: alternate frames on sibling contexts): a
        # launch that shares the device with another light-pass launch takes about twice as long and says nothing about the kernel.  The
        # averages bench.py's ms_per_frame / roofline_lighting are to be checked against are those of the launches that ran ALONE.
        trace = find(os.path.join(raw, "stats"), "kernel_trace.csv")
        if trace:
            light = []
            for r in csv.DictReader(open(trace)):
                if "sphere_lights_kernel" in r["Kernel_Name"]:
                    light.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Grid_Size_X"])))
            light.sort()
            alone, shared = collections.defaultdict(list), collections.defaultdict(list)
            for i, (b, e, name, grid) in enumerate(light):
                overlap = 0
                for j in range(max(0, i - 4), min(len(light), i + 5)):
                    if j != i:
                        overlap = max(overlap, min(e, light[j][1]) - max(b, light[j][0]))
                (shared if overlap > 0.05 * (e - b) else alone)[(name, grid)].append(e - b)
            lines += ["### light-pass launches, alone on the device vs sharing it with another light-pass launch (frames in flight)", "",
                      "| kernel | grid (threads) | alone: calls | alone: average us | alone: median us | overlapped: calls | overlapped: average us |", "|---|---|---|---|---|---|---|"]
            for key in sorted(set(alone) | set(shared), key=lambda k: -sum(alone.get(k, [])) - sum(shared.get(k, []))):
                a, o = sorted(alone.get(key, [])), shared.get(key, [])
                if len(a) + len(o) < 8:
                    continue
                lines.append("| `%s` | %d | %d | %s | %s | %d | %s |" % (key[0], key[1], len(a), ("%.2f" % (sum(a) / len(a) / 1e3)) if a else "-",
                                                                 ("%.2f" % (a[len(a) // 2] / 1e3)) if a else "-", len(o), ("%.2f" % (sum(o) / len(o) / 1e3)) if o else "-"))
            lines.append("")

        # The step kernels by grid: one kernel name carries cfg2's steps, cfg4's share, cfg4 whole (the N = 1 headline since r06) and cfg5's
        # 16 M particles; a step is TWO launches (the chunk range halved over the context's two streams) that run side by side, so a
        # launch lasts about as long as the step it belongs to.  bench.py's roofline.launch_ms is to be checked against the row of its grid.
        if trace:
            steps_by_grid = collections.defaultdict(list)
            for r in csv.DictReader(open(trace)):
                if "step_lean" in r["Kernel_Name"] or "step_kernel" in r["Kernel_Name"]:
                    steps_by_grid[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            lines += ["### step launches by grid (a step = two launches side by side on the context's two streams)", "",
                      "| kernel | grid (threads) | slots per launch | calls | average us | median us |", "|---|---|---|---|---|---|"]
            for key in sorted(steps_by_grid, key=lambda k: -sum(steps_by_grid[k])):
                v = sorted(steps_by_grid[key])
                if len(v) < 8:
                    continue
                lines.append("| `%s` | %d | %d | %d | %.2f | %.2f |" % (key[0], key[1], key[1], len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3))
            lines.append("")

    # PMC passes
    # Per kernel and counter the MEDIAN over its dispatches: a kernel name can carry launches of different scenes (the 4K kernel also
    # runs one small frame in the bench's exchange check) and a mean over them describes none of them.
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    agg = collections.defaultdict(dict)
    meta = {}
    for path in sorted(glob.glob(os.path.join(raw, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"])
    for k in vals:
        for cname, v in vals[k].items():
            v = sorted(v)
            agg[k][cname] = [len(v), v[len(v) // 2] * len(v)]      # [dispatches, median x dispatches]: the code below divides
    if agg:
        counters = sorted({c for k in agg for c in agg[k]})
        with open(os.path.join(out, "%s_pmc.csv" % tag), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "dispatches", "vgpr", "sgpr", "lds", "workgroup"] + counters)
            for k in sorted(agg):
                n = max(v[0] for v in agg[k].values())
                w.writerow([k, n] + list(meta[k]) + ["%.1f" % (agg[k][c][1] / agg[k][c][0]) if c in agg[k] else "" for c in counters])
        # what the counters are counters OF: bench.py uses the profile only while the tree's kernel sources hash to this
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        with open(os.path.join(out, "%s_pmc.meta.json" % tag), "w") as f:
            json.dump({"kernel_sources_sha256": bench.kernel_sources_sha256(), "tag": tag}, f)
        lines += ["## PMC counters (median dispatch of each kernel; separate passes, `--kernel-trace --pmc <set>`)", ""]
        for k in sorted(agg):
            if "ilm::" not in k:
                continue
            c = {name: v[1] / v[0] for name, v in agg[k].items()}
            lines.append("### `%s`  (VGPR %s, SGPR %s)" % (k, meta[k][0], meta[k][1]))
            if "FETCH_SIZE" in c:
                if "raster_tiles_kernel" in k:
                    # 64-byte sprite records gathered by index are tallied at their size, the 8-byte keys at half (profiles/r06_fetch_size_calibration.txt)
                    lines.append("* FETCH_SIZE %.1f KB per dispatch (raw) = %.2f MB; this kernel gathers 64-byte records (tallied x 1) and streams 8-byte keys "
                                 "(x 1/2): NOT doubled (profiles/r06_fetch_size_calibration.txt; r03-r05 doubled it and read 2.06 x the records)" % (c["FETCH_SIZE"], c["FETCH_SIZE"] * 1024 / 1e6))
                else:
                    lines.append("* FETCH_SIZE %.1f KB per dispatch (raw); x2 for wide coalesced reads per MI355X_MICROARCH.md = %.2f MB" % (c["FETCH_SIZE"], c["FETCH_SIZE"] * 2 * 1024 / 1e6))
            if "WRITE_SIZE" in c:
                lines.append("* WRITE_SIZE %.1f KB per dispatch = %.2f MB" % (c["WRITE_SIZE"], c["WRITE_SIZE"] * 1024 / 1e6))
            if "TCC_HIT_sum" in c and (c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0)) > 0:
                lines.append("* L2 hit rate %.1f %% (TCC_HIT %.0f, TCC_MISS %.0f)" % (100 * c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), c["TCC_HIT_sum"], c["TCC_MISS_sum"]))
            if "SQ_WAVES" in c:
                w_ = max(c["SQ_WAVES"], 1)
                lines.append("* waves %.0f; per wave: VALU %.0f, SALU %.0f, SMEM %.0f; VALU busy %.0f quad-cycles per wave; wave cycles %.0f" %
                             (c["SQ_WAVES"], c.get("SQ_INSTS_VALU", 0) / w_, c.get("SQ_INSTS_SALU", 0) / w_, c.get("SQ_INSTS_SMEM", 0) / w_,
                              c.get("SQ_ACTIVE_INST_VALU", 0) / w_, c.get("SQ_WAVE_CYCLES", 0) / w_))
            if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
                lines.append("* lanes active per vector instruction: %.1f of 64 (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU)" %
                             (c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]))
            if "SQ_WAIT_ANY" in c and "SQ_ACTIVE_INST_ANY" in c:
                lines.append("* SQ_WAIT_ANY %.3g, SQ_WAIT_INST_ANY %.3g, SQ_ACTIVE_INST_ANY %.3g (quad-cycles summed over waves)" %
                             (c["SQ_WAIT_ANY"], c["SQ_WAIT_INST_ANY"], c["SQ_ACTIVE_INST_ANY"]))
            lines.append("")
    for name in ("bench_unprofiled.json", "bench_under_rocprof.json"):
        p = os.path.join(out, name)
        if os.path.exists(p):
            try:
                d = json.loads(open(p).read().strip().splitlines()[-1])
                lines.append("* `%s`: value %.1f %s, ms_per_step %.5f, roofline.frac %.4f (launch_ms %.5f)" %
                             (name, d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_ms"]))
            except Exception as e:  # noqa: BLE001
                lines.append("* `%s`: unreadable (%s)" % (name, e))
    open(os.path.join(out, "%s_summary.md" % tag), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
