#!/bin/bash
# Run ON THE GPU BOX: kernel durations (rocprofv3 --kernel-trace, no counters) of tools/step_mix_probe.py per group of 5 dispatches.
set -u
cd "$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT=gpurun_out/time_mix_${1:-x}
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/p" -o t -- python tools/step_mix_probe.py > "$OUT/log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/p/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Kernel_Name"] or "step_lean" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
names = ["update only", "gravity", "noise", "gravity+noise", "gravity+noise, no update"]
for g in range(len(rows) // 5):
    grp = rows[g * 5:(g + 1) * 5]
    d = [(e - s) / 1000.0 for s, e in grp]
    gaps = [(grp[i + 1][0] - grp[i][1]) / 1000.0 for i in range(4)]
    print("%-26s kernel us: %s   gaps us: %s" % (names[g] if g < len(names) else g, " ".join("%.1f" % x for x in d), " ".join("%.1f" % x for x in gaps)))
PY
