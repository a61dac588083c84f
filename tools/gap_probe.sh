#!/bin/bash
# Run ON THE GPU BOX: kernel durations and the gaps between consecutive step launches of the bench's cfg2 loop (rocprofv3 --kernel-trace).
set -u
cd "$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT=gpurun_out/gap_${1:-x}
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/p" -o t -- python bench.py --steps 40 --warmup 5 --blocks 3 --no-cpu-baseline --no-lighting --no-cfg4 --no-next-rows > "$OUT/bench.json" 2> "$OUT/log"
python - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/p/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
steps = [(s, e) for s, e, k in rows if "step_" in k]
steps = steps[-100:]
d = sorted((e - s) / 1000.0 for s, e in steps)
gaps = [(steps[i + 1][0] - steps[i][1]) / 1000.0 for i in range(len(steps) - 1)]
print("last %d step kernels: duration median %.2f us (min %.2f max %.2f)" % (len(d), d[len(d) // 2], d[0], d[-1]))
print("gaps between consecutive step kernels, in order:", " ".join("%.1f" % g for g in gaps[-40:]))
gs = sorted(g for g in gaps if g < 100)
print("gap median %.2f mean %.2f" % (gs[len(gs) // 2], sum(gs) / len(gs)))
PY
