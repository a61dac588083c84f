#!/bin/bash
# Run ON THE GPU BOX: kernel durations, start-to-start periods and overlaps of the step launches of the bench's cfg2 loop (rocprofv3 --kernel-trace).
#   tools/gap_probe.sh <tag> [library dir]
set -u
cd "$(cd "$(dirname "$0")/.." && pwd)"
[ -n "${2:-}" ] && export LD_LIBRARY_PATH=$PWD/$2:${LD_LIBRARY_PATH:-}
export TMPDIR=/tmp
OUT=gpurun_out/gap_${1:-x}
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/p" -o t -- python bench.py --steps 40 --warmup 5 --blocks 3 --no-cpu-baseline --no-lighting --no-cfg4 --no-next-rows > "$OUT/bench.json" 2> "$OUT/log"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/p/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
steps = [r for r in rows if "step_" in r[2]][-160:]
by = collections.defaultdict(list)
for s, e, k, q in steps: by[(k.split("(")[0][-40:], q)].append((e - s) / 1000.0)
for key, d in by.items():
    d.sort(); print("%-48s queue %s: n=%d duration median %.2f us (min %.2f max %.2f)" % (key[0], key[1], len(d), d[len(d) // 2], d[0], d[-1]))
t0 = steps[0][0]
print("timeline of the last launches (start, end in us since the first; queue):")
for s, e, k, q in steps[-12:]:
    print("  %9.2f %9.2f  q%s %s" % ((s - t0) / 1000.0, (e - t0) / 1000.0, q, k.split("(")[0][-28:]))
span = (steps[-1][1] - steps[0][0]) / 1000.0
print("%d launches in %.1f us" % (len(steps), span))
PY
