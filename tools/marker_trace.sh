#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 --marker-trace --kernel-trace of a short bench.py run with ILM_TRACE=1 -- the roctx ranges of the C ABI's
# entry points (api.hip trace_api) next to the kernels they queue.  No --pmc in the same run (gpurun refuses that combination).
#   tools/marker_trace.sh <tag>      -> gpurun_out/profiles_<tag>/<tag>_marker_trace.txt
set -u
TAG=${1:-r05}
cd "$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT=gpurun_out/marker_$TAG
S=gpurun_out/profiles_$TAG
rm -rf "$OUT"; mkdir -p "$OUT" "$S"
ILM_TRACE=1 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- \
  python bench.py --steps 5 --warmup 2 --blocks 3 --light-frames 2 --light-ms 0 --sustain-s 0 --no-cpu-baseline --no-cfg4 > "$OUT/bench.json" 2> "$OUT/bench.err"
python - "$OUT" "$S/${TAG}_marker_trace.txt" <<'PY'
import csv, glob, sys, collections
out, dst = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(out + "/**/*marker_api_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
names = collections.Counter()
dur = collections.defaultdict(float)
for r in rows:
    n = r.get("Function") or r.get("Name") or r.get("Message") or "?"
    names[n] += 1
    try:
        dur[n] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3
    except (KeyError, ValueError):
        pass
with open(dst, "w") as f:
    f.write("# ILM_TRACE=1 rocprofv3 --marker-trace --kernel-trace -- python bench.py --steps 5 --warmup 2 --blocks 3 --light-frames 2 ... (tools/marker_trace.sh)\n")
    f.write("# roctx ranges pushed by the C ABI's entry points: calls, host microseconds inside the range (total, mean)\n")
    for n, c in names.most_common():
        f.write("%-40s %7d %12.1f %10.2f\n" % (n, c, dur[n], dur[n] / max(c, 1)))
    f.write("# %d marker records in %d file(s)\n" % (len(rows), len(glob.glob(out + "/**/*marker_api_trace.csv", recursive=True))))
    ks = glob.glob(out + "/**/*kernel_stats.csv", recursive=True)
    if ks:
        f.write("# kernels of the same run (rocprofv3 --stats), top 12 by total time\n")
        for r in list(csv.DictReader(open(ks[0])))[:12]:
            f.write("%-90s calls %6s total_ns %12s\n" % (r.get("Name", "?")[:90], r.get("Calls", "?"), r.get("TotalDurationNs", "?")))
print(open(dst).read())
PY
