#!/bin/bash
# Run ON THE GPU BOX: tools/ubench/valu alone (issue rate), then under the SQ counters so that the counters' units can be read off a kernel
# whose residency (W waves per SIMD for the whole launch) and instruction count are known by construction.
set -u
cd "$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_valu_${1:-x}
rm -rf "$OUT"; mkdir -p "$OUT"
tools/ubench/valu | tee "$OUT/valu.txt"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d "$OUT/p" -o pmc -- tools/ubench/valu > "$OUT/log" 2>&1
python - "$OUT" <<'PY' | tee "$OUT/units.txt"
import csv, glob, sys, collections
rows = collections.OrderedDict(); dur = {}
for f in glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for f in glob.glob(sys.argv[1] + "/p/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
for i in sorted(rows):
    x = rows[i]; t = dur.get(i, 0)
    w = x["SQ_WAVES"]
    print("dispatch %2d waves %6d  %.3f ms  VALU/wave %.0f  WAVE_CYCLES/wave %.0f  ACTIVE_INST_VALU/wave %.0f  BUSY_CYCLES %.3g  GRBM_GUI_ACTIVE %.3g  (GUI_ACTIVE/8/t = %.2f GHz)"
          % (i, w, t * 1e3, x["SQ_INSTS_VALU"] / w, x["SQ_WAVE_CYCLES"] / w, x["SQ_ACTIVE_INST_VALU"] / w, x["SQ_BUSY_CYCLES"], x["GRBM_GUI_ACTIVE"],
             x["GRBM_GUI_ACTIVE"] / 8 / t / 1e9 if t else 0))
PY
