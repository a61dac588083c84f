#!/bin/bash
# Run ON THE GPU BOX: instruction counters of tools/step_mix_probe.py, averaged per group of 5 dispatches.
set -u
cd "$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_mix_${1:-x}
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES -d "$OUT/p" -o pmc -- python tools/step_mix_probe.py > "$OUT/log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
rows = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_kernel" not in r["Kernel_Name"] and "step_lean" not in r["Kernel_Name"]: continue
        rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
names = ["update only", "gravity", "noise", "gravity+noise", "gravity+noise, no update"]
for g in range(len(ids) // 5):
    grp = [rows[i] for i in ids[g * 5:(g + 1) * 5]]
    w = sum(x["SQ_WAVES"] for x in grp) / 5
    print("%-26s waves %6d  per wave:" % (names[g] if g < len(names) else g, w),
          "  ".join("%s %.1f" % (k.replace("SQ_INSTS_", "").replace("SQ_", ""), sum(x[k] for x in grp) / 5 / w) for k in
                    ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES")))
PY
