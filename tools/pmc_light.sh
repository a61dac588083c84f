#!/bin/bash
# Run ON THE GPU BOX: lane utilisation of the lighting kernels (SQ_THREAD_CYCLES_VALU / (4 * SQ_ACTIVE_INST_VALU) of 16 lanes per cycle).
set -u
cd "$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_light_${1:-x}
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES -d "$OUT/p" -o pmc -- python bench.py --steps 20 --warmup 2 --light-frames 1 --light-ms 0 --no-cpu-baseline --no-cfg4 --no-next-rows > "$OUT/log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "sphere_lights" not in k and "render_slices" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    m = {n: sum(v) / len(v) for n, v in cs.items()}
    print(k)
    print("   waves %d  VALU/wave %.0f  SALU/wave %.0f  VMEM_RD/wave %.0f  lanes active per VALU cycle %.2f of 16  wave quad-cycles %.0f" % (
        m["SQ_WAVES"], m["SQ_INSTS_VALU"] / m["SQ_WAVES"], m["SQ_INSTS_SALU"] / m["SQ_WAVES"], m["SQ_INSTS_VMEM_RD"] / m["SQ_WAVES"],
        m["SQ_THREAD_CYCLES_VALU"] / (4 * m["SQ_ACTIVE_INST_VALU"]), m["SQ_WAVE_CYCLES"] / m["SQ_WAVES"]))
PY
