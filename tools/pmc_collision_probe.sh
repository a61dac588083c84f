#!/bin/bash
# Run ON THE GPU BOX: counters of the collision step's kernels (interpreter and step_lean_df_kernel, 1 / 2 / 4 units per wave) under
# tools/collision_probe.py.     tools/pmc_collision_probe.sh <tag> [sizes]   ->  gpurun_out/pmc_collision_probe_<tag>/summary.txt
set -u
cd "$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_collision_probe_${1:-x}
SIZES=${2:-1m}
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python tools/collision_probe.py --sizes $SIZES --reps 2 --steps 10"
timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d "$OUT/p1" -o pmc -- $CMD > "$OUT/p1.txt" 2> "$OUT/p1.log"
timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_WAIT_ANY -d "$OUT/p2" -o pmc -- $CMD > "$OUT/p2.txt" 2> "$OUT/p2.log"
timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$OUT/p3" -o pmc -- $CMD > "$OUT/p3.txt" 2> "$OUT/p3.log"
python - "$OUT" <<'PY' > "$OUT/summary.txt" 2>&1
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("p1", "p2", "p3"):
    for f in glob.glob(sys.argv[1] + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "step_kernel" not in k and "step_lean" not in k: continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    m = {n: sorted(v)[len(v) // 2] for n, v in cs.items()}
    w = m.get("SQ_WAVES", 0)
    if w < 1000: continue
    g = lambda n: m.get(n, 0.0)
    print(k)
    print("   dispatches %d  waves %d  VALU/wave %.0f  SALU/wave %.0f  VMEM_RD/wave %.1f  VMEM_WR/wave %.1f  LDS/wave %.1f  lanes per VALU instruction %.1f of 64  wave quad-cycles %.0f  waiting %.0f %%" % (
        len(cs["SQ_WAVES"]), w, g("SQ_INSTS_VALU") / w, g("SQ_INSTS_SALU") / w, g("SQ_INSTS_VMEM_RD") / w, g("SQ_INSTS_VMEM_WR") / w, g("SQ_INSTS_LDS") / w,
        g("SQ_THREAD_CYCLES_VALU") / max(g("SQ_ACTIVE_INST_VALU"), 1), g("SQ_WAVE_CYCLES") / w, 100.0 * g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1)))
    print("   HBM: FETCH_SIZE x 2 %.1f MB + WRITE_SIZE %.1f MB per dispatch = %.1f B per slot   L2 hit rate %.1f %%" % (
        g("FETCH_SIZE") * 2 / 1024, g("WRITE_SIZE") / 1024, (g("FETCH_SIZE") * 2 + g("WRITE_SIZE")) * 1024 / max(w * 64 * (2 if "2>(" in k else (4 if "4>(" in k else 1)), 1), 100.0 * g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1)))
PY
cat "$OUT/summary.txt"; cat "$OUT/p1.txt"
