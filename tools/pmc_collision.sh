#!/bin/bash
# Run ON THE GPU BOX: counters of the collision update (bench.py collision_step_1m: ilm::step_kernel<unorm16, DF>) next to the plain step's
# kernels, and the plain step through the interpreter (ILM_STEP_LEAN=0) for the "would a lean variant help" question.
#   tools/pmc_collision.sh <tag>  ->  gpurun_out/pmc_collision_<tag>/summary.txt
set -u
cd "$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_collision_${1:-x}
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python bench.py --steps 20 --warmup 5 --no-lighting --no-cfg4 --no-cpu-baseline"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -d "$OUT/p1" -o pmc -- $CMD > "$OUT/p1.json" 2> "$OUT/p1.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- $CMD > "$OUT/stats.json" 2> "$OUT/stats.log"
{
python - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "step_kernel" not in k and "step_lean" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    m = {n: sorted(v)[len(v) // 2] for n, v in cs.items()}          # median dispatch
    if m.get("SQ_WAVES", 0) < 1000: continue
    print(k)
    print("   dispatches %d  waves %d  VALU/wave %.0f  SALU/wave %.0f  SMEM/wave %.0f  VMEM_RD/wave %.1f  lanes active per VALU instruction %.1f of 64  wave quad-cycles %.0f" % (
        len(cs["SQ_WAVES"]), m["SQ_WAVES"], m["SQ_INSTS_VALU"] / m["SQ_WAVES"], m["SQ_INSTS_SALU"] / m["SQ_WAVES"], m["SQ_INSTS_SMEM"] / m["SQ_WAVES"],
        m["SQ_INSTS_VMEM_RD"] / m["SQ_WAVES"], m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"], m["SQ_WAVE_CYCLES"] / m["SQ_WAVES"]))
PY
echo
echo "kernel-trace statistics of the same command (rocprofv3 --kernel-trace --stats):"
grep -h "step_kernel\|step_lean\|Name" "$OUT"/stats/*kernel_stats.csv | head -12
echo
echo "bench rows: default, then ILM_STEP_LEAN=0 (every step through the interpreter)"
for lean in 1 0; do ILM_STEP_LEAN=$lean $CMD 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['next_rows']['collision_step_1m']
print('ILM_STEP_LEAN=$lean  cfg2 step %.2f us   plain (no spawner) %.2f us   collision %.2f us   samples/particle %.3f' % (d['roofline']['launch_ms']*1e3, r['us_per_step_update_positions'], r['us_per_step'], r['sdf_samples_per_particle']))"; done
} > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
