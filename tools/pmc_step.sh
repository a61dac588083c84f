#!/bin/bash
# Run ON THE GPU BOX: extra SQ counter passes for the particle step kernel only (instruction mix, issue activity, memory wait).
#   tools/pmc_step.sh <tag>      -> gpurun_out/pmc_step_<tag>.txt
set -u
TAG=${1:-x}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_step_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-lighting --no-cfg4 --no-next-rows"
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU" \
            "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LEVEL_WAVES SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_WAVES SQ_BUSY_CYCLES" \
            "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $pass -d "$OUT/p$i" -o pmc -- $CMD > "$OUT/p$i.json" 2> "$OUT/p$i.log" || echo "pass $i failed"
done
python - "$OUT" <<'PY' > gpurun_out/pmc_step_$TAG.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "step_kernel" not in k and "step_lean_kernel" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    waves = None
    for name in sorted(cs):
        v = sum(cs[name]) / len(cs[name])
        print("  %-28s %14.1f   (n=%d)" % (name, v, len(cs[name])))
PY
cat gpurun_out/pmc_step_$TAG.txt
