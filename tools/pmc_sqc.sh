#!/bin/bash
# Run ON THE GPU BOX: instruction-cache / scalar-cache counters of the particle step kernels (bench's cfg2 loop).
#   tools/pmc_sqc.sh <tag> [library dir]      -> gpurun_out/pmc_sqc_<tag>.txt
set -u
TAG=${1:-x}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
[ -n "${2:-}" ] && export LD_LIBRARY_PATH=$ROOT/$2:${LD_LIBRARY_PATH:-}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_sqc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python bench.py --steps 40 --warmup 5 --blocks 3 --no-cpu-baseline --no-lighting --no-cfg4 --no-next-rows"
i=0
for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE" \
            "SQC_TC_REQ SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_STALL" "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $pass -d "$OUT/p$i" -o pmc -- $CMD > "$OUT/p$i.json" 2> "$OUT/p$i.log" || echo "pass $i failed"
done
python - "$OUT" <<'PY' > gpurun_out/pmc_sqc_$TAG.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "step_kernel" not in k and "step_lean_kernel" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for name in sorted(cs):
        print("  %-30s %14.1f   (n=%d)" % (name, sum(cs[name]) / len(cs[name]), len(cs[name])))
PY
cat gpurun_out/pmc_sqc_$TAG.txt
