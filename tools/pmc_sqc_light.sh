#!/bin/bash
# Run ON THE GPU BOX: instruction-cache / scalar-cache counters of the light pass (bench lighting rows).
#   tools/pmc_sqc_light.sh <tag> [library dir]      -> gpurun_out/pmc_sqc_light_<tag>.txt
set -u
TAG=${1:-x}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
[ -n "${2:-}" ] && export LD_LIBRARY_PATH=$ROOT/$2:${LD_LIBRARY_PATH:-}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_sqc_light_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python bench.py --steps 20 --warmup 5 --blocks 3 --light-frames 2 --light-ms 0 --no-cpu-baseline --no-cfg4 --no-next-rows"
i=0
for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE" \
            "SQC_TC_REQ SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_STALL" "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $pass -d "$OUT/p$i" -o pmc -- $CMD > "$OUT/p$i.json" 2> "$OUT/p$i.log" || echo "pass $i failed"
done
python - "$OUT" <<'PY' > gpurun_out/pmc_sqc_light_$TAG.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "sphere_lights_kernel" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for name in sorted(cs):
        print("  %-30s %14.1f   (n=%d)" % (name, sum(cs[name]) / len(cs[name]), len(cs[name])))
PY
cat gpurun_out/pmc_sqc_light_$TAG.txt
