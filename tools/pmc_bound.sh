#!/bin/bash
# Run ON THE GPU BOX: which resource binds the cone trace?  Texture-addresser (TA), L1 (TCP), texture-data (TD) and L2 (TCC)
# busy / stall counters of the lighting kernels next to the SQ issue counters, each group in its own pass (counters of one
# hardware block share few slots).  tools/pmc_bound.sh <tag>  ->  gpurun_out/pmc_bound_<tag>/summary.txt
set -u
cd "$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_bound_${1:-x}
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python bench.py --steps 20 --warmup 2 --light-frames 1 --light-ms 0 --no-cpu-baseline --no-cfg4 --no-next-rows"
i=0
while read -r pass; do
  [ -z "$pass" ] && continue
  i=$((i + 1))
  rocprofv3 --kernel-trace --output-format csv --pmc $pass -d "$OUT/p$i" -o pmc -- $CMD > "$OUT/p$i.json" 2> "$OUT/p$i.log" || echo "pass $i ($pass) failed" >> "$OUT/errors.txt"
done <<'EOF'
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL
TA_TA_BUSY_sum TA_BUSY_avr
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
TA_TOTAL_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum
TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum
TD_TD_BUSY_sum TD_TC_STALL_sum
TD_SPI_STALL_sum TD_LOAD_WAVEFRONT_sum
TCC_REQ_sum TCC_BUSY_sum TCC_TAG_STALL_sum TCC_HIT_sum
EOF
python - "$OUT" <<'PY' > "$OUT/summary.txt"
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "sphere_lights" not in k and "step_kernel" not in k and "step_lean" not in k and "render_slices" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(sys.argv[1] + "/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if k in acc: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in sorted(acc.items()):
    print(k)
    if dur[k]: print("   launches %d  avg %.1f us (under the first counter pass)" % (len(dur[k]), sum(dur[k]) / len(dur[k])))
    for n, v in sorted(cs.items()):
        print("   %-40s %16.1f  (n=%d)" % (n, sum(v) / len(v), len(v)))
PY
cat "$OUT/errors.txt" 2>/dev/null
tail -n 200 "$OUT/summary.txt"
