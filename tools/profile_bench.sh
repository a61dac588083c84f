#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace statistics and PMC counters of the bench command.
#   tools/profile_bench.sh <tag>            e.g. r01
# Writes raw rocprofv3 output under gpurun_out/prof_<tag>/ and the judged summaries under gpurun_out/profiles_<tag>/
# (copy those into profiles/ and commit them).  Counters are collected in their own passes, each with
# --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md, "rocprofv3 PMC slots").
set -u
TAG=${1:-r01}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
SUM=gpurun_out/profiles_$TAG
rm -rf "$OUT" "$SUM"; mkdir -p "$OUT" "$SUM"
# the driver's exact command (kernel durations do not carry over between --steps 200 and --steps 20)
BENCH="python bench.py --steps 20 --warmup 5"
SHORT="python bench.py --steps 40 --warmup 5 --light-frames 1 --light-ms 0 --no-cpu-baseline"

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- $BENCH > "$SUM/bench_under_rocprof.json" 2> "$OUT/stats.log"
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
            "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_LEVEL_VMEM"; do
  name=$(echo "$pass" | tr ' ' '+')
  rocprofv3 --kernel-trace --output-format csv --pmc $pass -d "$OUT/pmc_$name" -o pmc -- $SHORT > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.log" || echo "pass $name failed" >> "$SUM/errors.txt"
done
python tools/summarize_prof.py "$OUT" "$SUM" "$TAG"
# un-profiled reference line (never compare a profiled arm with an un-profiled one: the clocks differ), taken AFTER the counters so
# that the record's profile-derived fractions come from THIS build's profile: bench.py reads profiles/<tag>_pmc.csv and refuses one
# whose kernel-source hash is not the tree's
cp "$SUM/${TAG}_pmc.csv" "$SUM/${TAG}_pmc.meta.json" profiles/ 2>/dev/null
$BENCH > "$SUM/bench_unprofiled.json" 2> "$OUT/bench_unprofiled.err"
ls -la "$SUM"
