#!/bin/bash
# Run ON THE GPU BOX: the first <count> seeds of each of sweep r06b's 24 processes again (same seeds, same process layout): are its failures
# a function of (process, seed), of the load, or of the box?   tools/fuzz_repro.sh <count>
cd "$(cd "$(dirname "$0")/.." && pwd)"
count=${1:-900}
mkdir -p gpurun_out/repro; rm -f gpurun_out/repro/*
for i in $(seq 0 23); do
  OMP_NUM_THREADS=1 python tools/fuzz_parity.py $((6500000 + i * 2500)) $count > gpurun_out/repro/p$i.txt 2>&1 &
done
wait
echo "failing (seed) lines:"; grep -h "^    (65" gpurun_out/repro/p*.txt | awk -F'[(,]' '{print $2}' | sort | uniq -c
echo "processes that passed: $(grep -l "FUZZ PASSED" gpurun_out/repro/p*.txt | wc -l) of 24"
