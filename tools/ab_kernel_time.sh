#!/bin/bash
# Run ON THE GPU BOX: average duration of the kernels whose name contains <pattern> over bench.py's next_rows, per library variant
# (rocprofv3 --kernel-trace --stats).      tools/ab_kernel_time.sh <pattern> base v1 ...
cd "$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
pat=$1; shift
for tag in "$@"; do
  out=gpurun_out/abk_$tag; rm -rf $out
  ILM_HIP_LIB=$PWD/tools/ab/$tag/libilluminant_hip.so LD_LIBRARY_PATH=$PWD/tools/ab/$tag:${LD_LIBRARY_PATH:-} rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- \
    python bench.py --no-cpu-baseline --no-cfg4 --light-frames 1 --light-ms 0 --steps 20 --warmup 5 > /dev/null 2>&1
  grep -h "$pat" $(find $out -name "*kernel_stats.csv") | awk -F, -v t=$tag '{printf "%-8s %s calls %s avg %.1f us\n", t, $1, $2, $4/1000}' | cut -c1-150
done
