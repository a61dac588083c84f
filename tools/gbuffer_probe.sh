#!/bin/bash
# Run ON THE GPU BOX: tools/gbuffer_probe.py alone and under rocprofv3 --kernel-trace --memory-copy-trace (the kernels' and the upload's durations).
cd "$(cd "$(dirname "$0")/.." && pwd)"
python tools/gbuffer_probe.py 2>&1 | grep Vector
export TMPDIR=/tmp
rm -rf /tmp/gp
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/gp -o gp -- python tools/gbuffer_probe.py --frames 50 > /tmp/gp.log 2>&1
for f in /tmp/gp/gp_kernel_stats.csv /tmp/gp/gp_memory_copy_stats.csv; do [ -f $f ] && cut -d, -f1-4,6,7 $f | grep -v fillBuffer; done
