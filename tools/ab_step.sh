#!/bin/bash
# Run ON THE GPU BOX: the particle step with variant builds of libilluminant_hip.so (tools/ab/<tag>/libilluminant_hip.so, made by the
# caller), interleaved on the same box: cfg2-shaped 16 x 256^2 chunks with and without the spawner, cfg4-shaped 8 x 1024^2.
#   tools/ab_step.sh base v1 ...
cd "$(cd "$(dirname "$0")/.." && pwd)"
for round in 1 2 3; do
for tag in "$@"; do
  ILM_HIP_LIB=tools/ab/$tag/libilluminant_hip.so python tools/perf_probe.py p 2>&1 | grep particles | sed "s/^particles/$tag/" | sed -E 's/ops=.gn. update=1 //; s/\([0-9.]+ wall\)//; s/Mslot-steps.*bytes//'
done
done
