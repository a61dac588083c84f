#!/bin/bash
# Run ON THE GPU BOX: the lighting rows of bench.py with variant builds of libilluminant_hip.so (tools/ab/<tag>/libilluminant_hip.so, made by
# the caller; the host mirror finds the library through RUNPATH, so LD_LIBRARY_PATH picks the variant).  tools/ab_lib.sh base w6 w7 ...
cd "$(cd "$(dirname "$0")/.." && pwd)"
for round in 1 2; do
for tag in "$@"; do
  ILM_HIP_LIB=$PWD/tools/ab/$tag/libilluminant_hip.so LD_LIBRARY_PATH=$PWD/tools/ab/$tag:${LD_LIBRARY_PATH:-} python bench.py --no-cpu-baseline --no-cfg4 --no-next-rows --steps 20 --light-frames 8 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', {k:v['roofline']['launch_ms'] for k,v in d['lighting'].items()}, 'step', d['roofline']['launch_ms'])"
done
done
