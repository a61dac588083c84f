#!/bin/bash
# Run ON THE GPU BOX: field generation time of bench.py's two scenes with variant builds of the library (tools/ab/<tag>/).  tools/ab_field.sh base v1 ...
cd "$(cd "$(dirname "$0")/.." && pwd)"
for round in 1 2 3; do
for tag in "$@"; do
  ILM_HIP_LIB=$PWD/tools/ab/$tag/libilluminant_hip.so LD_LIBRARY_PATH=$PWD/tools/ab/$tag:${LD_LIBRARY_PATH:-} python bench.py --no-cpu-baseline --no-cfg4 --no-next-rows --steps 5 --warmup 1 --blocks 3 --light-frames 1 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-12s' % '$tag', {k[:4]: v['field_generation']['ms_per_field'] for k,v in d['lighting'].items()})"
done
done
