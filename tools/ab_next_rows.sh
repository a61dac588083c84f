#!/bin/bash
# Run ON THE GPU BOX: the next_rows of bench.py (read-back, rasteriser, particle lights, resolve, 2.5D G-buffer, collision step) with variant
# builds of the library (tools/ab/<tag>/).  tools/ab_next_rows.sh base v1 ...
cd "$(cd "$(dirname "$0")/.." && pwd)"
for round in 1 2; do
for tag in "$@"; do
  ILM_HIP_LIB=$PWD/tools/ab/$tag/libilluminant_hip.so LD_LIBRARY_PATH=$PWD/tools/ab/$tag:${LD_LIBRARY_PATH:-} python bench.py --no-cpu-baseline --no-cfg4 --steps 20 --warmup 5 --sustain-s 0.2 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-8s' % '$tag', {k: (v.get('ms_per_frame') or v.get('us_per_step') or v.get('ms_per_readback_incl_pcie')) for k, v in d['next_rows'].items()})"
done
done
