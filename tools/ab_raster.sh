#!/bin/bash
# Run ON THE GPU BOX: the rasteriser row of bench.py with variant builds (tools/ab/<tag>/libilluminant_hip.so).  tools/ab_raster.sh base rw8 ...
cd "$(cd "$(dirname "$0")/.." && pwd)"
for round in 1 2; do
for tag in "$@"; do
  LD_LIBRARY_PATH=$PWD/tools/ab/$tag:${LD_LIBRARY_PATH:-} python bench.py --no-cpu-baseline --no-cfg4 --no-lighting --steps 200 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['next_rows']['rasterize_cfg2_1080p']['ms_per_frame'])"
done
done
