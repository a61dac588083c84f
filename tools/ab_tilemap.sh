#!/bin/bash
# Run ON THE GPU BOX: the lighting rows of bench.py under block -> tile mappings (ILM_LIGHT_TILE_MAP[:ILM_LIGHT_TILE_MACRO]).  tools/ab_tilemap.sh 2 4:2 4:4 ...
cd "$(cd "$(dirname "$0")/.." && pwd)"
for round in 1 2; do
for spec in "$@"; do
  tm=${spec%%:*}; macro=${spec##*:}
  ILM_LIGHT_TILE_MAP=$tm ILM_LIGHT_TILE_MACRO=$macro python bench.py --no-cpu-baseline --no-cfg4 --no-next-rows --steps 20 --light-frames 8 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tile_map %-5s' % '$spec', {k:v['roofline']['launch_ms'] for k,v in d['lighting'].items()})"
done
done
