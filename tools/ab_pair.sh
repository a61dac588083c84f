#!/bin/bash
# Run ON THE GPU BOX: lighting rows with four dword tap loads (ILM_SDF_PAIR_LOADS=0) vs one 16-byte load per tap row (1, the default); same box, same build.
cd "$(cd "$(dirname "$0")/.." && pwd)"
for round in 1 2; do
for v in 0 1; do
  export ILM_SDF_PAIR_LOADS=$v
  python bench.py --no-cpu-baseline --no-cfg4 --steps 30 --light-frames 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); L=d['lighting']; print('pair_loads=$v', {k:v['roofline']['launch_ms'] for k,v in L.items()}, d['next_rows']['particle_lights_1080p_4096']['ms_per_frame'])"
done
done
