#!/bin/bash
# Run ON THE GPU BOX: the collision row of bench.py with variant builds of the library (tools/ab/<tag>/).  tools/ab_collision.sh base v1 ...
cd "$(cd "$(dirname "$0")/.." && pwd)"
for round in 1 2; do
for tag in "$@"; do
  ILM_HIP_LIB=$PWD/tools/ab/$tag/libilluminant_hip.so LD_LIBRARY_PATH=$PWD/tools/ab/$tag:${LD_LIBRARY_PATH:-} python bench.py --no-cpu-baseline --no-cfg4 --no-lighting --steps 20 --warmup 5 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['next_rows']['collision_step_1m']; print('%-10s' % '$tag', 'collision %.2f us (min %.2f)  plain %.2f us' % (r['us_per_step'], r['us_per_step_min'], r['us_per_step_update_positions']))"
done
done
