#!/bin/bash
# Run ON THE GPU BOX: the collision row with the general sampler (ILM_DF_SLICE0=0) and the slice-0 form (1).
cd "$(cd "$(dirname "$0")/.." && pwd)"
for round in 1 2 3; do
for v in 0 1; do
  ILM_DF_SLICE0=$v python bench.py --no-cpu-baseline --no-cfg4 --no-lighting --steps 20 --warmup 5 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['next_rows']['collision_step_1m']; print('slice0=$v', 'collision %.2f us (min %.2f)  plain %.2f us  samples/particle %s' % (r['us_per_step'], r['us_per_step_min'], r['us_per_step_update_positions'], r.get('sdf_samples_per_particle')))"
done
done
