#!/bin/bash
# Run ON THE GPU BOX: the particle rows of bench.py (cfg2, cfg4 share, cfg4 whole, collision) with variant builds of the library (tools/ab/<tag>/).
#   tools/ab_steps.sh base v1 ...
cd "$(cd "$(dirname "$0")/.." && pwd)"
for round in 1 2; do
for tag in "$@"; do
  ILM_HIP_LIB=$PWD/tools/ab/$tag/libilluminant_hip.so LD_LIBRARY_PATH=$PWD/tools/ab/$tag:${LD_LIBRARY_PATH:-} python bench.py --no-cpu-baseline --no-lighting --steps 20 --warmup 5 2>/dev/null | \
    python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['next_rows']['collision_step_1m']; c2=d.get('cfg2_cache_resident') or d
print('%-10s' % '$tag', 'cfg2 %.2f us  share %.2f us  64m %.1f us  collision %.2f us (min %.2f)  plain %.2f us' % (c2['ms_per_step']*1e3, d['cfg4_share_8m_particles']['ms_per_step']*1e3, d['cfg4_full_64m_one_gpu']['ms_per_step']*1e3, r['us_per_step'], r['us_per_step_min'], r['us_per_step_update_positions']))"
done
done
