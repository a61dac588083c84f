#!/bin/bash
# Run ON THE GPU BOX: cfg2 (Spawner + Gravity x 4 + Noise + UpdatePositions) through the lean step kernels and through the interpreting kernel
# (ILM_STEP_LEAN=0), at the driver's 20-step run and at bench.py's default 200-step blocks (profiles/r05_collision_step.txt
# showed the interpreter AHEAD at 20 steps).   tools/ab_step_lean_cfg2.sh
cd "$(cd "$(dirname "$0")/.." && pwd)"
for round in 1 2; do
for steps in 20 200; do
for lean in 1 0; do
  ILM_STEP_LEAN=$lean python bench.py --steps $steps --warmup 5 --no-lighting --no-cfg4 --no-cpu-baseline --no-next-rows 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('cfg2_cache_resident', d)
print('steps $steps  ILM_STEP_LEAN=$lean  cfg2 %.2f us per step (min %.2f max %.2f)  live particles per step %.0f  chunks at end %s' % (c['ms_per_step']*1e3, c['timed_blocks']['ms_per_step_min']*1e3, c['timed_blocks']['ms_per_step_max']*1e3, c.get('live_particles_per_step_avg', 0), c.get('chunks_at_end')))"
done; done; done
# cfg4's share (8 chunks of 1024^2, HBM-resident) and cfg4 whole (64 chunks) the same way
for lean in 1 0 1 0; do
  ILM_STEP_LEAN=$lean python bench.py --steps 100 --warmup 5 --no-lighting --no-cpu-baseline --no-next-rows 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ILM_STEP_LEAN=$lean  cfg4 share (8.4 M) %.2f us per step   cfg4 whole (67 M) %.4f ms per step' % (d['cfg4_share_8m_particles']['ms_per_step']*1e3, d['cfg4_full_64m_one_gpu']['ms_per_step']))"
done
