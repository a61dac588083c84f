#!/bin/bash
# Run ON THE GPU BOX: the default bench line N times back to back in one call (same box, same binary) -- what of the spread between
# committed records is the box's and what is run-to-run.   usage: tools/bench_repeat.sh <round> [runs]
round=${1:-r05}; runs=${2:-5}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/${round}_bench_repeat.txt
echo "# python bench.py (defaults), $runs runs back to back on one box: $(date -u +%FT%TZ)" > $out
for i in $(seq 1 $runs); do
    python bench.py 2>/dev/null | python -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
L = d['lighting']
c2 = d.get('cfg2_cache_resident', d)
print('run $i: cfg2 %.1f Mparticle-steps/s (frac %.3f, %.5f ms/step) | cfg4 share %.1f (frac %.3f) | 64M (the headline) %.1f (frac %.3f, %.3f ms) | cfg3 %.4f ms | cfg5 %.4f ms (%.1f lit Mpx/s) | particle lights %.4f ms' % (
    c2.get('mparticle_steps_per_s', c2.get('value')), c2['roofline']['frac'], c2['ms_per_step'], d['cfg4_share_8m_particles']['mparticle_steps_per_s'], d['cfg4_share_8m_particles']['roofline']['frac'],
    d['cfg4_full_64m_one_gpu']['mparticle_steps_per_s'], d['cfg4_full_64m_one_gpu']['roofline']['frac'], d['cfg4_full_64m_one_gpu']['ms_per_step'],
    L['cfg3_1080p_64_lights_unorm16']['ms_per_frame'], L['cfg5_4k_256_lights_fp16']['ms_per_frame'], L['cfg5_4k_256_lights_fp16']['lit_mpixels_per_s'],
    d['next_rows']['particle_lights_1080p_4096']['ms_per_frame']))" >> $out
done
cat $out
