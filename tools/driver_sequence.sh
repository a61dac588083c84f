#!/bin/bash
# Run ON THE GPU BOX: what the driver runs at round end on a fresh box, timed -- pytest -m gpu, smoke(), the default bench.   tools/driver_sequence.sh <round>
round=${1:-r06}
cd "$(cd "$(dirname "$0")/.." && pwd)"
out=gpurun_out/${round}_driver_sequence.txt
{
echo "# the round-end sequence on one fresh box: $(date -u +%FT%TZ), git $(cat .git_head 2>/dev/null)"
t0=$(date +%s.%N); python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1; t1=$(date +%s.%N)
echo "pytest -m gpu: $(python3 -c "print('%.1f s' % ($t1 - $t0))")"
t0=$(date +%s.%N); python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1; t1=$(date +%s.%N)
echo "smoke(): $(python3 -c "print('%.1f s' % ($t1 - $t0))")"
t0=$(date +%s.%N); python bench.py > gpurun_out/${round}_bench_driver_cmd.json 2>/dev/null; rc=$?; t1=$(date +%s.%N)
echo "python bench.py: rc $rc, $(python3 -c "print('%.1f s' % ($t1 - $t0))")"
python3 - <<PY
import json
d = json.loads([l for l in open("gpurun_out/${round}_bench_driver_cmd.json") if l.startswith("{")][-1])
print("value %.1f %s | ms_per_step %.5f | roofline.frac %.4f (calibrated %.4f) | workload: %s" % (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("calibrated_frac") or 0.0, d["config"]["workload"][:60]))
print("cfg5 %.4f ms (%.1f lit Mpx/s) | cfg3 %.4f ms | particle lights %.4f ms | collision 1 M %.2f us, 8.4 M %.2f us" % (
    d["lighting"]["cfg5_4k_256_lights_fp16"]["ms_per_frame"], d["lighting"]["cfg5_4k_256_lights_fp16"]["lit_mpixels_per_s"], d["lighting"]["cfg3_1080p_64_lights_unorm16"]["ms_per_frame"],
    d["next_rows"]["particle_lights_1080p_4096"]["ms_per_frame"], d["next_rows"]["collision_step_1m"]["us_per_step"], d["next_rows"]["collision_step_8m"]["us_per_step"]))
PY
t0=$(date +%s.%N); python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1; t1=$(date +%s.%N)
echo "pytest -m gpu, a second time on the same box: $(python3 -c "print('%.1f s' % ($t1 - $t0))")"
} > $out 2>&1
cat $out
