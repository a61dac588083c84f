#!/bin/bash
# Run ON THE GPU BOX: everything profiles/<tag>_* quotes, in one call.   tools/profile_round.sh r03
set -u
TAG=${1:-r03}
cd "$(cd "$(dirname "$0")/.." && pwd)"
bash tools/profile_bench.sh "$TAG" > gpurun_out/profile_bench_$TAG.log 2>&1
S=gpurun_out/profiles_$TAG
tools/ubench/unorm > "$S/${TAG}_typed_unorm16_loads.txt" 2>&1
tools/ubench/valu > "$S/${TAG}_valu_issue_fma_and_packed_fma.txt" 2>&1
bash tools/pmc_light.sh $TAG > "$S/${TAG}_light_lane_activity.txt" 2>&1
bash tools/pmc_bound.sh $TAG > /dev/null 2>&1; cp gpurun_out/pmc_bound_$TAG/summary.txt "$S/${TAG}_pmc_bound_cell_array.txt"
bash tools/pmc_collision.sh $TAG > /dev/null 2>&1; cp gpurun_out/pmc_collision_$TAG/summary.txt "$S/${TAG}_collision_step.txt"
python tools/working_set_sweep.py > "$S/${TAG}_step_working_set_sweep.txt" 2>/dev/null
python bench.py > "$S/${TAG}_bench_default.json" 2> /dev/null
ILM_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$S/${TAG}_bench_forced_dist_world1.json" 2> /dev/null
# r05: one rank's strips under every light-split setting, the store-mode exchange on one device, the roctx ranges of a short run
python tools/strip_probe.py --gbuffer > "$S/${TAG}_strip_probe.txt" 2>&1
python tools/store_mode_probe.py 8 20 > "$S/${TAG}_store_mode_probe.txt" 2>&1
bash tools/marker_trace.sh "$TAG" > gpurun_out/marker_trace_$TAG.log 2>&1
ls -la "$S"
