#!/bin/bash
# Run ON THE GPU BOX: every strip of an n-rank lit frame rendered alone on this GPU (ILM_BENCH_STRIP), equal bands and cost-balanced strips:
# what the slowest rank of the composited frame would launch.   tools/strip_probe.sh 8
cd "$(cd "$(dirname "$0")/.." && pwd)"
n=${1:-8}
for kind in "" "balanced:"; do
for k in $(seq 0 $((n - 1))); do
  ILM_BENCH_STRIP=$kind$k/$n python bench.py --no-cpu-baseline --no-cfg4 --no-next-rows --steps 20 --light-frames 8 --light-ms 20 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${kind:-equal:}$k/$n', {k[:4]: (v['roofline']['launch_ms'], v.get('rows')) for k,v in d['lighting'].items()})"
done
done
