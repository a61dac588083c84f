#!/bin/bash
# bench.py's N > 1 branch at world sizes 2 / 4 / 8 with every rank on GPU 0 (tests/fake_rccl.cpp stands in for RCCL, which refuses two
# ranks on one device): FUNCTIONAL records -- every collective, strip table, count check and exchange mode of the N-rank line runs for
# real; the figures are those of N processes time-slicing one GPU and mean nothing as scaling.   usage: tools/stand_in_bench.sh <round> [N ...]
set -u
round=${1:-r05}; shift || true
worlds=${*:-2 4 8}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
lib=/tmp/libfake_rccl.so
g++ -O1 -std=c++17 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/fake_rccl.cpp -o $lib -L/opt/rocm/lib -lamdhip64 -lrt -lpthread || exit 1
for n in $worlds; do
    ILM_RCCL_LIB=$lib ILM_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 1500 python bench.py --gpus $n --steps 6 --warmup 2 --light-frames 4 --light-ms 0 --sustain-s 0 \
        > gpurun_out/${round}_bench_${n}_ranks_one_gpu_stand_in.json 2> gpurun_out/${round}_bench_${n}_ranks_one_gpu_stand_in.err
    echo "world $n: rc $? $(python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/${round}_bench_${n}_ranks_one_gpu_stand_in.json") if l.startswith("{")][0])
    f = d["scaling_detail"]["frames"]
    print({k: (v.get("store_mode", {}).get("every_rank_holds_the_frame_of_the_rccl_exchange"), v.get("pipelined_exchange", {}).get("both_lightmaps_hold_the_same_frame")) for k, v in f.items()},
          {k: d["lighting"][k]["verified_counts"] for k in d["lighting"] if isinstance(d["lighting"][k], dict) and "verified_counts" in d["lighting"][k]})
except Exception as e:
    print("no record:", e)
PY
)"
done
