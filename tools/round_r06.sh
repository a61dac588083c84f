set -u
bash tools/profile_round.sh r06 > gpurun_out/profile_round_r06.log 2>&1
lib=/tmp/libfake_rccl.so
g++ -O1 -std=c++17 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/fake_rccl.cpp -o $lib -L/opt/rocm/lib -lamdhip64 -lrt -lpthread
{ ILM_BENCH_FORCE_DIST=1 timeout 600 python bench.py --dry-collectives; echo; ILM_RCCL_LIB=$lib ILM_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python bench.py --gpus 8 --dry-collectives; } > gpurun_out/r06_collective_schedule.txt 2> gpurun_out/r06_collective_schedule.err
bash tools/stand_in_bench.sh r06 2 8 > gpurun_out/stand_in_r06.log 2>&1
tail -3 gpurun_out/stand_in_r06.log; wc -l gpurun_out/r06_collective_schedule.txt; ls gpurun_out/profiles_r06 | head -40
