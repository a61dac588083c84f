#!/bin/bash
# Run ON THE GPU BOX: tools/fuzz_parity.py over fresh seeds in parallel processes (the sweep is bound by the CPU oracle).  AT MOST 8 processes:
# from 16 per GPU on, kernels now and then read stale memory behind a completed copy (profiles/r06_memcpy_order_under_oversubscription.txt).
#   tools/fuzz_sweep.sh <first_seed> <seeds_per_process> <processes> <tag> [seconds per process: stop drawing seeds after that]   ->  gpurun_out/fuzz_<tag>/p<i>.txt + summary.txt
set -u
cd "$(cd "$(dirname "$0")/.." && pwd)"
first=$1; per=$2; procs=$3; tag=$4; seconds=${5:-0}
out=gpurun_out/fuzz_$tag; rm -rf "$out"; mkdir -p "$out"
t0=$(date +%s)
for i in $(seq 0 $((procs - 1))); do
  ILM_FUZZ_SECONDS=$seconds OMP_NUM_THREADS=1 python tools/fuzz_parity.py $((first + i * per)) $per > "$out/p$i.txt" 2>&1 &
done
wait
t1=$(date +%s)
{
  echo "tools/fuzz_sweep.sh $first $per $procs: seeds $first..$((first + per * procs - 1)) in $procs processes, $((t1 - t0)) s, library $(sha256sum illuminant_amd/lib/libilluminant_hip.so | cut -c1-16), git $(cat .git_head 2>/dev/null)"
  grep -h "FUZZ" "$out"/p*.txt | sort | uniq -c
  [ "$procs" -gt 8 ] && echo "NOTE: with more than 8 processes on one GPU kernels now and then read stale memory behind a completed copy (profiles/r06_memcpy_order_under_oversubscription.txt): re-run reported seeds alone"
  echo "seeds done: $(grep -h "^seeds " "$out"/p*.txt | awk '{split($2, a, "[.][.]"); n += a[2] - a[1] + 1} END {print n}')"
  python3 - "$out" <<'PY'
import glob, re, sys
tot = {"lighting scenes (statistics exact, floats <= 1e-4)": 0, "collision steps": 0, "collision elements (M)": 0.0, "particles bounced or redirected": 0, "particle elements (M)": 0.0,
       "particle elements that needed the absolute floor": 0, "field scenes": 0, "G-buffer scenes": 0}
worst_l, worst_p, problems = 0.0, 0.0, 0
for f in sorted(glob.glob(sys.argv[1] + "/p*.txt")):
    t = open(f).read()
    m = re.search(r"^seeds (\d+)\.\.(\d+)", t, re.M)
    if m: tot["lighting scenes (statistics exact, floats <= 1e-4)"] += int(m.group(2)) - int(m.group(1)) + 1
    m = re.search(r"collision update: (\d+) problems in (\d+) steps \(([\d.]+) M elements; (\d+) particles", t)
    if m: problems += int(m.group(1)); tot["collision steps"] += int(m.group(2)); tot["collision elements (M)"] += float(m.group(3)); tot["particles bounced or redirected"] += int(m.group(4))
    m = re.search(r"lighting: (\d+) scenes with differing statistics or > 1e-4 error; worst relative error ([\d.e+-]+)", t)
    if m: problems += int(m.group(1)); worst_l = max(worst_l, float(m.group(2)))
    m = re.search(r"particles: (\d+) steps with differing .*? ([\d.e+-]+)\s*$", t, re.M)
    if m: problems += int(m.group(1)); worst_p = max(worst_p, float(m.group(2)))
    m = re.search(r"particle floats: (\d+) failures .*? in ([\d.]+) M elements; (\d+) elements", t)
    if m: problems += int(m.group(1)); tot["particle elements (M)"] += float(m.group(2)); tot["particle elements that needed the absolute floor"] += int(m.group(3))
    m = re.search(r"particle lights: (\d+) scenes .*? of (\d+) \(([\d.]+) M pixel", t)
    if m: problems += int(m.group(1)); tot["particle-light scenes"] = tot.get("particle-light scenes", 0) + int(m.group(2)); tot["particle-light pairs (M)"] = tot.get("particle-light pairs (M)", 0.0) + float(m.group(3))
    m = re.search(r"not failures\): (\d+)", t)
    if m: tot["float misses explained by the reference's own cancellation"] = tot.get("float misses explained by the reference's own cancellation", 0) + int(m.group(1))
    m = re.search(r"field generation: (\d+) scenes with differing codes of (\d+)", t)
    if m: problems += int(m.group(1)); tot["field scenes"] += int(m.group(2))
    m = re.search(r"G-buffer meshes: (\d+) scenes with a differing texel of (\d+)", t)
    if m: problems += int(m.group(1)); tot["G-buffer scenes"] += int(m.group(2))
for k, v in tot.items(): print("   %-62s %s" % (k, ("%.1f" % v) if isinstance(v, float) else v))
print("   worst lighting relative error %.3g; worst particle error relative to (|want| + 1e-4 scale) %.3g; problems reported by the processes: %d" % (worst_l, worst_p, problems))
PY
  grep -L "FUZZ PASSED" "$out"/p*.txt | sed 's/^/NOT PASSED: /'
} > "$out/summary.txt"
cat "$out/summary.txt"
