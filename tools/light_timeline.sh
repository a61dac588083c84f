#!/bin/bash
# Run ON THE GPU BOX: the device timeline of a few lit frames of tools/light_one.py (kernels, copies, fills with the idle time in front of each).
#   tools/light_timeline.sh cfg3 [frames]
cd "$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
rm -rf /tmp/ltl
timeout 250 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ltl -o t -- python tools/light_one.py ${1:-cfg3} all 0 ${2:-6} > /tmp/ltl.out 2>/tmp/ltl.log
python3 - <<'PY'
import csv, glob
ev = []
for p in glob.glob('/tmp/ltl/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[-50:]))
for p in glob.glob('/tmp/ltl/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'copy ' + r.get('Direction', '')))
ev.sort()
last = [i for i, e in enumerate(ev) if 'sphere_lights' in e[2]]
a = last[-4] if len(last) >= 4 else 0
t0 = ev[a][0]; prev = ev[a][0]
for s, e, n in ev[a:]:
    print('%9.1f us  +%6.1f idle  %8.1f us  %s' % ((s - t0) / 1e3, max(0, s - prev) / 1e3, (e - s) / 1e3, n))
    prev = max(prev, e)
PY
