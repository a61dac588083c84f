#!/bin/bash
# Run ON THE GPU BOX: the device timeline of one rasterised frame of tools/raster_one.py (kernels and copies with their start offsets and
# the idle time in front of each) -- where a frame's time goes that no kernel accounts for.    tools/raster_timeline.sh [steps] [frames]
cd "$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
rm -rf /tmp/rtl
timeout 250 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/rtl -o t -- python tools/raster_one.py ${1:-25} ${2:-4} > /tmp/rtl.out 2>/tmp/rtl.log
grep "ms per frame" /tmp/rtl.out
python3 - <<'PY'
import csv, glob
ev = []
for p in glob.glob('/tmp/rtl/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[-70:]))
for p in glob.glob('/tmp/rtl/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'copy ' + r.get('Direction', '')))
ev.sort()
# the last frame: from the last raster_setup_kernel on
starts = [i for i, e in enumerate(ev) if 'raster_setup_kernel' in e[2]]
if len(starts) >= 2:
    a, b = starts[-2], starts[-1]
    t0 = ev[a][0]
    prev_end = ev[a - 1][1] if a > 0 else t0
    busy = 0
    for s, e, n in ev[a:b]:
        print('%9.1f us  +%7.1f idle  %8.1f us  %s' % ((s - t0) / 1e3, max(0, s - prev_end) / 1e3, (e - s) / 1e3, n))
        busy += e - s
        prev_end = max(prev_end, e)
    print('frame period %.1f us, kernels + copies %.1f us' % ((ev[b][0] - t0) / 1e3, busy / 1e3))
PY
