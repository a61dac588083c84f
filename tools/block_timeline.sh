#!/bin/bash
# Run ON THE GPU BOX: the device timeline of the driver's timed blocks (bench.py --steps 20 --warmup 5): every step kernel of the last
# block with its start relative to the block's first kernel, its duration and the idle time in front of it.
cd "$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
rm -rf /tmp/btl
timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/btl -o t -- python bench.py --steps ${1:-20} --warmup 5 --no-cpu-baseline --no-lighting --no-cfg4 --no-next-rows > /tmp/btl.json 2>/tmp/btl.log
python3 - <<'PY'
import csv, glob
ev = []
for p in glob.glob('/tmp/btl/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[-44:]))
ev.sort()
steps = [e for e in ev if 'step_' in e[2]]
# blocks: separated by more than 100 us of idle
blocks, cur = [], [steps[0]]
for a, b in zip(steps, steps[1:]):
    if b[0] - a[1] > 100000: blocks.append(cur); cur = []
    cur.append(b)
blocks.append(cur)
print('%d blocks of' % len(blocks), sorted(set(len(b) for b in blocks)), 'step kernels')
for blk in blocks[-2:]:
    t0 = blk[0][0]; prev = t0
    print('block: %d kernels, first start to last end %.1f us' % (len(blk), (blk[-1][1] - t0) / 1e3))
    for s, e, n in blk:
        print('   %8.1f us  +%5.1f idle  %6.1f us  %s' % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n))
        prev = e
PY
