#!/bin/bash
# Run ON THE GPU BOX: counters of one command, averaged per kernel (all ilm:: kernels), bounded in time.
#   tools/pmc_kernels.sh "FETCH_SIZE TCC_HIT_sum ..." <seconds> python tools/raster_one.py 220 3
export TMPDIR=/tmp
C="$1"; T="$2"; shift; shift
rm -rf /tmp/pmck; timeout "$T" rocprofv3 --kernel-trace --output-format csv --pmc $C -d /tmp/pmck -o q -- "$@" > /tmp/pmck.out 2>/tmp/pmck.log || echo "(command ended with $?)"
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for p in glob.glob('/tmp/pmck/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        k = r['Kernel_Name'].replace('void ', '').split('(')[0]
        a = agg[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k in sorted(agg):
    if 'ilm::' in k:
        print(k, {c: round(v[1] / v[0], 1) for c, v in sorted(agg[k].items())}, 'n=%d' % max(v[0] for v in agg[k].values()))
PY
tail -2 /tmp/pmck.out
