#!/bin/bash
# Run ON THE GPU BOX: counters of every dispatch of the light kernels of one command, dispatch by dispatch.
#   tools/pmc_dispatches.sh "SQ_WAVES SQ_INSTS_VALU ..." python tools/light_one.py cfg5 far 8
export TMPDIR=/tmp
C="$1"; shift
rm -rf /tmp/pmcd; rocprofv3 --kernel-trace --output-format csv --pmc $C -d /tmp/pmcd -o q -- "$@" > /tmp/pmcd.out 2>/tmp/pmcd.log
python3 - <<'PY'
import csv, glob, collections
rows = collections.OrderedDict()
for p in glob.glob('/tmp/pmcd/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        k = r['Kernel_Name']
        if 'sphere_light' not in k: continue
        d = rows.setdefault(int(r['Dispatch_Id']), {})
        d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
for i, d in sorted(rows.items()):
    w = d.get('SQ_WAVES', 0) or 1
    print(i, ' '.join('%s=%.0f' % kv for kv in sorted(d.items())), '| per wave:', ' '.join('%s=%.1f' % (k, v / w) for k, v in sorted(d.items()) if k != 'SQ_WAVES'))
PY
