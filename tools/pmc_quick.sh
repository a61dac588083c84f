#!/bin/bash
# quick PMC look at one command (run on the GPU box): tools/pmc_quick.sh "<counters>" <cmd...>
export TMPDIR=/tmp
C="$1"; shift
rm -rf /tmp/pmcq; rocprofv3 --kernel-trace --output-format csv --pmc $C -d /tmp/pmcq -o q -- "$@" > /tmp/pmcq.out 2>/tmp/pmcq.log
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for p in glob.glob('/tmp/pmcq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        k=r['Kernel_Name'].replace('void ','').split('(')[0]
        a=agg[k][r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for k in sorted(agg):
    if 'ilm::' in k:
        print(k, {c: round(v[1]/v[0],1) for c,v in sorted(agg[k].items())}, 'n=%d' % max(v[0] for v in agg[k].values()))
PY
