#!/bin/bash
# Build container: registers / scratch / LDS of every kernel of one csrc file as the compiler reports them (no GPU needed).
#   tools/kernel_resources.sh lighting [extra hipcc flags]
cd "$(cd "$(dirname "$0")/.." && pwd)/illuminant_amd/csrc"
f=$1; shift
extra=""
case $f in particles|lighting|fields|raster) extra="-fno-slp-vectorize";; esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $extra "$@" -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/kr_$f.o 2>&1 | \
  python3 -c "
import re, sys, subprocess
rows, cur = [], None
keep = {'VGPRs': 'vgpr', 'TotalSGPRs': 'sgpr', 'ScratchSize [bytes/lane]': 'scratch', 'Occupancy [waves/SIMD]': 'occ', 'LDS Size [bytes/block]': 'lds',
        'SGPRs Spill': 'sspill', 'VGPRs Spill': 'vspill'}
for line in sys.stdin:
    m = re.search(r'remark:\s+(.*?)\s*\[-Rpass', line)
    if not m:
        if 'error' in line: print(line, end='')
        continue
    t = m.group(1)
    if t.startswith('Function Name:'):
        cur = {'fn': t.split(':', 1)[1].strip()}; rows.append(cur)
    elif cur is not None:
        k, _, v = t.partition(':')
        if k.strip() in keep: cur[keep[k.strip()]] = v.strip()
names = subprocess.run(['c++filt'], input='\n'.join(r['fn'] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    print('%-100s' % n.split('(')[0][:100], ' '.join('%s=%s' % (k, v) for k, v in r.items() if k != 'fn'))
"
