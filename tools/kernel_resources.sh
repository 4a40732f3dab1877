#!/bin/bash
# Registers / scratch / LDS / occupancy of every kernel of a translation unit, as the compiler reports them (no GPU needed).
#   tools/kernel_resources.sh msm_g1.hip [-DZK_... extra flags]
cd "$(dirname "$0")/../zksnark_rs_amd/csrc"
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off "$@" -c $f -o /tmp/kr_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re, sys, subprocess
cur = None
rows = []
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = {'name': m.group(1)}; rows.append(cur); continue
    for key, pat in (('vgpr', r' VGPRs: (\d+)'), ('agpr', r'AGPRs: (\d+)'), ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'), ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'), ('lds', r'LDS Size \[bytes/block\]: (\d+)'), ('sgpr', r' SGPRs: (\d+)')):
        m = re.search(pat, line)
        if m and cur is not None: cur[key] = m.group(1)
names = subprocess.run(['c++filt'] + [r['name'] for r in rows], capture_output=True, text=True).stdout.split('\n')
print('%-70s %5s %5s %7s %4s %6s' % ('kernel', 'vgpr', 'agpr', 'scratch', 'occ', 'lds'))
for r, n in zip(rows, names):
    n = re.sub(r'\(.*', '', n).replace('void ', '').replace('zk::', '').replace('Fp<FqParams>', 'Fq').replace('Fp<FrParams>', 'Fr')
    print('%-70s %5s %5s %7s %4s %6s' % (n[:70], r.get('vgpr'), r.get('agpr'), r.get('scratch'), r.get('occ'), r.get('lds')))
"
rm -f /tmp/kr_$$.o
