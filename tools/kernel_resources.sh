#!/bin/bash
# Register / scratch / LDS / occupancy of every kernel of one device file (no GPU needed): tools/kernel_resources.sh device/build.hip [extra flags]
cd "$(dirname "$0")/../intrinsic3d_amd/csrc"
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result "$@" -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 | python3 -c "
import sys, re
cur = None; rows = []
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m: cur = {'name': m.group(1)}; rows.append(cur); continue
    for key in ('VGPRs', 'AGPRs', 'SGPRs', 'ScratchSize \[bytes/lane\]', 'Occupancy \[waves/SIMD\]', 'LDS Size \[bytes/block\]', 'SGPRs Spill', 'VGPRs Spill'):
        m = re.search(r'remark: .*?    ' + key + r': (\d+)', l)
        if m and cur is not None: cur[key.split(' ')[0] + ('Spill' if 'Spill' in key else '')] = int(m.group(1))
import subprocess
for r in rows:
    n = subprocess.run(['/usr/bin/c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    n = re.sub(r'\(.*', '', n)[:70]
    print('%-70s vgpr %3d agpr %3d sgpr %3d scratch %4d occ %d lds %6d' % (n, r.get('VGPRs', -1), r.get('AGPRs', 0), r.get('SGPRs', -1), r.get('ScratchSize', -1), r.get('Occupancy', -1), r.get('LDS', -1)))
"
