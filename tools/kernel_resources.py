#!/usr/bin/env python
"""Per-kernel register / LDS / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py dualvar_amd/csrc/conv.hip [name filter]
"""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dualvar_amd.build import FLAGS   # noqa: E402


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    r = subprocess.run(['hipcc'] + FLAGS + ['-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null'],
                       capture_output=True, text=True)
    cur = None
    rows = []
    for line in r.stderr.splitlines():
        m = re.search(r'remark: [^:]*:\d+:\d+: +(Function Name|[A-Za-z ]+(?:\[[^\]]*\])?): (.*?) \[-Rpass', line) or \
            re.search(r': +(Function Name|Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)', line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k in ('Function Name', 'Name'):
            cur = {'name': v}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(' [')[0]] = v
    for c in rows:
        name = subprocess.run(['c++filt', c['name']], capture_output=True, text=True).stdout.strip()
        name = re.sub(r'\(anonymous namespace\)::', '', name)
        name = re.sub(r'\(.*$', '', name)
        if flt and flt not in name:
            continue
        print('%-90s vgpr %4s agpr %4s sgpr %4s lds %6s scratch %3s occ %s' % (
            name[:90], c.get('VGPRs'), c.get('AGPRs'), c.get('TotalSGPRs'), c.get('LDS Size'), c.get('ScratchSize'), c.get('Occupancy')))


if __name__ == '__main__':
    main()
