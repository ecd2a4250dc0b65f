#!/usr/bin/env python
"""Per-kernel register / LDS / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py dualvar_amd/csrc/conv.hip [name filter]

Reads the report the library build keeps next to the object (csrc/<name>.res) when it is newer than the source, otherwise
compiles the file.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dualvar_amd.build import FLAGS, RES_FLAG, parse_resources   # noqa: E402


def demangle(name):
    name = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    return re.sub(r'\(.*$', '', name)


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    res = src.replace('.hip', '.res')
    if os.path.exists(res) and os.path.getmtime(res) >= os.path.getmtime(src):
        text = open(res).read()
    else:
        text = subprocess.run(['hipcc'] + FLAGS + [RES_FLAG, '-c', src, '-o', '/dev/null'], capture_output=True, text=True).stderr
    for c in parse_resources(text):
        name = demangle(c['name'])
        if flt and flt not in name:
            continue
        print('%-90s vgpr %4s agpr %4s sgpr %4s lds %6s scratch %3s occ %s' % (
            name[:90], c.get('VGPRs'), c.get('AGPRs'), c.get('TotalSGPRs'), c.get('LDS Size'), c.get('ScratchSize'), c.get('Occupancy')))


if __name__ == '__main__':
    main()
