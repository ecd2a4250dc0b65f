"""Build libdualvar_hip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the
resulting .so travels to the GPU box with the repo snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libdualvar_hip.so')
SOURCES = ['conv.hip', 'conv_tap.hip', 'conv_tap_wgrad.hip', 'conv_experiments.hip', 'elementwise.hip', 'loss.hip', 'augment.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-comment', '-Wno-inline-asm', '-ffp-contract=off']
# every compile also reports registers / LDS / scratch per kernel; the report is kept next to the object (csrc/<name>.res) and
# tests/test_abi_and_host.py fails on any kernel with scratch > 0 (a spill arrived silently in round 3)
RES_FLAG = '-Rpass-analysis=kernel-resource-usage'


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def res_path(src):
    return os.path.join(CSRC, src.replace('.hip', '.res'))


def parse_resources(text):
    """hipcc -Rpass-analysis=kernel-resource-usage remarks -> [{'name', 'VGPRs', 'AGPRs', 'TotalSGPRs', 'ScratchSize', 'Occupancy', 'LDS Size'}]"""
    import re
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r': +(Function Name|Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|'
                      r'LDS Size \[bytes/block\]): (\S+)', line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k in ('Function Name', 'Name'):
            cur = {'name': v}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(' [')[0]] = v
    return rows


def build_lib(force=False, verbose=False):
    hipcc = os.environ.get('HIPCC', 'hipcc')
    hdrs = [os.path.join(CSRC, 'common.hpp'), os.path.join(CSRC, 'conv_common.hpp'), os.path.join(HERE, '..', 'include', 'dualvar_hip.h')]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + hdrs) or not os.path.exists(res_path(src)):
            cmd = [hipcc] + FLAGS + [RES_FLAG, '-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), file=sys.stderr)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (src, out.decode(errors='replace')))
        with open(res_path(src), 'w') as f:
            f.write(out.decode(errors='replace'))
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s' % r.stdout.decode(errors='replace'))
    return LIB


if __name__ == '__main__':
    print(build_lib(force='--force' in sys.argv, verbose=True))
