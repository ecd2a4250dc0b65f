#!/usr/bin/env python
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<round>_pmc_traffic.json.

Usage (on the GPU box, three separate passes as MI355X_MICROARCH.md prescribes -- FETCH_SIZE and WRITE_SIZE do not
fit one pass, and PMC must not be combined with the API trace domains):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/r01_pmc_traffic.json
Units / corrections (guide, section HBM): the counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide
(16 B/lane) coalesced reads, so it is doubled; WRITE_SIZE is taken as is.  Output: bytes per launch per kernel."""
import collections
import csv
import glob
import json
import os
import re
import sys


def _dt(n):
    """element type of a kernel instantiation from its (mangled or demangled) name"""
    if re.search(r'fp8e4_t', n):
        return 'fp8'
    if re.search(r'fp8e5_t', n):
        return 'fp8'
    if re.search(r'IDF16b|<__bf16|<bf16|bool _Accum', n):          # (rocprof's demangler garbles <__bf16, ...>)
        return 'bf16'
    return 'f32'


def short(name):
    """map a (mangled or rocprof-demangled) kernel name to the label dualvar_amd.engine gives the launch"""
    n = name
    m = re.search(r'conv_wgrad_dma_kernelI(?:DF16b|f)Li(\d+)ELi(\d+)E', n) or re.search(r'conv_wgrad_dma_kernel<[^,]+, (\d+), (\d+)', n)
    if m:
        # the form that carries the BatchNorm backward's apply (dv_conv3d_wgrad_bn) is a family of its own
        bna = re.search(r'conv_wgrad_dma_kernelI\S*Lb1ELb1ELb1EE', n) or re.search(r'conv_wgrad_dma_kernel<[^>]*true, true, true>', n)
        return 'conv_wgrad<%s,16,%s,%s>%s' % (_dt(n), m.group(1), m.group(2), '+bn_bwd_apply' if bna else '')
    m = re.search(r'conv_wgrad_kernelI(?:DF16b|f)Li(\d+)ELi(\d+)ELi(\d+)E', n) or re.search(r'conv_wgrad_kernel<[^,]+, (\d+), (\d+), (\d+)', n)
    if m:
        return 'conv_wgrad<%s,%s,%s,%s>' % ((_dt(n),) + m.groups())
    m = re.search(r'conv_gemm_kernelI(?:DF16b|f|\d+fp8e\d_t)Li(\d)ELi(\d+)ELi(\d+)ELi(\d+)E', n) or \
        re.search(r'conv_gemm_kernel<[^,]+, (\d), (\d+), (\d+), (\d+)', n) or \
        re.search(r'conv_gemm_kernel<bool _Accum, int, E, (\d+), (\d+), (\d+)', n)
    if m:
        g = m.groups()
        if len(g) == 3:                                             # garbled bf16 dgrad form
            g = ('1',) + g
        return 'conv_gemm<%s,%s,%s,%s,%s>' % (_dt(n), 'DGRAD' if g[0] == '1' else 'FWD', g[1], g[2], g[3])
    m = re.search(r'conv_gemm_ks_kernelILi(\d)ELi(\d+)E', n) or re.search(r'conv_gemm_ks_kernel<(\d), (\d+)', n)
    if m:
        return 'conv_gemm_ks<f32,%s,64,%s>' % ('DGRAD' if m.group(1) == '1' else 'FWD', m.group(2))
    # the LDS-staged kernels of round 4 (csrc/conv_tap.hip, conv_tap_wgrad.hip)
    m = re.search(r'conv_tap_kernelILi(\d)E', n) or re.search(r'conv_tap_kernel<(\d)', n)
    if m:
        bm = '128' if re.search(r'conv_tap_kernel<[^>]*, 128>|conv_tap_kernelI\S*Li128EEE', n) else '256'
        return 'conv_tap<f32,%s,%s,64>' % ('sp' if m.group(1) == '0' else 'tm', bm)          # (fwd and dgrad are one kernel)
    if re.search(r'conv_pp_fwd_kernel', n):
        return 'conv_pp<f32,FWD>'
    m = re.search(r'conv_wgrad_tm_kernelILi\d+ELi\d+ELi(\d+)ELi\d+ELi\d+ELi(\d+)E', n) or \
        re.search(r'conv_wgrad_tm_kernel<\d+, \d+, (\d+), \d+, \d+, (\d+)(?:, (?:true|false))?>', n)      # (BatchNorm-on-load instantiations: same family)
    if m:
        return 'conv_wgrad<f32,16,64,%d>' % (int(m.group(1)) * int(m.group(2)))
    if re.search(r'conv_wgrad_sp_kernel', n):
        return 'conv_wgrad<f32,16,64,192>'
    if re.search(r'conv_wgrad_pp_kernel', n):
        return 'conv_wgrad<f32,16,64,224>' + ('+bn_bwd_apply' if re.search(r'pp_kernelILb1E|pp_kernel<true>', n) else '')
    m = re.search(r'conv_wgrad_f32s_kernelILi(\d+)ELi(\d+)E', n) or re.search(r'conv_wgrad_f32s_kernel<(\d+), (\d+)', n)
    if m:
        return 'conv_wgrad<f32,16,%s,%s>' % (m.group(1), m.group(2))
    for key, pat in (('bn_bwd_reduce_multi', r'bn_bwd_reduce_multi'), ('bn_bwd_apply_multi', r'bn_bwd_apply_multi'),
                     ('bn_apply_multi', r'bn_apply_multi'), ('bn_bwd_reduce', r'bn_bwd_reduce_kernel'),
                     ('bn_bwd_apply', r'bn_bwd_apply_kernel'), ('bn_apply_maxpool', r'bn_apply_maxpool'), ('bn_apply', r'bn_apply_kernel'),
                     ('maxpool_fwd', r'maxpool_fwd|pool333_fwd_tile'), ('maxpool_bwd', r'maxpool_bwd|pool333_bwd_tile')):
        if re.search(pat, n):
            return '%s<%s>' % (key, _dt(n))
    if re.search(r'bn_stats_multi', n):
        return 'bn_stats_multi'
    if re.search(r'wgrad_reduce_kernel', n):
        return 'wgrad_reduce'
    return None


def load(d, counter):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                agg[r['Kernel_Name']] += float(r['Counter_Value'])
                cnt[r['Kernel_Name']] += 1
    return agg, cnt


def main():
    fdir, wdir, out = sys.argv[1:4]
    f, nf = load(fdir, 'FETCH_SIZE')
    w, nw = load(wdir, 'WRITE_SIZE')
    fam = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
    per_kernel = {}
    for k in set(f) | set(w):
        fb = 2.0 * f.get(k, 0.0) * 1024.0 / max(nf.get(k, 1), 1)
        wb = w.get(k, 0.0) * 1024.0 / max(nw.get(k, 1), 1)
        per_kernel[k] = {'launches': nf.get(k, nw.get(k, 0)), 'fetch_bytes_per_launch': fb, 'write_bytes_per_launch': wb}
        s = short(k)
        if s:
            e = fam[s]
            e[0] += nf.get(k, 0); e[1] += 2.0 * f.get(k, 0.0) * 1024.0
            e[2] += nw.get(k, 0); e[3] += w.get(k, 0.0) * 1024.0
    families = {s: {'launches': e[0], 'hbm_bytes_per_launch': (e[1] / max(e[0], 1)) + (e[3] / max(e[2], 1))} for s, e in fam.items()}
    json.dump({'note': 'FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, KiB -> bytes, per launch',
               'families': families, 'kernels': per_kernel}, open(out, 'w'), indent=1, sort_keys=True)
    print('wrote', out, {k: int(v['hbm_bytes_per_launch']) for k, v in families.items()})


if __name__ == '__main__':
    main()
