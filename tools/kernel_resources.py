#!/usr/bin/env python
"""Static resource table of every kernel in libdensereg_hip.so: registers, scratch, LDS, occupancy, as reported by
hipcc's -Rpass-analysis=kernel-resource-usage for gfx950 (no GPU needed).

    python tools/kernel_resources.py > profiles/r01_kernel_resources.md
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect():
    """[(demangled kernel name, {'vgpr', 'agpr', 'sgpr', 'scratch', 'lds', 'occ'})] of the product build.  Compiles the product's
    sources with the resource remarks INTO A TEMPORARY DIRECTORY: the shipped densereg_amd/lib/*.so is never touched."""
    import tempfile
    with tempfile.TemporaryDirectory(prefix='dr_kres_') as tmp:
        env = dict(os.environ, DR_HIPCC_EXTRA='-Rpass-analysis=kernel-resource-usage', DR_OUT_DIR=tmp)
        log = subprocess.run([os.path.join(ROOT, 'build.sh'), '--product-only'], cwd=ROOT, env=env, capture_output=True, text=True)
    text = log.stdout + log.stderr
    rows, cur = [], None
    pats = (('sgpr', r'TotalSGPRs: (\d+)'), ('vgpr', r'\bVGPRs: (\d+)'), ('agpr', r'AGPRs: (\d+)'),
            ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'), ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'),
            ('lds', r'LDS Size \[bytes/block\]: (\d+)'))
    for line in text.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = {'name': m.group(1)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in pats:
            m = re.search(pat, line)
            if m and key not in cur:
                cur[key] = int(m.group(1))
    if not rows:
        sys.exit('no resource remarks in the build output:\n' + text[-2000:])
    names = subprocess.run(['c++filt'] + [r['name'] for r in rows], capture_output=True, text=True).stdout.strip().split('\n')
    out, seen = [], set()
    for r, n in sorted(zip(rows, names), key=lambda rn: rn[1]):
        n = re.sub(r'\(.*\)$', '', n).replace('void ', '')
        if n not in seen:
            seen.add(n)
            out.append((n, r))
    return out


def main():
    rows = collect()
    print('# Kernel resources of libdensereg_hip.so (hipcc -Rpass-analysis=kernel-resource-usage, gfx950)\n')
    print('Static, from the compiler: registers, scratch, LDS per workgroup and the occupancy they allow. Template arguments of')
    print('`conv_igemm_kernel`: BM, BN, WM, WN, ABL (ablation, 0 = product), BK, GL (LDS-DMA refill), BF (bf16 operands).\n')
    print('| kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | LDS B/workgroup | waves/SIMD |')
    print('|---|---:|---:|---:|---:|---:|---:|')
    for n, r in rows:
        print('| `%s` | %s | %s | %s | %s | %s | %s |' % (n, r.get('vgpr', '-'), r.get('agpr', '-'), r.get('sgpr', '-'),
                                                      r.get('scratch', '-'), r.get('lds', '-'), r.get('occ', '-')))


if __name__ == '__main__':
    main()
