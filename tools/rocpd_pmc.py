#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; ROCm 7.2 rocpd DBs).

    python tools/rocpd_pmc.py gpurun_out/pmc_fetch/fetch_results.db gpurun_out/pmc_write/write_results.db

Unit handling as MI355X_MICROARCH.md "HBM" prescribes: counters are kilobytes; on gfx950 FETCH_SIZE
reports exactly half the bytes of a wide coalesced streaming read (128-B requests tallied as 64 B), so it is
doubled; WRITE_SIZE is uncalibrated and taken as reported.
"""
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute('select kernel_name, count(*), sum(value), sum(duration) from counters_collection '
                      'where counter_name=? group by kernel_name', (counter,)).fetchall()
    return {r[0]: (r[1], r[2], r[3]) for r in rows}


def main(fetch_db, write_db):
    f = per_kernel(fetch_db, 'FETCH_SIZE')
    w = per_kernel(write_db, 'WRITE_SIZE')
    names = sorted(set(f) | set(w), key=lambda n: -(2 * f.get(n, (0, 0, 0))[1] + w.get(n, (0, 0, 0))[1]))
    print('# HBM traffic per kernel from PMC passes (FETCH_SIZE x2 gfx950 correction, WRITE_SIZE as reported)')
    print()
    print('| kernel | launches | read MB/launch (corrected) | write MB/launch | total GB (all launches) |')
    print('|---|---:|---:|---:|---:|')
    for n in names:
        nf, kb_f, _ = f.get(n, (0, 0.0, 0))
        nw, kb_w, _ = w.get(n, (0, 0.0, 0))
        launches = max(nf, nw, 1)
        rd = 2.0 * kb_f * 1024 / launches / 1e6
        wr = kb_w * 1024 / max(nw, 1) / 1e6
        short = n if len(n) < 90 else n[:87] + '...'
        print('| `%s` | %d | %.2f | %.2f | %.2f |' % (short, launches, rd, wr, (2.0 * kb_f + kb_w) * 1024 / 1e9))


def short_name(n):
    """rocprof kernel name -> the row name of densereg_profile.h (net.h, kKernelNames): one row per kernel template family"""
    import re
    m = re.search(r'conv_igemm_kernel<(\d+), (\d+), \d+, \d+, \d+, (\d+), \d+, \d+, \d+, \d+, (\d+)>', n)
    if m:                                                          # BM, BN, WM, WN, ABL, BK, GL, BF, WK, XB, MF
        bm, bn, bk, mf = m.groups()
        return 'conv_igemm%s_%sx%s%s' % ('16' if mf == '16' else '', bm, bn, 'k64' if bk == '64' else '')
    m = re.search(r'conv_igemm_kernel<(\d+), (\d+)', n)
    if m:
        return 'conv_igemm_%sx%s' % (m.group(1), m.group(2))
    if 'conv_x3_kernel' in n or 'conv_x3h_kernel' in n:            # every tile / variant of conv_x3.h and conv_x3h.h is one profile row (net.h: KID_CONV_X3)
        return 'conv_x3_128x128'
    if 'conv_splitk_kernel' in n:
        return 'conv_splitk_32x32'
    if 'conv_wgrad_row_kernel' in n:
        return 'conv_wgrad_row96'
    if 'conv_wgrad_group_kernel' in n:
        return 'conv_wgrad_group'
    m = re.search(r'conv_wgrad_x3_kernel<(\d+)', n)
    if m:
        return 'conv_wgrad_x3_%s' % m.group(1)
    m = re.search(r'conv_wgrad(?:_bf16|_tr)?_kernel<(\d+)', n)
    if m:
        return 'conv_wgrad_%s' % m.group(1)
    if 'conv_wgrad16_kernel' in n:
        return 'conv_wgrad16'
    if 'stem_wgrad_kernel' in n:
        return 'stem_wgrad'
    if 'hg_tail_eval_kernel' in n:
        return 'hourglass_tail_fused'
    return None


def to_json(fetch_db, write_db, mode, out_path, stamp_path=None):
    """Aggregate by kernel family into out_path[mode].  stamp_path: JSON written on the GPU box when the passes ran
    ({"kernel_source_hash", "git_head", "date"}; tools/gpu/pmc_passes.sh) -- stored as out_path[mode]["_stamp"] together with the
    full kernel template names of each family, so that bench.py can refuse a measurement of other kernels."""
    import json
    import os
    f = per_kernel(fetch_db, 'FETCH_SIZE')
    w = per_kernel(write_db, 'WRITE_SIZE')
    agg = {}
    names = {}
    for n in set(f) | set(w):
        k = short_name(n)
        if not k:
            continue
        names.setdefault(k, []).append(n.split('(')[0])
        a = agg.setdefault(k, [0, 0.0, 0, 0.0])
        nf, kb_f, _ = f.get(n, (0, 0.0, 0))
        nw, kb_w, _ = w.get(n, (0, 0.0, 0))
        a[0] += nf; a[1] += kb_f; a[2] += nw; a[3] += kb_w
    data = json.load(open(out_path)) if os.path.exists(out_path) else {}
    data[mode] = {k: {'read_bytes_per_launch': 2.0 * v[1] * 1024 / max(v[0], 1), 'write_bytes_per_launch': v[3] * 1024 / max(v[2], 1),
                      'launches_sampled': v[0],
                      'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950), KB -> bytes'}
                  for k, v in agg.items()}
    stamp = json.load(open(stamp_path)) if stamp_path and os.path.exists(stamp_path) else {}
    stamp['kernels'] = {k: sorted(set(v)) for k, v in names.items()}
    data[mode]['_stamp'] = stamp
    json.dump(data, open(out_path, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    if len(sys.argv) >= 6 and sys.argv[3] == '--json':
        to_json(sys.argv[1], sys.argv[2], sys.argv[4], sys.argv[5], sys.argv[6] if len(sys.argv) > 6 else None)
    else:
        main(sys.argv[1], sys.argv[2])
