#!/usr/bin/env python
"""Idle time between kernels in a rocprofv3 --kernel-trace (rocpd / SQLite) database: how much of the timed region no kernel was
running on the device, and after which kernels the gaps sit.

    python tools/rocpd_gaps.py gpurun_out/prof/x_results.db [fraction of the trace to analyse, from the end; default 0.5]
"""
import collections
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    name = name.replace('dr::', '')
    return name if len(name) < 60 else name[:57] + '...'


def main(path, frac=0.5):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
    sel = 'select name, start, end%s from kernels order by start' % (', ' + qcol if qcol else '')
    rows = db.execute(sel).fetchall()
    if not rows:
        sys.exit('no kernels')
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    cut = t1 - (t1 - t0) * frac
    rows = [r for r in rows if r[1] >= cut]
    span = max(r[2] for r in rows) - rows[0][1]
    busy_sum = sum(r[2] - r[1] for r in rows)
    # union of the busy intervals over every queue
    union, cur_s, cur_e = 0, rows[0][1], rows[0][2]
    gaps = collections.defaultdict(lambda: [0, 0])
    prev_name = rows[0][0]
    for name, s, e, *q in rows[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            g = gaps[short(prev_name) + ' -> ' + short(name)]
            g[0] += 1; g[1] += s - cur_e
            cur_s, cur_e = s, e
            prev_name = name
        elif e > cur_e:
            cur_e = e
            prev_name = name
    union += cur_e - cur_s
    print('# idle time between kernels: `%s` (last %.0f %% of the trace)' % (path, frac * 100))
    print()
    print('columns of `kernels`: %s' % ', '.join(cols))
    print()
    print('| kernels | span ms | sum of kernel durations ms | device busy (union) ms | idle ms | idle %% |')
    print('|---:|---:|---:|---:|---:|---:|')
    print('| %d | %.2f | %.2f | %.2f | %.2f | %.1f |' % (len(rows), span / 1e6, busy_sum / 1e6, union / 1e6, (span - union) / 1e6, 100.0 * (span - union) / span))
    if qcol:
        print()
        print('| %s | kernels | busy ms |' % qcol)
        print('|---|---:|---:|')
        per = collections.defaultdict(lambda: [0, 0])
        for r in rows:
            per[r[3]][0] += 1; per[r[3]][1] += r[2] - r[1]
        for k, v in sorted(per.items(), key=lambda kv: -kv[1][1]):
            print('| %s | %d | %.2f |' % (k, v[0], v[1] / 1e6))
    print()
    print('| gap after -> before | count | total us | avg us |')
    print('|---|---:|---:|---:|')
    for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
        print('| `%s` | %d | %.1f | %.2f |' % (k, v[0], v[1] / 1e3, v[1] / 1e3 / v[0]))
    hist = collections.Counter()
    for k, v in gaps.items():
        pass
    print()
    print('gaps in total: %d, %.2f ms' % (sum(v[0] for v in gaps.values()), sum(v[1] for v in gaps.values()) / 1e6))


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
