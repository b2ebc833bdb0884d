#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2 rocpd/SQLite) --kernel-trace --stats database as a markdown table.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/r01_x_kernel_stats.md
"""
import sqlite3
import sys


def main(path, title=''):
    db = sqlite3.connect(path)
    rows = db.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) '
                      'from kernels group by name order by sum(duration) desc').fetchall()
    total = sum(r[2] for r in rows) or 1
    print('# rocprofv3 --kernel-trace --stats summary' + (': ' + title if title else ''))
    print()
    print('source: `%s` (durations in microseconds)' % path)
    print()
    print('| kernel | calls | total us | avg us | min us | max us | % |')
    print('|---|---:|---:|---:|---:|---:|---:|')
    for name, n, tot, avg, mn, mx in rows:
        short = name if len(name) < 110 else name[:107] + '...'
        print('| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.1f |' % (short, n, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
