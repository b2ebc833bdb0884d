#!/usr/bin/env python
"""Dump every counter of rocprofv3 --pmc rocpd DBs, averaged per launch, per kernel.

    python tools/rocpd_counters.py gpurun_out/pmc_a/*.db gpurun_out/pmc_b/*.db [--match conv_igemm]
"""
import sqlite3
import sys


def main(argv):
    match = None
    if '--match' in argv:
        i = argv.index('--match')
        match = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    table = {}
    for path in argv:
        db = sqlite3.connect(path)
        for name, counter, n, total, dur in db.execute(
                'select kernel_name, counter_name, count(*), sum(value), sum(duration) from counters_collection '
                'group by kernel_name, counter_name'):
            if match and match not in name:
                continue
            table.setdefault(name, {})[counter] = (n, total / n, dur / n)
    for name, cs in table.items():
        print('## `%s`' % (name if len(name) < 100 else name[:97] + '...'))
        print()
        print('| counter | launches | value / launch | avg duration us |')
        print('|---|---:|---:|---:|')
        for c in sorted(cs):
            n, v, d = cs[c]
            print('| %s | %d | %.4g | %.1f |' % (c, n, v, d / 1e3))
        print()


if __name__ == '__main__':
    main(sys.argv[1:])
