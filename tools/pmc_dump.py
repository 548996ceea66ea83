#!/usr/bin/env python
"""Per-kernel average of every counter in a rocprofv3 PMC result database (rocpd sqlite)."""
import sqlite3
import sys


def main(path, pattern=''):
    c = sqlite3.connect(path)
    rows = c.execute('select kernel_name, counter_name, dispatch_id, value from counters_collection').fetchall()
    agg = {}
    for name, cn, did, val in rows:
        if pattern and pattern not in name:
            continue
        a = agg.setdefault((name[:70], cn), [0, 0.0])
        a[0] += 1
        a[1] += val
    for (name, cn), (n, tot) in sorted(agg.items()):
        print(f'{name:70s} {cn:34s} n={n:3d} avg={tot / n:16.1f}')


if __name__ == '__main__':
    main(*sys.argv[1:])
