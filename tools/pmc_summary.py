#!/usr/bin/env python
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database (`rocprofv3 --pmc X --kernel-trace`).

    python tools/pmc_summary.py gpurun_out/pmc_fetch/f_results.db [name-substring ...]
"""
import sqlite3
import sys


def main(path, subs):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute('pragma table_info(counters_collection)')]
    name_col = 'kernel_name' if 'kernel_name' in cols else 'name'
    rows = c.execute(f'select {name_col}, counter_name, count(*), avg(value), min(value), max(value) from counters_collection '
                     f'group by {name_col}, counter_name order by 4 desc').fetchall()
    print(f'# source: {path}; columns: {cols}')
    print(f'{"dispatches":>10} {"avg":>16} {"min":>16} {"max":>16}  counter  kernel')
    for name, cn, n, avg, mn, mx in rows:
        if subs and not any(s in name for s in subs):
            continue
        print(f'{n:10d} {avg:16.1f} {mn:16.1f} {mx:16.1f}  {cn}  {name[:110]}')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2:])
