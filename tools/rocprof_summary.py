#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (``rocprofv3 --kernel-trace --stats -d DIR -o NAME`` writes
DIR/NAME_results.db on ROCm 7.2) into the per-kernel table committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof1/r1_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path, top=60):
    c = sqlite3.connect(path)
    rows = c.execute('select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) '
                     'from kernels group by name order by 3 desc').fetchall()
    total = sum(r[2] for r in rows)
    print(f'# source: {path}')
    print(f'# kernels: {len(rows)} distinct, {sum(r[1] for r in rows)} dispatches, total {total / 1e6:.3f} ms')
    print(f'{"calls":>7} {"total_us":>12} {"pct":>6} {"avg_us":>10} {"min_us":>10} {"max_us":>10}  name')
    for name, n, tot, avg, mn, mx in rows[:top]:
        print(f'{n:7d} {tot / 1e3:12.1f} {100 * tot / total:6.2f} {avg / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f}  {name[:150]}')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)
