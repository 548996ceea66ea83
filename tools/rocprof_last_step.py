#!/usr/bin/env python
"""Per-kernel totals of the LAST bench step in a rocprofv3 rocpd database (steady state: excludes the one-time
MIOpen/hipBLASLt find / warm-up kernels).  A step is delimited by consecutive box_decode_kernel dispatches.

    python tools/rocprof_last_step.py gpurun_out/prof2/r_results.db
"""
import sqlite3
import sys


def main(path, top=40):
    c = sqlite3.connect(path)
    ends = [r[0] for r in c.execute("select end from kernels where name like '%box_decode_kernel%' order by end")]
    lo, hi = ends[-2], ends[-1]
    rows = c.execute('select name, count(*), sum(end-start), avg(end-start) from kernels where start > ? and end <= ? '
                     'group by name order by 3 desc', (lo, hi)).fetchall()
    total = sum(r[2] for r in rows)
    print(f'# source: {path}; last step: {sum(r[1] for r in rows)} dispatches, kernel time {total / 1e6:.3f} ms, '
          f'wall {(hi - lo) / 1e6:.3f} ms')
    print(f'{"calls":>6} {"total_us":>11} {"pct":>6} {"avg_us":>10}  name')
    for name, n, tot, avg in rows[:top]:
        print(f'{n:6d} {tot / 1e3:11.1f} {100 * tot / total:6.2f} {avg / 1e3:10.2f}  {name[:130]}')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
