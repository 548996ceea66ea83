#!/usr/bin/env python
"""Where the GPU idles inside one step of a rocprofv3 kernel trace: the last interval between two dispatches of a delimiter kernel
(default: the optimizer's multi_tensor_apply), its busy time (union of the kernel intervals) and the largest idle gaps with the
kernels on either side.

    python tools/rocprof_gaps.py r_results.db [delimiter substring] [top]
"""
import sqlite3
import sys


def main(path, delim='multi_tensor_apply', top=25):
    c = sqlite3.connect(path)
    ends = [r[0] for r in c.execute('select end from kernels where name like ? order by end', (f'%{delim}%',))]
    # steps = runs of delimiter kernels separated by other work: take the last two run ends
    allk = c.execute('select start, end, name from kernels order by start').fetchall()
    runs, prev_is = [], False
    for s, e, n in allk:
        is_d = delim in n
        if prev_is and not is_d:
            runs.append(last_end)
        if is_d:
            last_end = e
        prev_is = is_d
    if prev_is:
        runs.append(last_end)
    lo, hi = runs[-2], runs[-1]
    ks = [(s, e, n) for s, e, n in allk if s >= lo and e <= hi]
    busy, cur_s, cur_e, gaps = 0, None, None, []
    for s, e, n in ks:
        if cur_e is None:
            cur_s, cur_e, last_n = s, e, n
            gaps.append((s - lo, '(step start)', n))
            continue
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, last_n, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        last_n = n
    busy += cur_e - cur_s
    print(f'# {path}: last step {len(ks)} dispatches, wall {(hi - lo) / 1e6:.3f} ms, GPU busy {busy / 1e6:.3f} ms, idle {(hi - lo - busy) / 1e6:.3f} ms')
    hist = [0, 0, 0, 0]
    for g, _, _ in gaps:
        hist[0 if g < 5e3 else 1 if g < 20e3 else 2 if g < 100e3 else 3] += g
    print(f'# idle by gap size: <5us {hist[0] / 1e6:.2f} ms, 5-20us {hist[1] / 1e6:.2f} ms, 20-100us {hist[2] / 1e6:.2f} ms, >100us {hist[3] / 1e6:.2f} ms')
    t0 = lo
    for g, a, b in sorted(gaps, reverse=True)[:top]:
        print(f'{g / 1e3:9.1f} us  after {a[:70]:70s} before {b[:70]}')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 'multi_tensor_apply', int(sys.argv[3]) if len(sys.argv) > 3 else 25)
