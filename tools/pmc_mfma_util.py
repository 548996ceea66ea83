#!/usr/bin/env python
"""MFMA-busy fraction per kernel from one rocprofv3 PMC pass with `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`.

SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs of the chip (cycles), GRBM_GUI_ACTIVE over its 8 XCDs, so
    MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)
(checked on the fp32 GEMMs: 85-90 % busy at 125-140 TFLOP/s of the 157 TFLOP/s fp32-MFMA peak).

    python tools/pmc_mfma_util.py gpurun_out/pmc_mfma/p_results.db > profiles/..._mfma_busy.txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute('select kernel_name, counter_name, dispatch_id, value, duration from counters_collection').fetchall()
    per = {}
    for name, cn, did, val, dur in rows:
        per.setdefault((name, did), {})[cn] = val
        per[(name, did)]['dur'] = dur
    agg = {}
    for (name, _), d in per.items():
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'GRBM_GUI_ACTIVE' in d and d['GRBM_GUI_ACTIVE'] > 0:
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += d['SQ_VALU_MFMA_BUSY_CYCLES']
            a[2] += d['GRBM_GUI_ACTIVE']
    print(f'# source: {path}')
    print(f'{"dispatches":>10} {"mfma_busy":>10} {"gui_active_cycles(sum)":>24}  kernel')
    for name, (n, busy, act) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        if busy == 0:
            continue
        print(f'{n:10d} {busy / (act * 128):10.3f} {act:24.0f}  {name[:120]}')


if __name__ == '__main__':
    main(sys.argv[1])
