#!/bin/bash
# round 6: two taps per barrier as the 8 x 32 geometry's default - conv / config tests, then lc and waymo with and without it
O=$PWD/gpurun_out/r06_h3; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -k "conv or halo or waymo or baseline_configs or neck or lss or i2p or head_full" 2>&1 | tail -4 > $O/tests.txt
for wl in lc waymo; do
  for rep in 1 2; do
    timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl tap2', d['value'], d['ms_per_step'], d['verified'].get('bit_identical'))" >> $O/bench.txt
    FF3D_HALO_TAP2=0 timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl one tap', d['value'], d['ms_per_step'], d['verified'].get('bit_identical'))" >> $O/bench.txt
  done
done
cat $O/tests.txt $O/bench.txt
