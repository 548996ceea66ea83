#!/bin/bash
# round 6: 4 x 62 stored pixels + two taps per barrier (GEO 2) for pair outputs at 180 x 180: micro-benchmark, tests, default bench A/B
O=$PWD/gpurun_out/r06_h4; mkdir -p $O
run() { timeout 300 python tools/experiments/exp_halo.py 2>&1 | grep "nchw" >> $O/ab.txt; }
for rep in 1 2; do
  B=32 run
  B=32 FF3D_HALO_TAP2=0 run
  B=4 run
  B=4 FF3D_HALO_TAP2=0 run
  B=2 H=90 W=100 run
  B=2 H=90 W=100 FF3D_HALO_TAP2=0 run
done
cat $O/ab.txt
timeout 1500 python -m pytest tests -q -m gpu -x -k "conv or halo or head or baseline_configs or neck or small_batch" 2>&1 | tail -4 > $O/tests.txt; cat $O/tests.txt
for rep in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default geo2+tap2', d['value'], d['ms_per_step'], d['verified'].get('bit_identical'))" >> $O/bench.txt
  FF3D_HALO_TAP2=0 timeout 600 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default one tap', d['value'], d['ms_per_step'], d['verified'].get('bit_identical'))" >> $O/bench.txt
done
cat $O/bench.txt
