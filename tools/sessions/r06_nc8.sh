#!/bin/bash
# round 6: the NCHW-source conv at the neck's input convs (shared_conv_pts / the mb2 neck's input conv): neck tests + lc bench A/B
O=$PWD/gpurun_out/r06_nc8; mkdir -p $O
timeout 1500 python -m pytest tests/test_head_gpu.py tests/test_baseline_configs_gpu.py tests/test_round5_gpu.py -q -m gpu -k "encoder or neck or lc or chain or lss or i2p or LC" 2>&1 | tail -6 | tee $O/tests.txt
for rep in 1 2; do
  timeout 600 python bench.py --workload lc --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lc nchw-source input conv', d['value'], d['ms_per_step'], d['verified'].get('bit_identical'))" | tee -a $O/bench.txt
  FF3D_HALO_NCHW_SRC=0 timeout 600 python bench.py --workload lc --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lc conversion + conv     ', d['value'], d['ms_per_step'], d['verified'].get('bit_identical'))" | tee -a $O/bench.txt
done
