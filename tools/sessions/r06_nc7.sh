#!/bin/bash
# round 6: the NCHW-source halo conv as a product entry point (ABI 2.09): its tests, the experiment table, head suites, default bench A/B
O=$PWD/gpurun_out/r06_nc7; mkdir -p $O
timeout 900 python -m pytest tests/test_round6_gpu.py -q -m gpu -x -k "nchw_source" 2>&1 | tail -15 | tee $O/tests_nchw.txt
FF3D_LIB=$PWD/focalformer3d_amd/lib/libff3d_hip_exp.so B=32 H=180 W=180 timeout 300 python tools/experiments/exp_halo_nchw.py 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
for rep in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nchw-source convs', d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), d['config']['single_stream_eager']['value'])" | tee -a $O/bench.txt
  FF3D_HALO_NCHW_SRC=0 timeout 600 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('conversion + conv ', d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), d['config']['single_stream_eager']['value'])" | tee -a $O/bench.txt
done
