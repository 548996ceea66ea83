#!/bin/bash
# round 6: the opt-in fused value projection (FF3D_FUSE_VALUE=1: un-embedded pair + periodic bias-table GEMM) on the round's tree
O=$PWD/gpurun_out/r06_fv; mkdir -p $O
for rep in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'], d['verified'].get('bit_identical'))" >> $O/bench.txt
  FF3D_FUSE_VALUE=1 timeout 600 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused value', d['value'], d['ms_per_step'], d['verified'].get('bit_identical'))" >> $O/bench.txt
done
cat $O/bench.txt
