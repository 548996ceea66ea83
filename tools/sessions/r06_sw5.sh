#!/bin/bash
# round 6, third session: roi_mlp.0 at 1 / 4 frames on the swapped-operand 256 x 128 instance (FF3D_GEMM_SWAP_MINM=512) vs the 128 x 128 tiles (default below 4 096 rows)
O=$PWD/gpurun_out/r06_sw5; mkdir -p $O
for rep in 1 2; do
for mm in 512 4096; do
FF3D_GEMM_SWAP_MINM=$mm timeout 300 python bench.py --latency-b1 --steps 5 --warmup 2 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['latency_b1_ms']; print('swap min M $mm: one frame graph replay %.4f ms (device %.4f), verified %s' % (d['graph_replay']['mean'], d['graph_replay_device']['mean'], d['verified']['bit_identical']))" | tee -a $O/ab.txt
FF3D_GEMM_SWAP_MINM=$mm timeout 300 python bench.py --batch 4 --steps 300 --warmup 30 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('swap min M $mm: batch 4 pipelined %.4f ms per step, %.1f frames/s, verified %s' % (d['ms_per_step'], d['value'], d['verified'].get('bit_identical')))" | tee -a $O/ab.txt
done
done
