#!/bin/bash
# round 6, third session: K-sliced fc2 + sum-LayerNorm - new tests, micro-benchmark, decoder / small-batch / head tests, latency
O=$PWD/gpurun_out/r06_ksl; mkdir -p $O
timeout 600 python -m pytest tests/test_round6_gpu.py -x -q -m gpu -k "k_sliced" 2>&1 | tail -5 | tee $O/tests_new.txt
timeout 400 python tools/experiments/exp_small_rows.py 2>&1 | grep -v amdgpu.ids | tee $O/small_rows.txt
for v in 1 0 1 0; do
FF3D_LIN_LN_KSLICES=$v timeout 300 python bench.py --latency-b1 --steps 5 --warmup 2 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['latency_b1_ms']; print('kslices $v: graph replay %.4f ms (device %.4f), eager %.4f, verified %s' % (d['graph_replay']['mean'], d['graph_replay_device']['mean'], d['eager']['mean'], d['verified']['bit_identical']))" | tee -a $O/latency.txt
done
