#!/bin/bash
# round 6, third session: K-sliced fc2 with the 16-byte-lane sum-LayerNorm, threshold 1 536 rows - tests, micro-benchmark, whole suite, latency A/B
O=$PWD/gpurun_out/r06_ksl2; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python tools/experiments/exp_small_rows.py 2>&1 | grep -v amdgpu.ids | grep "fc2\|add + LN alone" | tee $O/small_rows.txt
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -6 $O/tests.log | cut -c1-300
for v in 1 0 1 0; do
FF3D_LIN_LN_KSLICES=$v timeout 300 python bench.py --latency-b1 --steps 5 --warmup 2 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['latency_b1_ms']; print('kslices $v: graph replay %.4f ms (device %.4f), eager %.4f, verified %s' % (d['graph_replay']['mean'], d['graph_replay_device']['mean'], d['eager']['mean'], d['verified']['bit_identical']))" | tee -a $O/latency.txt
done
for v in 1 0 1 0; do
FF3D_LIN_LN_KSLICES=$v timeout 300 python bench.py --batch 2 --steps 200 --warmup 20 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('kslices $v batch 2 pipelined: %.3f ms per step, %.1f frames/s, verified %s' % (d['ms_per_step'], d['value'], d['verified'].get('bit_identical')))" | tee -a $O/latency.txt
done
