#!/bin/bash
# round 6, third session: split-K pyramid convs as the default rule - new tests, whole GPU suite, one-frame / four-frame figures
O=$PWD/gpurun_out/r06_ks2; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_round6_gpu.py -x -q -m gpu -k "split_k" 2>&1 | tail -5 | tee $O/tests_new.txt
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -6 $O/tests.log | cut -c1-300
for ksp in 1 0 1 0; do
  for b in 1 4; do
    FF3D_CONV_KSPLIT=$ksp timeout 300 python bench.py --graph on --batch $b --steps 200 --warmup 20 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('conv ksplit $ksp batch $b graph (%s): %.3f ms per step, %.1f frames/s, verified %s' % (d['config']['execution'][:40], d['ms_per_step'], d['value'], d['verified'].get('bit_identical')))" | tee -a $O/small_batch.txt
  done
done
timeout 300 python bench.py --latency-b1 --steps 5 --warmup 2 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(json.dumps(d.get('latency_b1_ms'))[:1500])" | tee $O/latency.txt
