#!/bin/bash
# round 6, third session: the decoder's projections on the own kernels at EVERY row count in eager steps too (FF3D_LIN_MIN_ROWS=0; today
# below 1 536 rows eager steps hand them to the vendor GEMM): whole GPU suite + the batch-1 / 4-frame eager figures, both settings
O=$PWD/gpurun_out/r06_mr0; mkdir -p $O
export TMPDIR=/tmp
( time FF3D_LIN_MIN_ROWS=0 timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/tests_minrows0.log 2>&1; echo "rc=$?" >> $O/tests_minrows0.log
tail -5 $O/tests_minrows0.log | cut -c1-300
for mr in 0 1536 0 1536; do
  for b in 1 4; do
    FF3D_LIN_MIN_ROWS=$mr timeout 300 python bench.py --graph off --batch $b --steps 50 --warmup 10 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('min_rows $mr batch $b eager: %.3f ms per step, %.1f frames/s' % (d['ms_per_step'], d['value']))" | tee -a $O/eager_small_batch.txt
  done
done
