#!/bin/bash
O=gpurun_out/r02_q; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q --timeout 1500 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log | cut -c1-250
for B in 32 4 1 8; do
  timeout 300 python bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_b${B}.json 2> $O/bench_b${B}.err
  cut -c1-200 $O/bench_b${B}.json
done
