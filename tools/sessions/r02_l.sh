#!/bin/bash
O=gpurun_out/r02_l; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q --timeout 1500 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log | cut -c1-200
timeout 600 python bench.py > $O/bench_b32.json 2> $O/bench_b32.err; tail -2 $O/bench_b32.err; cut -c1-250 $O/bench_b32.json
for B in 1 4 8; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_b${B}.json 2> $O/bench_b${B}.err
  cut -c1-200 $O/bench_b${B}.json
done
