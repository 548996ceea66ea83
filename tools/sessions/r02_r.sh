#!/bin/bash
O=gpurun_out/r02_r; mkdir -p $O
timeout 900 python -m pytest tests/test_head_gpu.py tests/test_bench_shape_gpu.py -m gpu -q --timeout 800 -k "golden or full_size or batch32" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log | cut -c1-250
for B in 32 4 8; do
  timeout 300 python bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_b${B}.json 2> $O/bench_b${B}.err
  cut -c1-200 $O/bench_b${B}.json
done
