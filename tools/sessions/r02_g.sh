#!/bin/bash
# round 2, GPU session g: full GPU suite on the range-normalised split-fp16 path + bench
O=gpurun_out/r02_g; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 1500 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 600 python bench.py > $O/bench_b32.json 2> $O/bench_b32.err; tail -3 $O/bench_b32.err; cut -c1-300 $O/bench_b32.json
for B in 1 4 8; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_b${B}.json 2> $O/bench_b${B}.err
  cut -c1-240 $O/bench_b${B}.json
done
