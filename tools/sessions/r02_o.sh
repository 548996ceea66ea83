#!/bin/bash
O=gpurun_out/r02_o; mkdir -p $O
timeout 900 python -m pytest tests/test_training_gpu.py tests/test_ops_gpu.py -m gpu -q --timeout 600 -k "training or iou3d or gaussian or targets or split_f16 or topk or rowbias or merge_aug" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log | cut -c1-250
for B in 32 4; do
  timeout 300 python bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_b${B}.json 2> $O/bench_b${B}.err
  cut -c1-200 $O/bench_b${B}.json
done
