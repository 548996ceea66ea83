#!/bin/bash
O=$PWD/gpurun_out/r02_an; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/experiments/exp_roi.py 2>&1 | tail -1 | tee -a $O/roi.txt
B=4 timeout 200 python tools/experiments/exp_roi.py 2>&1 | tail -1 | tee -a $O/roi.txt
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "roi" > $O/pytest_roi.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_roi.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c75-100 $O/bench.json
