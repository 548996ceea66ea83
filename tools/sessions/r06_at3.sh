#!/bin/bash
# round 6: training attention on the fp32 matrix cores, tiles prefetched into registers one tile ahead: debug case, errors + times, the test
O=$PWD/gpurun_out/r06_at3; mkdir -p $O
timeout 120 python tools/experiments/exp_mha_dbg.py 2>&1 | tail -20 | head -6
timeout 300 python tools/experiments/exp_mha_train.py > $O/mfma.txt 2>&1
cat $O/mfma.txt
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "masked_self_attention_training" 2>&1 | tail -4 > $O/tests_attn.txt; cat $O/tests_attn.txt
