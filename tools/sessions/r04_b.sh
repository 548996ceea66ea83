#!/bin/bash
# round 4 session b: RCCL all-gather captured INSIDE the graph (thread-local capture mode), batches in flight at 32 frames
O=$PWD/gpurun_out/r04_b; mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; echo -n "$name: " | tee -a $O/pipeline_ab.txt; timeout 300 python tools/experiments/exp_pipeline.py "$@" 2>$O/$name.err | tail -1 | tee -a $O/pipeline_ab.txt; echo " rc=${PIPESTATUS[0]}" | tee -a $O/pipeline_ab.txt; }
run graph_cc_tl --mode graph_cc --capture-mode thread_local --batch 4 --steps 80
run graph2_cc_tl --mode graph2_cc --capture-mode thread_local --batch 4 --steps 80
run graph2_cc_tl_s4 --mode graph2_cc --capture-mode thread_local --batch 4 --steps 80 --slots 4
TORCH_NCCL_ASYNC_ERROR_HANDLING=0 TORCH_NCCL_ENABLE_MONITORING=0 run graph_cc_nowd --mode graph_cc --batch 4 --steps 80
run graph2_relaxed --mode graph2_cc --capture-mode relaxed --batch 4 --steps 80
run eager_b32 --mode eager --batch 32 --steps 12
run graph_b32 --mode graph --batch 32 --steps 12
run graph2_b32 --mode graph2 --batch 32 --steps 12
run graph2_b16 --mode graph2 --batch 16 --steps 24
run graph2_b32_s3 --mode graph2 --batch 32 --steps 12 --slots 3
