#!/bin/bash
# round 4 session a: new / changed parity tests, halo-conv cache-policy + priority A/B, batches-in-flight experiment,
# project-after-gather A/B, default bench line (with other_workloads)
O=$PWD/gpurun_out/r04_a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_shape_gpu.py::test_head_batch4_c256_configs3_per_gpu_step tests/test_baseline_configs_gpu.py::test_config2_lc_chain_full_size_vs_oracle tests/test_train_forward_gpu.py tests/test_small_batch_gpu.py -x -q -m gpu > $O/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -3 $O/pytest_new.log | cut -c1-400
for abl in 0 32 64 96 0 32; do echo -n "FF3D_HALO_ABLATE=$abl: " | tee -a $O/halo_policy_ab.txt; FF3D_HALO_ABLATE=$abl timeout 120 python tools/experiments/exp_halo.py 2>&1 | tail -1 | tee -a $O/halo_policy_ab.txt; done
for m in eager graph graph2 eager_cc graph_cc graph2_cc graph2; do
  echo -n "$m: " | tee -a $O/pipeline_ab.txt
  timeout 200 python tools/experiments/exp_pipeline.py --mode $m --batch 4 --steps 80 2>$O/pipeline_$m.err | tail -1 | tee -a $O/pipeline_ab.txt; echo " rc=$?" | tee -a $O/pipeline_ab.txt
done
for s in 3 4; do echo -n "graph2 slots=$s: " | tee -a $O/pipeline_ab.txt; timeout 200 python tools/experiments/exp_pipeline.py --mode graph2 --slots $s --batch 4 --steps 80 2>>$O/pipeline_slots.err | tail -1 | tee -a $O/pipeline_ab.txt; done
for bb in 1 2 8 16; do echo -n "graph2 batch=$bb: " | tee -a $O/pipeline_ab.txt; timeout 200 python tools/experiments/exp_pipeline.py --mode graph2 --batch $bb --steps 60 2>>$O/pipeline_slots.err | tail -1 | tee -a $O/pipeline_ab.txt; echo -n "graph batch=$bb: " | tee -a $O/pipeline_ab.txt; timeout 200 python tools/experiments/exp_pipeline.py --mode graph --batch $bb --steps 60 2>>$O/pipeline_slots.err | tail -1 | tee -a $O/pipeline_ab.txt; done
timeout 300 python tools/experiments/exp_project_after_gather.py > $O/project_after_gather.txt 2>&1; echo "pag rc=$?"; tail -3 $O/project_after_gather.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-300 $O/bench_default.json
