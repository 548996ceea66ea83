#!/bin/bash
# round 2 evidence session: default bench (+ CPU baselines), rocprofv3 kernel stats, PMC passes (one counter set per run,
# --pmc with --kernel-trace only), batch sweep, vendor mode, 1-rank RCCL + 2-rank gloo rehearsals, auxiliary benches
O=$PWD/gpurun_out/r02_ev; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > $O/bench_b32.json 2> $O/bench_b32.err; cut -c1-160 $O/bench_b32.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_b32 -o r -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_b32_under_rocprof.json 2> $O/rocprof_b32.err )
DB=$(find $O/prof_b32 -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/b32_kernel_stats_last_step.txt 2>&1
python tools/rocprof_summary.py $DB 60 > $O/b32_kernel_stats_all.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_$C -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_$C.json 2> $O/pmc_$C.err )
  python tools/pmc_summary.py $(find $O/pmc_$C -name '*_results.db' | head -1) msda_fwd conv3x3_halo splitmm split_nchw bev_flatten roi_grid topk > $O/pmc_$C.txt 2>&1
done
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_mfma -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_mfma.json 2> $O/pmc_mfma.err )
python tools/pmc_mfma_util.py $(find $O/pmc_mfma -name '*_results.db' | head -1) > $O/pmc_mfma_busy.txt 2>&1
find $O -name '*.db' -delete
for B in 1 2 4 8 16 64; do
  timeout 300 python bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_b${B}.json 2> $O/bench_b${B}.err
done
timeout 300 python bench.py --dense vendor --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_b32_vendor.json 2> $O/bench_b32_vendor.err
timeout 300 python bench.py --channels 128 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_b32_c128.json 2> $O/bench_b32_c128.err
FF3D_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_b32_rccl_1rank.json 2> $O/bench_b32_rccl_1rank.err
FF3D_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --batch 16 --steps 5 --warmup 2 > $O/bench_gloo2_weak.json 2> $O/bench_gloo2_weak.err
FF3D_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --global-batch 32 --steps 5 --warmup 2 > $O/bench_gloo2_strong.json 2> $O/bench_gloo2_strong.err
timeout 300 python tools/bench_waymo_shape.py 8 > $O/waymo_shape_b8.json 2> $O/waymo.err
timeout 300 python tools/bench_neck.py 32 > $O/neck_b32.json 2> $O/neck.err
timeout 300 python tools/bench_lss.py 1 > $O/lss_b1.json 2> $O/lss.err
timeout 300 python tools/bench_i2p.py > $O/i2p.json 2> $O/i2p.err
for v in replay_only old_flow bias_relu; do timeout 120 python tools/debug_graph2.py $v > $O/graph_$v.log 2>&1; echo "graph $v rc=$? iters=$(grep -c 'OK iter' $O/graph_$v.log)" >> $O/graph_fault.txt; done
python - <<'PY'
import json, glob, os
O = os.environ.get('O', 'gpurun_out/r02_ev')
for f in sorted(glob.glob('gpurun_out/r02_ev/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(os.path.basename(f), d['value'], d['ms_per_step'], d['n_gpus'], d['scaling'], d['roofline']['frac'], d.get('configs3_strong', {}).get('value'))
    except Exception as e:
        print(os.path.basename(f), 'FAILED', str(e)[:80])
PY
cat $O/graph_fault.txt; head -12 $O/pmc_FETCH_SIZE.txt | cut -c1-160; head -8 $O/pmc_mfma_busy.txt | cut -c1-160
for f in waymo_shape_b8 neck_b32 lss_b1 i2p; do echo $f; tail -2 $O/$f.json | cut -c1-300; done
