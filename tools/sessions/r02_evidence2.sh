#!/bin/bash
# round 2 evidence session 2 (after the weight-stationary value GEMM): default bench (+ CPU baselines), rocprofv3 kernel stats,
# PMC passes (one counter set per run, --pmc with --kernel-trace only), batch sweep
O=$PWD/gpurun_out/r02_ev2; mkdir -p $O
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
timeout 300 python bench.py --channels 128 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_b32_c128.json 2> $O/bench_b32_c128.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_b4 -o r -- python $R/bench.py --batch 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_b4_under_rocprof.json 2> $O/rocprof_b4.err )
python tools/rocprof_last_step.py $(find $O/prof_b4 -name '*_results.db' | head -1) 50 > $O/b4_kernel_stats_last_step.txt 2>&1
find $O -name '*.db' -delete
timeout 300 python tools/bench_waymo_shape.py 8 > $O/waymo_shape_b8.json 2> $O/waymo.err
timeout 300 python tools/bench_neck.py 32 > $O/neck_b32.json 2> $O/neck.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r02_ev2/bench_*.json')) + sorted(glob.glob('gpurun_out/r02_ev2/*shape*.json')) + sorted(glob.glob('gpurun_out/r02_ev2/neck*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(os.path.basename(f), d.get('value', d), d.get('ms_per_step'))
    except Exception as e:
        print(os.path.basename(f), 'FAILED', str(e)[:80])
PY
head -22 $O/b32_kernel_stats_last_step.txt | cut -c1-150
cat $O/pmc_mfma_busy.txt | head -20
