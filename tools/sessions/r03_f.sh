#!/bin/bash
# round 3 session f: pass-major MFMA order (all split-fp16 kernels), backward kernels of the two native ops, benches
O=$PWD/gpurun_out/r03_f; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "bev_pool or locatt or local_context" > $O/pytest_bwd.log 2>&1; echo "bwd tests rc=$?"; tail -4 $O/pytest_bwd.log | cut -c1-300
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -4 $O/pytest_all.log | cut -c1-300
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config']['execution'][:16], {k: v for k, v in d['roofline_dense']['dense_launches_ms'].items()}, (d.get('configs3_strong') or {}).get('value'))
PY
}
FF3D_MFMA_ORDER=tile timeout 600 python bench.py --no-cpu-baseline --no-strong-probe > $O/bench_b32_halo_tile_order.json 2> $O/bench_b32_tile.err; show $O/bench_b32_halo_tile_order.json
timeout 600 python bench.py --no-cpu-baseline --no-strong-probe > $O/bench_b32.json 2> $O/bench_b32.err; show $O/bench_b32.json
FF3D_CONV_HALO8=1 timeout 600 python bench.py --no-cpu-baseline --no-strong-probe > $O/bench_b32_halo8.json 2> $O/bench_b32_halo8.err; show $O/bench_b32_halo8.json
timeout 300 python bench.py --batch 4 --steps 30 --warmup 5 --no-cpu-baseline --no-strong-probe > $O/bench_b4.json 2> $O/bench_b4.err; show $O/bench_b4.json
timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-strong-probe > $O/bench_b1.json 2> $O/bench_b1.err; show $O/bench_b1.json
FF3D_BENCH_FORCE_DIST=1 timeout 300 python bench.py --batch 4 --steps 40 --warmup 5 --no-cpu-baseline --no-strong-probe > $O/bench_b4_rccl1_eager.json 2> $O/bench_b4_rccl1_eager.err; show $O/bench_b4_rccl1_eager.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_b32 -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-strong-probe > $O/bench_b32_under_rocprof.json 2> $O/rocprof_b32.err )
DB=$(find $O/prof_b32 -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/b32_kernel_stats_last_step.txt 2>&1
find $O -name '*.db' -delete
head -24 $O/b32_kernel_stats_last_step.txt | cut -c1-150
