#!/bin/bash
# round 4 session q: soak runs of the pipelined graph mode (thousands of replays, with and without the captured collective), PMC bytes of
# the wide NMS kernel
O=$PWD/gpurun_out/r04_q; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
show() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]).read().strip().splitlines():
    if line.startswith('{'):
        d = json.loads(line)
        print(sys.argv[1].split('/')[-1], d['value'], d['steps'], d['ms_per_step'], d['config']['execution'][:70], d['config']['detections_last_batch'])
PY
}
b() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/$name.json 2> $O/$name.err; echo "rc=$?"; show $O/$name.json; }
b soak_b4_5000 --batch 4 --steps 5000 --warmup 5
FF3D_BENCH_FORCE_DIST=1 b soak_b4_rccl1_5000 --batch 4 --steps 5000 --warmup 5
b soak_b1_10000 --batch 1 --steps 10000 --warmup 5
b soak_b32_400 --steps 400
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_$C -o p -- python $R/tools/experiments/exp_nms.py > $O/pmc_nms_$C.log 2> $O/pmc_nms_$C.err )
  python tools/pmc_summary.py $(find $O/pmc_$C -name '*_results.db' | head -1) heatmap_nms > $O/pmc_nms_$C.txt 2>&1
  find $O/pmc_$C -name '*.db' -delete
  grep -i "nms" $O/pmc_nms_$C.txt | head -4 | cut -c1-200
done
