#!/bin/bash
# round 5 session f: kernel tables (rocprofv3) of the default / lc / waymo eager steps on the round's tree; ws bf16 A/B (NJ, occupancy, 8-byte
# stores restored); flatten grid order A/B at kernel level
O=$PWD/gpurun_out/r05_f; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() { name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o r -- python $R/bench.py --graph off --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_under_rocprof_$name.json 2> $O/rocprof_$name.err )
  DB=$(find $O/prof_$name -name '*_results.db' | head -1)
  python tools/rocprof_last_step.py $DB 60 > $O/bench_${name}_kernel_stats_last_step.txt 2>&1
  find $O/prof_$name -name '*.db' -delete; rm -rf $O/prof_$name
  head -22 $O/bench_${name}_kernel_stats_last_step.txt | cut -c1-170
}
prof b32 --steps 5 --warmup 3
prof lc --workload lc --steps 4 --warmup 2
prof waymo --workload waymo --steps 4 --warmup 2
FF3D_FLATTEN_ORDER=frame-slowest prof waymo_flatten_old --workload waymo --steps 4 --warmup 2
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b waymo --workload waymo --steps 10
FF3D_GEMM_WS_BF16_NJ=2 FF3D_GEMM_WS_BF16_OCC=2 b waymo_nj2_occ2 --workload waymo --steps 10
b default
python - <<'PY'
import json
for n in ('waymo', 'waymo_nj2_occ2', 'default'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_f/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), {k: v for k, v in d['roofline_dense']['dense_launches_ms'].items() if 'bf16' in k})
    except Exception as e:
        print(n, 'no line', e)
PY
