#!/bin/bash
# round 4 session d: waymo workload under the pipelined runner (hang in session c), whole GPU suite, halo N-tile-major A/B,
# rocprofv3 kernel stats + PMC passes of the bench command
O=$PWD/gpurun_out/r04_d; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
show() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]).read().strip().splitlines():
    if line.startswith('{'):
        d = json.loads(line)
        print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config']['execution'][:90])
PY
}
b() { name=$1; shift; PYTHONFAULTHANDLER=1 timeout -s ABRT 240 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/$name.json 2> $O/$name.err; echo "rc=$?"; show $O/$name.json; grep -v amdgpu.ids $O/$name.err | tail -25 | cut -c1-200; }
b bench_waymo_auto --workload waymo
b bench_waymo_slots1 --workload waymo --slots 1
b bench_waymo_eager --workload waymo --graph off
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -5 $O/pytest_all.log | cut -c1-300
for abl in 0 128 0 128; do echo -n "FF3D_HALO_ABLATE=$abl: " | tee -a $O/halo_ntile_major_ab.txt; FF3D_LIB=$R/focalformer3d_amd/lib/libff3d_hip_exp.so FF3D_HALO_ABLATE=$abl timeout 120 python tools/experiments/exp_halo.py 2>&1 | tail -1 | tee -a $O/halo_ntile_major_ab.txt; done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_b32 -o r -- python $R/bench.py --graph off --steps 6 --warmup 3 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_under_rocprof_b32_eager.json 2> $O/rocprof_b32.err )
DB=$(find $O/prof_b32 -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/bench_b32_eager_kernel_stats_last_step.txt 2>&1
python tools/rocprof_summary.py $DB 40 > $O/bench_b32_eager_kernel_stats_all.txt 2>&1
find $O/prof_b32 -name '*.db' -delete
head -12 $O/bench_b32_eager_kernel_stats_last_step.txt | cut -c1-150
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_b32p -o r -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_under_rocprof_b32_pipelined.json 2> $O/rocprof_b32p.err )
DB=$(find $O/prof_b32p -name '*_results.db' | head -1)
python tools/rocprof_summary.py $DB 40 > $O/bench_b32_pipelined_kernel_stats_all.txt 2>&1
find $O/prof_b32p -name '*.db' -delete
head -8 $O/bench_b32_pipelined_kernel_stats_all.txt | cut -c1-150
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_$C -o p -- python $R/bench.py --graph off --steps 3 --warmup 1 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/pmc_$C.json 2> $O/pmc_$C.err )
  python tools/pmc_summary.py $(find $O/pmc_$C -name '*_results.db' | head -1) msda_fwd conv3x3_halo splitmm split_nchw bev_flatten roi_grid linear > $O/pmc_$C.txt 2>&1
  find $O/pmc_$C -name '*.db' -delete
  grep -i "msda\|halo" $O/pmc_$C.txt | head -4 | cut -c1-170
done
