O=$PWD/gpurun_out/r06_z; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python $R/bench.py --value-mode gather_first --graph off --steps 5 --warmup 3 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions > $O/bench_under_rocprof_gf.json 2> $O/rocprof_gf.err )
DB=$(find $O/prof -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 40 > $O/bench_gather_first_tables_kernel_stats_last_step.txt 2>&1
rm -rf $O/prof
head -22 $O/bench_gather_first_tables_kernel_stats_last_step.txt | cut -c1-150
