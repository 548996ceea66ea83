#!/bin/bash
O=$PWD/gpurun_out/r02_aj; mkdir -p $O
export TMPDIR=/tmp
for B in 1 2 4 8; do
  for g in off auto; do
    timeout 300 python bench.py --batch $B --graph $g --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_b${B}_$g.json 2> $O/bench_b${B}_$g.err; echo "B=$B graph=$g rc=$? $(cut -c75-100 $O/bench_b${B}_$g.json)"
  done
done
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_b32.json 2> $O/bench_b32.err; echo "B=32 rc=$? $(cut -c75-100 $O/bench_b32.json)"
timeout 120 python tools/debug_graph3.py event_sync > $O/g3.log 2>&1; tail -1 $O/g3.log
