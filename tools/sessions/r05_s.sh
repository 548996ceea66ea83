#!/bin/bash
# round 5 session s: fused feed-forward kernel, block height A/B (register spills at MT = 5?)
O=$PWD/gpurun_out/r05_s; mkdir -p $O
export TMPDIR=/tmp
for mt in auto 5 4 3 2; do
  if [ $mt = auto ]; then timeout 120 python tools/bench_ffn_rows.py 19200 1024; else FF3D_FFN_MT=$mt timeout 120 python tools/bench_ffn_rows.py 19200 1024; fi
done 2>&1 | tee $O/ffn_rows_block_height_ab.txt
