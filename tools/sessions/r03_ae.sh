#!/bin/bash
# round 3 session ae: default bench line with the FULL cpu_baseline protocol (5 warm-up + 20 timed frames, tools/analysis_tools/benchmark.py:62-91)
O=$PWD/gpurun_out/r03_ae; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py --cpu-full-protocol > $O/bench_default_cpu_full_protocol.json 2> $O/bench.err; echo "bench rc=$?"
python - $O/bench_default_cpu_full_protocol.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['cpu_baseline'])
PY
